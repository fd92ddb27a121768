"""End-to-end golden fixtures from the REFERENCE'S OWN, UNMODIFIED Python (this container only).

    python tests/golden/make_pipeline_golden.py     # needs /root/reference; writes tests/golden/pipeline_reference_stage{1,2,2_syn4}.npz

What runs, unmodified, from /root/reference:
    scene/gaussian_model.py      GaussianModel (activations, get_* properties, update_visibility :312-342)
    scene/cameras.py             Camera (matrices, intrinsics)
    scene/direct_light_map.py    DirectLightMap (softplus texture, direct_light lookup)
    bvh/__init__.py              RayTracer (leaf boxes, 0.05 d origin offset)
    gaussian_renderer/r3dg_rasterization.py   the autograd wrapper around `_C`
    gaussian_renderer/neilf.py   render_view(is_training=True) + rendering_equation + calculate_loss      (stage 2)
    gaussian_renderer/render.py  render_view + calculate_loss                                              (stage 1)
    utils/loss_utils.py, utils/graphics_utils.py, utils/sh_utils.py, utils/general_utils.py, arguments/__init__.py
What stands in for what this container cannot build or import:
    r3dg_rasterization._C / bvh_tracing._C / simple_knn._C  ->  the CPU oracle (oracle/rasterizer.py, oracle/bvh.py: C restatements
        of the reference kernels, pinned to the real reference build on the GPU by tests/test_reference_gpu.py) behind modules
        with the reference's extension names and signatures -- i.e. exactly the drop-in boundary of include/r3dg_hip.h;
    kornia.filters.spatial_gradient  ->  relightable3dgaussian_amd.train_step.spatial_gradient (kornia 0.6.12's published algorithm);
    torchvision / plyfile / nvdiffrast / pyexr / imageio / cv2 / tensorboard / lpips ...  ->  import-only mocks (never called);
    device="cuda"  ->  CPU (factory functions patched, `.cuda()` is the identity).
The fixtures hold the inputs, every rendered map, the loss and the gradient of every parameter: tests/test_reference_pipeline_gpu.py
feeds the same inputs to the HIP pipeline (autograd glue and fused iteration) on the GPU box, where /root/reference does not exist.
Nothing of the reference is copied: inputs and outputs only.
"""
import importlib.machinery
import os
import sys
import types
from argparse import ArgumentParser
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

import make_golden as mg  # noqa: E402  (the import-mock finder and the CPU factory patches)


def _t(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def install_oracle_extensions():
    """Modules named like the reference's compiled extensions, backed by the CPU oracle."""
    from oracle import bvh as obvh, rasterizer as orc

    def rasterize_gaussians(bg, means3D, features, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                            viewmatrix, projmatrix, tan_fovx, tan_fovy, cx, cy, H, W, sh, degree, campos, prefiltered,
                            computer_pseudo_normal, debug):
        out = orc.rasterize_gaussians(bg, means3D, features, colors, opacity, scales, rotations, scale_modifier,
                                      cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, cx, cy, H, W, sh, degree,
                                      campos, prefiltered, computer_pseudo_normal, debug)
        R, n_contrib, color, opac, depth, feature, normal, xyz, weights, radii, state = out
        handle = torch.zeros(1, dtype=torch.uint8)
        handle.state = state                       # the three "opaque buffers": the state dict rides on the first one
        return (R, _t(n_contrib, torch.int32), _t(color), _t(opac), _t(depth), _t(feature), _t(normal), _t(xyz),
                _t(weights.astype(np.float32)), _t(radii, torch.int32), handle, torch.zeros(1, dtype=torch.uint8),
                torch.zeros(1, dtype=torch.uint8))

    def rasterize_gaussians_backward(bg, means3D, features, radii, colors, scales, rotations, scale_modifier, cov3D_precomp,
                                     viewmatrix, projmatrix, tan_fovx, tan_fovy, gC, gO, gD, gF, sh, degree, campos,
                                     geomBuffer, R, binningBuffer, imageBuffer, backward_geometry, debug):
        g = orc.rasterize_gaussians_backward(bg, means3D, features, radii, colors, scales, rotations, scale_modifier,
                                             cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, gC, gO, gD, gF, sh,
                                             degree, campos, geomBuffer.state, backward_geometry)
        return tuple(_t(np.asarray(x, np.float32)) for x in g[:9])

    rmod = types.ModuleType("r3dg_rasterization")
    rc = types.ModuleType("r3dg_rasterization._C")
    rc.rasterize_gaussians, rc.rasterize_gaussians_backward = rasterize_gaussians, rasterize_gaussians_backward
    rc.mark_visible = lambda means3D, vm, pm: torch.from_numpy(orc.mark_visible(means3D, vm, pm))
    rmod._C = rc
    rmod.__path__ = []
    sys.modules["r3dg_rasterization"], sys.modules["r3dg_rasterization._C"] = rmod, rc

    def create_bvh(means3D, scales, rotations, nodes, aabbs):
        n, a, m = obvh.create_bvh(nodes.numpy(), aabbs.numpy())
        nodes.copy_(torch.from_numpy(n))
        aabbs.copy_(torch.from_numpy(a))
        return nodes, aabbs, torch.from_numpy(m.astype(np.int64))

    def trace_bvh_opacity(nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities, normals):
        c, v = obvh.trace_bvh_opacity(nodes.numpy(), aabbs.numpy(), rays_o.numpy(), rays_d.numpy(), means3D.numpy(),
                                      covs3D.numpy(), opacities.numpy(), normals.numpy())
        return torch.from_numpy(c), torch.from_numpy(v)

    bmod = types.ModuleType("bvh_tracing")
    bc = types.ModuleType("bvh_tracing._C")
    bc.create_bvh, bc.trace_bvh_opacity = create_bvh, trace_bvh_opacity
    bmod._C = bc
    bmod.__path__ = []
    sys.modules["bvh_tracing"], sys.modules["bvh_tracing._C"] = bmod, bc

    kmod = types.ModuleType("simple_knn")
    kc = types.ModuleType("simple_knn._C")
    kc.distCUDA2 = lambda pts: torch.full((pts.shape[0],), 1e-3)          # import-time dependency of scene/__init__ only
    kmod._C = kc
    kmod.__path__ = []
    sys.modules["simple_knn"], sys.modules["simple_knn._C"] = kmod, kc

    from relightable3dgaussian_amd import train_step
    kf = types.ModuleType("kornia.filters")
    kf.spatial_gradient = lambda x, mode="sobel", order=1, normalized=True: train_step.spatial_gradient(x)
    kf.laplacian = mock.MagicMock()
    kmain = types.ModuleType("kornia")
    kmain.filters = kf
    kmain.__path__ = []
    sys.modules["kornia"], sys.modules["kornia.filters"] = kmain, kf


def make_inputs(P, res, seed, stage2):
    from relightable3dgaussian_amd import synthetic as syn
    scene = syn.make_scene(P=P, seed=seed, stage2=stage2, scale_log_mean=-2.6)
    cam = syn.look_at_camera((2.9, 1.1, 1.3), width=res, height=res)
    g = torch.Generator().manual_seed(seed + 17)
    raw = dict(xyz=scene["xyz"], normal=scene["normal"] * (0.5 + torch.rand(P, 1, generator=g)),
               scaling=torch.log(scene["scales"]), rotation=scene["rotations"] * (0.5 + torch.rand(P, 1, generator=g)),
               opacity=torch.logit(scene["opacity"].clamp(1e-4, 1 - 1e-4)), shs_dc=scene["shs"][:, :1].clone(),
               shs_rest=scene["shs"][:, 1:].clone())
    if stage2:
        raw.update(base_color=torch.randn(P, 3, generator=g), roughness=torch.randn(P, 1, generator=g),
                   incidents_dc=0.3 * torch.rand(P, 1, 3, generator=g), incidents_rest=0.05 * torch.randn(P, 15, 3, generator=g),
                   env=0.5 * torch.rand(1, 16, 32, 3, generator=g))
    gt = torch.rand(3, res, res, generator=g)
    gt = torch.nn.functional.avg_pool2d(gt[None], 5, 1, 2)[0].clamp(0, 1).contiguous()       # smooth-ish target
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, res), torch.linspace(-1, 1, res), indexing="ij")
    mask = (1.25 - 1.6 * (xx * xx + yy * yy).sqrt()).clamp(0, 1)[None].contiguous()
    bg = torch.tensor([1.0, 1.0, 1.0])
    return raw, cam, gt, mask, bg


def reference_camera(cam, gt, mask):
    from scene.cameras import Camera
    w2c = cam.world_view_transform.t().numpy().astype(np.float64)
    R = w2c[:3, :3].T                                   # the reference stores R transposed (scene/cameras.py, getWorld2View2)
    T = w2c[:3, 3]
    H, W = cam.image_height, cam.image_width
    fx, fy = W / (2 * cam.tanfovx), H / (2 * cam.tanfovy)
    c = Camera(colmap_id=0, R=R, T=T, FoVx=cam.FoVx, FoVy=cam.FoVy, fx=fx, fy=fy, cx=cam.cx, cy=cam.cy, image=gt,
               image_name="synthetic", uid=0, data_device="cpu", image_mask=mask)
    return c


def to_model(GaussianModel, raw, stage2):
    pc = GaussianModel(3, render_type="neilf" if stage2 else "render")
    P = lambda t: torch.nn.Parameter(t.clone().requires_grad_(True))
    pc._xyz, pc._normal, pc._scaling, pc._rotation, pc._opacity = (P(raw[k]) for k in ("xyz", "normal", "scaling", "rotation",
                                                                                      "opacity"))
    pc._shs_dc, pc._shs_rest = P(raw["shs_dc"]), P(raw["shs_rest"])
    if stage2:
        pc._base_color, pc._roughness = P(raw["base_color"]), P(raw["roughness"])
        pc._incidents_dc, pc._incidents_rest = P(raw["incidents_dc"]), P(raw["incidents_rest"])
    return pc


def options(stage2, iteration_lambdas):
    from arguments import OptimizationParams, PipelineParams
    parser = ArgumentParser()
    opt = OptimizationParams(parser)
    pipe = PipelineParams(parser)
    for k, v in iteration_lambdas.items():
        setattr(opt, k, v)
    return opt, pipe


def grads_of(pc, names):
    return {"g_" + n: getattr(pc, "_" + n).grad.detach().numpy().copy() for n in names}


def main():
    sys.meta_path.append(mg._Finder())
    install_oracle_extensions()
    sys.path.insert(0, REF)
    for p in mg._cpu_factories():
        p.start()
    torch.cuda.empty_cache = lambda: None
    from scene.gaussian_model import GaussianModel
    from scene.direct_light_map import DirectLightMap
    import gaussian_renderer.neilf as nf
    import importlib
    rd = importlib.import_module("gaussian_renderer.render") if not isinstance(sys.modules.get("gaussian_renderer.render"), types.ModuleType) else sys.modules["gaussian_renderer.render"]

    # ---------------- stage 2: script/run_nerf.sh:20-39 ----------------
    P, res, K = 1500, 96, 16
    raw, cam, gt, mask, bg = make_inputs(P, res, seed=41, stage2=True)
    pc = to_model(GaussianModel, raw, True)
    light = DirectLightMap(16)
    light.env = torch.nn.Parameter(raw["env"].clone().requires_grad_(True))
    opt, pipe = options(True, dict(lambda_light=0.01, lambda_env_smooth=0.01, lambda_base_color_smooth=0,
                                   lambda_roughness_smooth=0, lambda_light_smooth=0))
    rcam = reference_camera(cam, gt, mask)
    pc.update_visibility(K)
    results = nf.render_view(rcam, pc, pipe, bg, is_training=True, dict_params={"env_light": light})
    loss, tb = nf.calculate_loss(rcam, pc, results, opt, light)
    loss.backward()
    names = ("xyz", "normal", "scaling", "rotation", "opacity", "shs_dc", "shs_rest", "base_color", "roughness",
             "incidents_dc", "incidents_rest")
    out = {("raw_" + k): v.numpy() for k, v in raw.items()}
    out.update(grads_of(pc, names))
    out.update(g_env=light.env.grad.numpy().copy(), loss=np.float64(loss.item()), K=K, res=res,
               gt=gt.numpy(), mask=mask.numpy(), bg=bg.numpy(), wvt=cam.world_view_transform.numpy(),
               fpt=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(),
               cam_scalars=np.array([cam.FoVx, cam.FoVy, cam.tanfovx, cam.tanfovy, cam.cx, cam.cy], np.float64),
               ref_wvt=rcam.world_view_transform.numpy(), ref_fpt=rcam.full_proj_transform.numpy(),
               ref_campos=rcam.camera_center.numpy(), visibility=pc._visibility_tracing.numpy(),
               incident_dirs=pc._incident_dirs.numpy(), incident_areas=pc._incident_areas.numpy(),
               num_rendered=np.int64(results["num_rendered"]))
    for k in ("render", "pbr", "normal", "pseudo_normal", "opacity", "depth", "base_color", "roughness", "diffuse",
              "visibility", "diffuse_light"):
        out["map_" + k] = results[k].detach().numpy()
    out["tb"] = np.array([tb[k] for k in ("l1", "ssim", "l1_pbr", "ssim_pbr", "loss_light", "loss_env_smooth")], np.float64)
    np.savez_compressed(os.path.join(HERE, "pipeline_reference_stage2.npz"), **out)
    print("stage 2: loss %.6f, num_rendered %d, visible fraction %.3f" % (loss.item(), results["num_rendered"],
                                                                          (pc._visibility_tracing > 0).float().mean()))

    # ---------------- stage 2 with the Synthetic4Relight / DTU objective: script/run_syn4.sh:22-42, run_dtu.sh:24-45 ----------------
    # Same inputs, camera, target, mask and visibility caches as the fixture above; only the lambdas differ
    # (--lambda_base_color_smooth 1 --lambda_roughness_smooth 0.5 --lambda_light_smooth 1): the fixture stores the loss, its
    # terms and the gradient of every parameter (the scripts freeze the geometry groups through learning rate 0; autograd
    # still produces their gradients, which pin the smoothness terms' pull on opacity and on the rendered normal).
    pc = to_model(GaussianModel, raw, True)
    light = DirectLightMap(16)
    light.env = torch.nn.Parameter(raw["env"].clone().requires_grad_(True))
    opt, pipe = options(True, dict(lambda_light=0.01, lambda_env_smooth=0.01, lambda_base_color_smooth=1,
                                   lambda_roughness_smooth=0.5, lambda_light_smooth=1))
    pc.update_visibility(K)
    assert torch.equal(pc._visibility_tracing, torch.from_numpy(out["visibility"]))
    results = nf.render_view(rcam, pc, pipe, bg, is_training=True, dict_params={"env_light": light})
    loss, tb = nf.calculate_loss(rcam, pc, results, opt, light)
    loss.backward()
    syn4 = grads_of(pc, names)
    syn4.update(g_env=light.env.grad.numpy().copy(), loss=np.float64(loss.item()),
                tb=np.array([tb[k] for k in ("l1", "ssim", "l1_pbr", "ssim_pbr", "loss_light", "loss_env_smooth",
                                             "loss_base_color_smooth", "loss_roughness_smooth", "loss_light_smooth")],
                            np.float64),
                lambdas=np.array([1.0, 0.5, 1.0], np.float64))
    np.savez_compressed(os.path.join(HERE, "pipeline_reference_stage2_syn4.npz"), **syn4)
    print("stage 2 (run_syn4 objective): loss %.6f; smoothness terms %.5f %.5f %.5f" % (
        loss.item(), tb["loss_base_color_smooth"], tb["loss_roughness_smooth"], tb["loss_light_smooth"]))

    # ---------------- stage 1: script/run_nerf.sh:7-14 ----------------
    raw, cam, gt, mask, bg = make_inputs(P, res, seed=43, stage2=False)
    pc = to_model(GaussianModel, raw, False)
    opt, pipe = options(False, dict(lambda_normal_render_depth=0.01, lambda_normal_smooth=0.01, lambda_mask_entropy=0.1,
                                    lambda_depth_var=1e-2))
    rcam = reference_camera(cam, gt, mask)
    iteration = 6500
    pkg = rd.render_view(rcam, pc, pipe, bg, 1.0, None, computer_pseudo_normal=True)
    loss, tb = rd.calculate_loss(rcam, pc, pkg, opt, iteration)
    loss.backward()
    names = ("xyz", "normal", "scaling", "rotation", "opacity", "shs_dc", "shs_rest")
    out = {("raw_" + k): v.numpy() for k, v in raw.items()}
    out.update(grads_of(pc, names))
    out.update(loss=np.float64(loss.item()), res=res, iteration=iteration, gt=gt.numpy(), mask=mask.numpy(), bg=bg.numpy(),
               wvt=cam.world_view_transform.numpy(), fpt=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(),
               cam_scalars=np.array([cam.FoVx, cam.FoVy, cam.tanfovx, cam.tanfovy, cam.cx, cam.cy], np.float64),
               num_rendered=np.int64(pkg["num_rendered"]),
               g_viewspace=pkg["viewspace_points"].grad.numpy().copy())
    for k in ("render", "normal", "pseudo_normal", "opacity", "depth", "depth_var"):
        out["map_" + k] = pkg[k].detach().numpy()
    out["tb"] = np.array([tb[k] for k in ("loss_l1", "ssim", "loss_mask_entropy", "loss_normal_render_depth",
                                          "loss_normal_smooth", "loss_depth_var")], np.float64)
    np.savez_compressed(os.path.join(HERE, "pipeline_reference_stage1.npz"), **out)
    print("stage 1: loss %.6f, num_rendered %d" % (loss.item(), pkg["num_rendered"]))


if __name__ == "__main__":
    main()
