"""Golden vectors for the densify / prune / reset-opacity row, produced by running the REFERENCE's own
`GaussianModel` (scene/gaussian_model.py:465-497, 563-566, 667-937) on CPU in this container.

    python tests/golden/make_densify_golden.py      # needs /root/reference; writes tests/golden/densify_reference_*.npz

The model is driven exactly as train.py:158-175 drives it: real `torch.optim.Adam` groups from `training_setup`, two Adam
steps so every group carries non-trivial moments, `add_densification_stats` + the max-radii update for a few views,
then `densify_and_prune` (or `prune`, or `reset_opacity`).  The only intervention is `torch.normal`, which is replaced
by `mean + std * Z` with a stored standard-normal table Z so the split children are reproducible without the CUDA
Philox stream.  Only inputs and outputs are stored; nothing of the reference's source is copied.
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (auto-mock finder + CPU factory patches)

GROUPS_STAGE1 = ["xyz", "normal", "rotation", "scaling", "opacity", "f_dc", "f_rest"]
GROUPS_STAGE2 = GROUPS_STAGE1 + ["base_color", "roughness", "incidents_dc", "incidents_rest", "visibility_dc",
                                 "visibility_rest"]
ATTR = dict(xyz="_xyz", normal="_normal", rotation="_rotation", scaling="_scaling", opacity="_opacity", f_dc="_shs_dc",
            f_rest="_shs_rest", base_color="_base_color", roughness="_roughness", incidents_dc="_incidents_dc",
            incidents_rest="_incidents_rest", visibility_dc="_visibility_dc", visibility_rest="_visibility_rest")
SHAPES = dict(xyz=(3,), normal=(3,), rotation=(4,), scaling=(3,), opacity=(1,), f_dc=(1, 3), f_rest=(15, 3),
              base_color=(3,), roughness=(1,), incidents_dc=(1, 3), incidents_rest=(15, 3), visibility_dc=(1, 1),
              visibility_rest=(15, 1))


def build_model(GaussianModel, P, stage2, g, extent):
    from torch import nn
    m = GaussianModel(3, render_type="neilf" if stage2 else "render")
    names = GROUPS_STAGE2 if stage2 else GROUPS_STAGE1

    def rnd(*s):
        return torch.randn(*s, generator=g)
    vals = {n: 0.5 * rnd(P, *SHAPES[n]) for n in names}
    vals["xyz"] = 1.3 * rnd(P, 3)
    # log-scales straddling percent_dense*extent (clone vs split) with a few above 0.1*extent (world-size prune)
    vals["scaling"] = np.log(0.01 * extent) + 1.2 * rnd(P, 3)
    vals["opacity"] = 2.5 * rnd(P, 1) - 1.0         # some sigmoid(opacity) < 0.005
    for n in names:
        setattr(m, ATTR[n], nn.Parameter(vals[n].clone().contiguous().requires_grad_(True)))
    m.max_radii2D = torch.zeros(P)
    m.spatial_lr_scale = 1.0
    args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                                 position_lr_delay_mult=0.01, position_lr_max_steps=30000, normal_lr=2e-3,
                                 rotation_lr=1e-3, scaling_lr=5e-3, opacity_lr=5e-2, sh_lr=2.5e-3, base_color_lr=1e-2,
                                 roughness_lr=1e-2, light_lr=1e-3, light_rest_lr=-1, visibility_lr=2.5e-3,
                                 visibility_rest_lr=-1)
    m.training_setup(args)
    # two Adam steps with random gradients: non-trivial exp_avg / exp_avg_sq in every group
    for _ in range(2):
        for grp in m.optimizer.param_groups:
            p = grp["params"][0]
            p.grad = 0.1 * rnd(*p.shape)
        m.step()
    return m, names


def snapshot(m, names, prefix):
    out = {}
    for grp in m.optimizer.param_groups:
        n = grp["name"]
        p = grp["params"][0]
        st = m.optimizer.state[p]
        out["%s_%s" % (prefix, n)] = p.detach().numpy().copy()
        out["%s_%s_exp_avg" % (prefix, n)] = st["exp_avg"].numpy().copy()
        out["%s_%s_exp_avg_sq" % (prefix, n)] = st["exp_avg_sq"].numpy().copy()
    for s in ("weights_accum", "xyz_gradient_accum", "normal_gradient_accum", "denom", "max_radii2D"):
        out["%s_%s" % (prefix, s)] = getattr(m, s).numpy().copy()
    return out


def accumulate_views(m, P, g, views, rec):
    """train.py:160-165 for `views` synthetic views; the per-view inputs are stored so the HIP accumulate kernel can be
    replayed on them."""
    for v in range(views):
        radii = torch.randint(0, 24, (P,), generator=g, dtype=torch.int32)
        radii[torch.rand(P, generator=g) < 0.3] = 0
        vis = radii > 0
        vsp = torch.zeros(P, 3)
        vsp.grad = 4e-4 * torch.randn(P, 3, generator=g) * (torch.rand(P, 1, generator=g) < 0.7)
        m._normal.grad = 3e-9 * torch.randn(P, 3, generator=g)
        weights = torch.rand(P, 1, generator=g) * (torch.rand(P, 1, generator=g) < 0.8) * 2e-4
        m.add_densification_stats(vsp, vis, weights)
        m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], radii[vis])
        rec["view%d_radii" % v] = radii.numpy().copy()
        rec["view%d_viewspace_grad" % v] = vsp.grad.numpy().copy()
        rec["view%d_normal_grad" % v] = m._normal.grad.numpy().copy()
        rec["view%d_weights" % v] = weights.numpy().copy()
    m._normal.grad = None


def case(GaussianModel, name, P, stage2, seed, op, max_screen_size, extent=4.0, tg=2e-4, tn=2e-9, views=3):
    g = torch.Generator().manual_seed(seed)
    m, names = build_model(GaussianModel, P, stage2, g, extent)
    rec = dict(op=op, stage2=int(stage2), extent=extent, grad_threshold=tg, grad_normal_threshold=tn,
               min_opacity=0.005, max_screen_size=0.0 if max_screen_size is None else float(max_screen_size),
               percent_dense=m.percent_dense, weights_threshold=1e-4, views=views, group_names=np.array(names))
    rec.update(snapshot(m, names, "pre"))          # state before the view statistics
    accumulate_views(m, P, g, views, rec)
    rec.update({k: v for k, v in snapshot(m, names, "in").items() if not any(
        k == "in_%s%s" % (n, s) for n in names for s in ("", "_exp_avg", "_exp_avg_sq"))})   # stats after the views
    Z = torch.randn(2 * P, 3, generator=g)
    rec["normal_table"] = Z.numpy().copy()

    def fake_normal(mean, std, **kw):
        n = std.shape[0]
        return mean + std * Z[:n]
    with mock.patch.object(torch, "normal", fake_normal):
        if op == "densify_and_prune":
            m.densify_and_prune(tg, 0.005, extent, max_screen_size, tn)
        elif op == "prune":
            m.prune(0.005, extent, max_screen_size)
        elif op == "reset_opacity":
            m.reset_opacity()
    rec.update(snapshot(m, names, "out"))
    path = os.path.join(HERE, "densify_reference_%s.npz" % name)
    np.savez_compressed(path, **rec)
    print("%-28s P %d -> %d  (%d KB)" % (name, P, m._xyz.shape[0], os.path.getsize(path) // 1024))


def main():
    sys.meta_path.append(mg._Finder())
    sys.path.insert(0, mg.REF)
    for p in mg._cpu_factories():
        p.start()
    from scene.gaussian_model import GaussianModel
    case(GaussianModel, "stage1_densify", 192, False, 11, "densify_and_prune", 20)
    case(GaussianModel, "stage1_densify_nosize", 128, False, 12, "densify_and_prune", None)
    case(GaussianModel, "stage2_densify", 64, True, 13, "densify_and_prune", 20)
    case(GaussianModel, "stage1_prune", 160, False, 14, "prune", 20)
    case(GaussianModel, "stage1_reset_opacity", 96, False, 15, "reset_opacity", None)
    lr_schedule_fixture()
    composition_fixture()
    checkpoint_fixtures()
    ssim_fixture()
    wrapper_trace_fixture()
    camera_fixture()
    dataset_fixture()




def lr_schedule_fixture():
    """tests/golden/lr_schedule_reference.npz: values of the reference's get_expon_lr_func (utils/general_utils.py:30-63)
    for the position learning rate; call after main()'s import setup."""
    from utils.general_utils import get_expon_lr_func
    steps = np.array([0, 1, 10, 100, 999, 1000, 5000, 15000, 29999, 30000, 40000, -3])
    out = {}
    for name, args in (("default", dict(lr_init=0.00016 * 4.0, lr_final=0.0000016 * 4.0, lr_delay_mult=0.01, max_steps=30000)),
                       ("delayed", dict(lr_init=1e-3, lr_final=1e-5, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=2000)),
                       ("disabled", dict(lr_init=0.0, lr_final=0.0, max_steps=100))):
        f = get_expon_lr_func(**args)
        out[name] = np.array([f(int(s)) for s in steps], dtype=np.float64)
        out[name + "_args"] = np.array([args.get("lr_init"), args.get("lr_final"), args.get("lr_delay_mult", 1.0),
                                        args.get("max_steps"), args.get("lr_delay_steps", 0)], dtype=np.float64)
    np.savez(os.path.join(HERE, "lr_schedule_reference.npz"), steps=steps, **out)


def composition_fixture():
    """tests/golden/composition_reference.npz: GaussianModel.set_transform + create_from_gaussians + the incident reset of
    scene_composition (relighting.py:28-52) on two small stage-2 models; call after main()'s import setup."""
    from torch import nn
    from scene.gaussian_model import GaussianModel
    g = torch.Generator().manual_seed(42)
    names = GROUPS_STAGE2
    rec, models = {}, []
    for j, P in enumerate((40, 25)):
        m = GaussianModel(3, render_type="neilf")
        for n in names:
            v = 0.5 * torch.randn(P, *SHAPES[n], generator=g)
            if n == "rotation":
                v = torch.nn.functional.normalize(torch.randn(P, 4, generator=g), dim=-1)
            setattr(m, ATTR[n], nn.Parameter(v.clone().requires_grad_(True)))
            rec["obj%d_%s" % (j, n)] = v.numpy().copy()
        A = torch.linalg.qr(torch.randn(3, 3, generator=g)).Q
        if torch.det(A) < 0:
            A[:, 0] = -A[:, 0]
        T = torch.eye(4)
        T[:3, :3] = A * (0.5 + j)                         # rotation x uniform scale
        T[:3, 3] = torch.randn(3, generator=g)
        rec["obj%d_transform" % j] = T.numpy().copy()
        m.set_transform(transform=T)
        models.append(m)
    comp = GaussianModel.create_from_gaussians(models, types.SimpleNamespace(sh_degree=3))
    comp._incidents_dc.data[:] = 0
    comp._incidents_rest.data[:] = 0
    for n in names:
        rec["out_" + n] = getattr(comp, ATTR[n]).detach().numpy().copy()
    rec["group_names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "composition_reference.npz"), **rec)


def checkpoint_fixtures():
    """tests/golden/checkpoint_reference_stage{1,2}.pth: `(GaussianModel.capture(), iteration)` exactly as train.py:194-195
    saves it, from small models with two Adam steps and three views of densification statistics.  Also checks, here where
    the reference is importable, that the reference's own restore() + optimizer.load_state_dict() accept what
    relightable3dgaussian_amd.checkpoint.capture() writes (the round trip through this repo's fused step object)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from scene.gaussian_model import GaussianModel
    from relightable3dgaussian_amd import checkpoint as ck
    from relightable3dgaussian_amd.fused_step import FusedStage1Step
    from relightable3dgaussian_amd.densify import DensificationStats
    for stage2, P, seed in ((False, 24, 51), (True, 16, 52)):
        g = torch.Generator().manual_seed(seed)
        m, names = build_model(GaussianModel, P, stage2, g, 4.0)
        accumulate_views(m, P, g, 3, {})
        obj = (m.capture(), 777)
        path = os.path.join(HERE, "checkpoint_reference_stage%d.pth" % (2 if stage2 else 1))
        torch.save(obj, path)
        if not stage2:
            r = ck.restore(path)
            step = FusedStage1Step(r)                       # CPU tensors: construction launches nothing
            ck.load_moments(step, r)
            step.stats = DensificationStats(P, torch.device("cpu"))
            for n in ck.STAT_NAMES:
                getattr(step.stats, n).copy_(r.stats[n])
            step.stats.max_radii2D.copy_(r.max_radii2D)
            lrs = {grp["name"]: grp["lr"] for grp in m.optimizer.param_groups}
            ours = ck.capture(step, 777, spatial_lr_scale=r.spatial_lr_scale, learning_rates=lrs)
            fresh = GaussianModel(3, render_type="render")
            args = types.SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6,
                                         position_lr_delay_mult=0.01, position_lr_max_steps=30000, normal_lr=2e-3,
                                         rotation_lr=1e-3, scaling_lr=5e-3, opacity_lr=5e-2, sh_lr=2.5e-3)
            fresh.restore(ours[0], args, is_training=True, restore_optimizer=False)
            fresh.optimizer.load_state_dict(ours[0][13])     # restore() swallows exceptions: call it in the open
            for grp_a, grp_b in zip(fresh.optimizer.param_groups, m.optimizer.param_groups):
                assert grp_a["name"] == grp_b["name"]
                pa, pb = grp_a["params"][0], grp_b["params"][0]
                assert torch.equal(pa.data, pb.data), grp_a["name"]
                sa, sb = fresh.optimizer.state[pa], m.optimizer.state[pb]
                assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])
                assert float(sa["step"]) == float(sb["step"]) == 2.0
            assert torch.equal(fresh.denom, m.denom) and torch.equal(fresh.max_radii2D, m.max_radii2D)
            print("reference restore() + load_state_dict() accept checkpoint.capture():", path)


def ssim_fixture():
    """tests/golden/ssim_reference.npz: the reference's SSIM (utils/loss_utils.py:20-63) and L1 on two random images, with
    the gradient w.r.t. the rendered image -- the pin of train_step.ssim / image_loss, which in turn are the parity
    targets of the HIP SSIM kernels.  (kornia, imported by that module, is auto-mocked; ssim does not use it.)"""
    from utils.loss_utils import ssim, tv_loss
    l1_loss = torch.nn.functional.l1_loss               # neilf.py:226: Ll1 = F.l1_loss(rendered_image, gt_image)
    g = torch.Generator().manual_seed(77)
    rec = {}
    for name, (H, W) in (("a", (40, 56)), ("b", (13, 9))):          # b: smaller than the window in one direction
        x = torch.rand(3, H, W, generator=g).requires_grad_(True)
        y = (x.detach() + 0.2 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
        s = ssim(x, y)
        l1 = l1_loss(x, y)
        loss = (1.0 - 0.2) * l1 + 0.2 * (1.0 - s)                   # train.py / render.py: lambda_dssim = 0.2
        loss.backward()
        rec[name + "_tv"] = float(tv_loss(x.detach()))
        from utils.image_utils import psnr
        rec[name + "_psnr"] = psnr(x.detach(), y).numpy().copy()
        rec.update({name + "_x": x.detach().numpy().copy(), name + "_y": y.numpy().copy(), name + "_ssim": float(s),
                    name + "_l1": float(l1), name + "_loss": float(loss), name + "_grad": x.grad.numpy().copy()})
    # rgb_to_srgb (clip=True) with its gradient: values below 0, around the linear/power knee, inside (0,1) and above 1
    from utils.graphics_utils import rgb_to_srgb
    v = torch.cat([torch.linspace(-0.2, 0.01, 40), torch.linspace(0.002, 0.005, 40), torch.linspace(0.01, 2.5, 112)])
    v = v.reshape(3, 8, 8).clone().requires_grad_(True)
    w = torch.rand(3, 8, 8, generator=g) + 0.5
    out = rgb_to_srgb(v)
    (out * w).sum().backward()
    rec.update(srgb_in=v.detach().numpy().copy(), srgb_out=out.detach().numpy().copy(), srgb_w=w.numpy().copy(),
               srgb_grad=v.grad.numpy().copy())
    np.savez_compressed(os.path.join(HERE, "ssim_reference.npz"), **rec)


def wrapper_trace_fixture():
    """tests/golden/wrapper_trace_reference.json: what the reference's autograd wrapper
    (gaussian_renderer/r3dg_rasterization.py:58-261) hands to `_C.rasterize_gaussians` / `_backward` and where it routes the
    nine gradients, recorded by tests/wrapper_trace.py with a fake `_C`."""
    import json
    sys.path.insert(0, os.path.dirname(HERE))
    import wrapper_trace
    import gaussian_renderer.r3dg_rasterization as ref

    def install(fwd, bwd):
        ref._C = types.SimpleNamespace(rasterize_gaussians=fwd, rasterize_gaussians_backward=bwd)
    doc = {v: wrapper_trace.run(ref.GaussianRasterizationSettings, ref.GaussianRasterizer, install, v)
           for v in ("sh_scale", "color_cov")}
    import bvh as ref_bvh

    def install_bvh(create, trace):
        ref_bvh._C.create_bvh = create
        ref_bvh._C.trace_bvh_opacity = trace
    doc["raytracer"] = wrapper_trace.run_raytracer(ref_bvh.RayTracer, install_bvh)
    with open(os.path.join(HERE, "wrapper_trace_reference.json"), "w") as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)


def camera_fixture():
    """tests/golden/camera_reference.npz: world_view_transform / full_proj_transform / camera_center / c2w of the reference's
    `Camera` (scene/cameras.py:8-73, built the way relighting.py:150-158 builds it: R = w2c[:3,:3]^T, T = w2c[:3,3], FoV)
    for a few poses and image sizes -- the pin of synthetic.look_at_camera, whose matrices every benchmark and parity test
    hands to the ops."""
    from scene.cameras import Camera
    rec = {}
    poses = [((3.2, 1.0, 1.5), (0.0, 0.0, 0.0), 800, 800), ((-2.0, 3.1, 0.4), (0.2, -0.1, 0.3), 640, 400),
             ((0.5, -4.0, 2.5), (0.0, 0.0, 0.0), 1800, 700)]
    for j, (eye, target, W, H) in enumerate(poses):
        e, t = np.asarray(eye, np.float64), np.asarray(target, np.float64)
        fwd = (t - e) / np.linalg.norm(t - e)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        Rw2c = np.stack([right, down, fwd], 0)
        w2c = np.eye(4)
        w2c[:3, :3] = Rw2c
        w2c[:3, 3] = -Rw2c @ e
        fovx = 0.6911112070083618
        fovy = 2 * np.arctan(H / (2 * (W / (2 * np.tan(fovx / 2)))))
        cam = Camera(colmap_id=0, R=w2c[:3, :3].T.astype(np.float32), T=w2c[:3, 3].astype(np.float32), FoVx=fovx,
                     FoVy=fovy, fx=None, fy=None, cx=None, cy=None, image=torch.zeros(3, H, W), image_name=None, uid=0,
                     data_device="cpu")
        rec.update({"cam%d_eye" % j: e, "cam%d_target" % j: t, "cam%d_size" % j: np.array([W, H]),
                    "cam%d_world_view_transform" % j: cam.world_view_transform.numpy().copy(),
                    "cam%d_full_proj_transform" % j: cam.full_proj_transform.numpy().copy(),
                    "cam%d_camera_center" % j: cam.camera_center.numpy().copy(),
                    "cam%d_fovy" % j: np.array(fovy)})
    rec["n"] = np.array(len(poses))
    np.savez(os.path.join(HERE, "camera_reference.npz"), **rec)


def dataset_fixture():
    """tests/golden/blender_dataset_reference.npz: a small split written by synthetic.write_blender_dataset and read back by
    the reference's own readCamerasFromTransforms (scene/dataset_readers.py:215-270; imageio, absent here, is shimmed
    with PIL for the PNG read): R, T, FovX, FovY and the image arrays the reference ends up with."""
    import tempfile
    from PIL import Image
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from relightable3dgaussian_amd import synthetic as syn
    import scene.utils as scene_utils
    import scene.dataset_readers as readers
    scene_utils.imageio.imread = lambda path, **kw: np.asarray(Image.open(path))
    readers.load_img_rgb = scene_utils.load_img_rgb
    g = torch.Generator().manual_seed(5)
    cams = syn.orbit_cameras(5, width=24, height=24)[:3]
    imgs = [torch.rand(3, 24, 24, generator=g) for _ in cams]
    rec = {}
    with tempfile.TemporaryDirectory() as root:
        syn.write_blender_dataset(root, cams, imgs, split="train")
        infos = readers.readCamerasFromTransforms(root, "transforms_train.json", True)
    assert len(infos) == 3
    for j, (ci, img) in enumerate(zip(infos, imgs)):
        rec.update({"v%d_R" % j: np.asarray(ci.R, np.float64), "v%d_T" % j: np.asarray(ci.T, np.float64),
                    "v%d_fov" % j: np.array([ci.FovX, ci.FovY]), "v%d_image" % j: np.asarray(ci.image, np.float64),
                    "v%d_written" % j: img.numpy().copy(), "v%d_name" % j: np.array(ci.image_name)})
    norm = readers.getNerfppNorm(infos)                      # scene.cameras_extent = norm["radius"]
    rec["extent_radius"] = np.array(norm["radius"])
    rec["extent_translate"] = np.asarray(norm["translate"], np.float64)
    rec["n"] = np.array(3)
    np.savez_compressed(os.path.join(HERE, "blender_dataset_reference.npz"), **rec)


if __name__ == "__main__":
    main()
