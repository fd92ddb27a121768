"""The HIP pipeline against the REFERENCE'S OWN, UNMODIFIED Python run end to end.

tests/golden/pipeline_reference_stage{1,2}.npz were produced by tests/golden/make_pipeline_golden.py, which imports the
reference's GaussianModel, Camera, DirectLightMap, RayTracer, the r3dg_rasterization autograd wrapper, render_view and
calculate_loss of gaussian_renderer/{neilf,render}.py unmodified and runs them on CPU with the compiled extensions
replaced by the CPU oracle behind the extension names (the drop-in boundary).  Here the same raw parameters, camera,
target image and object mask go through this repo's pipeline on the GPU:
  * the autograd path (drop-in ops + the PyTorch restatement of the glue, train_step.Stage2Step / bench_core.render_stage1 +
    train_step.stage1_loss), and
  * the fused iterations (fused_step.FusedStage2Step / FusedStage1Step),
and every rendered map, the loss and the gradient of every parameter must agree.  Tolerances: maps 2e-5 * max (+1e-5), loss
1e-5 relative, gradients 2e-3 * max on EVERY entry (round 5: the 0.4 % outlier allowance of earlier rounds was never used --
gpurun_out/r05_a_pipeline.log: 0 entries above the bound, largest 6.6e-4 of the scale -- and is gone)."""
import os
import types

import numpy as np
import pytest
import torch

from tests.helpers import report

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(stage):
    z = np.load(os.path.join(GOLD, "pipeline_reference_stage%d.npz" % stage))
    return {k: z[k] for k in z.files}


def _camera(z):
    from relightable3dgaussian_amd.synthetic import SynthCamera
    fovx, fovy, tanx, tany, cx, cy = [float(v) for v in z["cam_scalars"]]
    res = int(z["res"])
    t = lambda k: torch.from_numpy(z[k]).to(DEV)
    return SynthCamera(res, res, fovx, fovy, tanx, tany, cx, cy, t("wvt"), t("fpt"), t("campos"))


def _params(z, stage2):
    from relightable3dgaussian_amd.bench_core import GaussianParams
    p = GaussianParams.__new__(GaussianParams)
    P_ = lambda k: torch.nn.Parameter(torch.from_numpy(z["raw_" + k]).to(DEV).contiguous())
    p.xyz, p.normal, p.scaling, p.rotation, p.opacity = P_("xyz"), P_("normal"), P_("scaling"), P_("rotation"), P_("opacity")
    p.features_dc, p.features_rest = P_("shs_dc"), P_("shs_rest")
    p.stage2 = stage2
    if stage2:
        p.base_color, p.roughness = P_("base_color"), P_("roughness")
        p.incidents_dc, p.incidents_rest, p.env = P_("incidents_dc"), P_("incidents_rest"), P_("env")
    return p


class _Checker:
    def __init__(self):
        self.msgs, self.ok = [], True

    def __call__(self, name, got, want, rtol, atol=0.0):
        want = torch.as_tensor(np.asarray(want)).reshape(got.shape)
        ok, msg = report(name, got, want, rtol, atol)
        self.msgs.append(msg)
        self.ok &= ok

    def done(self):
        print("\n".join(self.msgs))
        assert self.ok, "\n".join(self.msgs)


def test_cameras_of_the_fixtures_are_the_reference_cameras():
    z = _load(2)
    assert np.abs(z["wvt"] - z["ref_wvt"]).max() < 2e-6 and np.abs(z["fpt"] - z["ref_fpt"]).max() < 2e-6
    assert np.abs(z["campos"] - z["ref_campos"]).max() < 2e-6


def test_stage2_iteration_matches_the_reference_python(monkeypatch):
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    from relightable3dgaussian_amd.train_step import Stage2Step
    z = _load(2)
    K = int(z["K"])
    cam = _camera(z)
    bg, gt = torch.from_numpy(z["bg"]).to(DEV), torch.from_numpy(z["gt"]).to(DEV)
    vis, dirs, areas = (torch.from_numpy(z[k]).to(DEV) for k in ("visibility", "incident_dirs", "incident_areas"))
    chk = _Checker()
    # ---- autograd path: drop-in ops + PyTorch glue
    p = _params(z, True)
    step = Stage2Step(p, None, DEV, K)                      # traces its own visibility with the HIP BVH
    near = (vis - 0.9).abs() < 1e-4
    cls = ((step.visibility == 0) != (vis == 0)) & ~near
    assert cls.float().mean().item() <= 1e-4, "visibility caches differ from the reference's (oracle-traced) caches"
    chk("incident_dirs", step.incident_dirs, z["incident_dirs"], 0, 5e-5)        # sin/cos of angles up to ~40 rad, GPU vs CPU libm
    step.visibility, step.incident_dirs, step.incident_areas = vis, dirs, areas     # identical caches from here on
    loss, outs = step(cam, bg, gt)
    loss.backward()
    assert outs[0] == int(z["num_rendered"])
    chk("render", outs[2], z["map_render"], 2e-5, 1e-5)
    chk("opacity", outs[3], z["map_opacity"], 2e-5, 1e-5)
    chk("pseudo_normal", outs[6], z["map_pseudo_normal"], 1e-3, 1e-4)
    chk("loss", loss.detach().reshape(1), np.array([z["loss"]], np.float32), 1e-5)
    names = {"xyz": p.xyz, "normal": p.normal, "scaling": p.scaling, "rotation": p.rotation, "opacity": p.opacity,
             "shs_dc": p.features_dc, "shs_rest": p.features_rest, "base_color": p.base_color, "roughness": p.roughness,
             "incidents_dc": p.incidents_dc, "incidents_rest": p.incidents_rest, "env": p.env}
    for k, t in names.items():
        chk("autograd g_" + k, t.grad, z["g_" + k], 2e-3, 1e-9)
    # ---- fused iteration, on BOTH shading paths; which one ran is asserted, not assumed: the fixture's directions were generated
    # on the CPU and sit ~2e-5 from the device's ray set (FixedRaySet.try_build admits 5e-5) -- a silent fall-back to the
    # general kernels would otherwise pass for a test of the fixed-ray-set kernels
    for want_frs in (True, False):
        monkeypatch.setenv("R3DG_SHADE_FRS", "1" if want_frs else "0")
        fused = FusedStage2Step(_params(z, True), K)
        fused.visibility, fused.incident_dirs, fused.incident_areas = vis, dirs, areas
        # ... and the normals those cached directions were generated FROM: the reference's get_normal on the CPU (F.normalize,
        # eps 1e-3).  The device's activation kernel differs from it by an ulp on six Gaussians whose normal is (0, 0, -c):
        # -1 there, -0.99999994 here -- and next to -z an ulp of the normal is a different rotation_between_z altogether
        fused._ray_normals = torch.nn.functional.normalize(torch.from_numpy(z["raw_normal"]), dim=-1, eps=1e-3).to(DEV)
        fo = fused.forward_backward(cam, bg, gt)
        torch.cuda.synchronize()
        from relightable3dgaussian_amd.shading_ops import FixedRaySet
        assert (fused._frs is not None) == want_frs, "shading path: wanted %s, ran %s (cached vs regenerated directions: %s)" % (
            "fixed ray set" if want_frs else "general", "fixed ray set" if fused._frs is not None else "general",
            FixedRaySet.last_mismatch)
        tag = "fused[%s]" % ("fixed-ray-set kernels, %d Gaussians off the rotated path" % fused._frs.n_invalid if want_frs
                             else "general kernels")
        chk.msgs.append("---- " + tag + ("; cached (CPU-generated) vs regenerated directions: %.2e" % FixedRaySet.last_mismatch
                                         if want_frs else ""))
        tag = "fused[frs]" if want_frs else "fused[general]"
        chk(tag + " render", fo[2], z["map_render"], 2e-5, 1e-5)
        chk(tag + " loss", fused.loss().reshape(1), np.array([z["loss"]], np.float32), 1e-5)
        g = fused.grads
        for k in ("xyz", "normal", "scaling", "rotation", "opacity", "base_color", "roughness", "env"):
            chk(tag + " g_" + k, g[k], z["g_" + k], 2e-3, 1e-9)
        chk(tag + " g_shs", g["shs"], np.concatenate([z["g_shs_dc"], z["g_shs_rest"]], 1), 2e-3, 1e-9)
        chk(tag + " g_incidents", g["incidents"], np.concatenate([z["g_incidents_dc"], z["g_incidents_rest"]], 1), 2e-3, 1e-9)
    chk.done()


def test_stage2_syn4_objective_matches_the_reference_python():
    """script/run_syn4.sh:22-42 / run_dtu.sh: the stage-2 objective with the three edge-aware smoothness terms
    (--lambda_base_color_smooth 1 --lambda_roughness_smooth 0.5 --lambda_light_smooth 1) through the reference's own
    calculate_loss (tests/golden/pipeline_reference_stage2_syn4.npz: same inputs, camera, target, object mask and visibility
    caches as the run_nerf.sh fixture) vs the autograd restatement, the fused iteration, and the fused FROZEN-GEOMETRY
    iteration those scripts' learning rates select."""
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    from relightable3dgaussian_amd.train_step import STAGE2_WEIGHTS_SYN4, Stage2Step
    z, y = _load(2), np.load(os.path.join(GOLD, "pipeline_reference_stage2_syn4.npz"))
    K = int(z["K"])
    cam = _camera(z)
    bg, gt, mask = (torch.from_numpy(z[k]).to(DEV) for k in ("bg", "gt", "mask"))
    vis, dirs, areas = (torch.from_numpy(z[k]).to(DEV) for k in ("visibility", "incident_dirs", "incident_areas"))
    chk = _Checker()
    p = _params(z, True)
    step = Stage2Step(p, None, DEV, K, loss_weights=STAGE2_WEIGHTS_SYN4)
    step.visibility, step.incident_dirs, step.incident_areas = vis, dirs, areas
    loss, outs = step(cam, bg, gt, mask)
    loss.backward()
    chk("loss", loss.detach().reshape(1), np.array([y["loss"]], np.float32), 1e-5)
    names = {"xyz": p.xyz, "normal": p.normal, "scaling": p.scaling, "rotation": p.rotation, "opacity": p.opacity,
             "shs_dc": p.features_dc, "shs_rest": p.features_rest, "base_color": p.base_color, "roughness": p.roughness,
             "incidents_dc": p.incidents_dc, "incidents_rest": p.incidents_rest, "env": p.env}
    for k, t in names.items():
        chk("autograd g_" + k, t.grad, y["g_" + k], 2e-3, 1e-9)
    cat = lambda a, b: np.concatenate([y[a], y[b]], 1)
    # fused iteration, everything trains: every gradient
    fused = FusedStage2Step(_params(z, True), K, loss_weights=STAGE2_WEIGHTS_SYN4)
    fused.visibility, fused.incident_dirs, fused.incident_areas = vis, dirs, areas
    fused.forward_backward(cam, bg, gt, image_mask=mask)
    torch.cuda.synchronize()
    chk("fused loss", fused.loss().reshape(1), np.array([y["loss"]], np.float32), 1e-5)
    tb = y["tb"]
    N = gt.shape[-1] * gt.shape[-2]
    sm = fused.sums.sum(1)[7:10].cpu().numpy() / (3.0 * N)
    chk("fused smoothness terms", torch.from_numpy(sm), tb[6:9].astype(np.float32), 1e-4)
    for k in ("xyz", "normal", "scaling", "rotation", "opacity", "base_color", "roughness", "env"):
        chk("fused g_" + k, fused.grads[k], y["g_" + k], 2e-3, 1e-9)
    chk("fused g_shs", fused.grads["shs"], cat("g_shs_dc", "g_shs_rest"), 2e-3, 1e-9)
    chk("fused g_incidents", fused.grads["incidents"], cat("g_incidents_dc", "g_incidents_rest"), 2e-3, 1e-9)
    # frozen geometry (the scripts' learning rates): the groups that train get the reference's gradients, the others none
    lrs = dict(xyz=0.0, normal=0.0, scaling=0.0, rotation=0.0, opacity=0.0, shs=0.0, shs_rest=0.0, base_color=0.01,
               roughness=0.01, incidents=0.001, incidents_rest=0.0001, env=0.1)
    fr = FusedStage2Step(_params(z, True), K, loss_weights=STAGE2_WEIGHTS_SYN4, lrs=lrs)
    assert fr.frozen_geometry
    fr.visibility, fr.incident_dirs, fr.incident_areas = vis, dirs, areas
    fr.forward_backward(cam, bg, gt, image_mask=mask)
    torch.cuda.synchronize()
    chk("frozen loss", fr.loss().reshape(1), np.array([y["loss"]], np.float32), 1e-5)
    for k in ("base_color", "roughness", "env"):
        chk("frozen g_" + k, fr.grads[k], y["g_" + k], 2e-3, 1e-9)
    chk("frozen g_incidents", fr.grads["incidents"], cat("g_incidents_dc", "g_incidents_rest"), 2e-3, 1e-9)
    assert all(float(fr.grads[k].abs().max()) == 0.0 for k in ("xyz", "normal", "scaling", "rotation", "opacity", "shs"))
    chk.done()


def test_stage1_iteration_matches_the_reference_python():
    from relightable3dgaussian_amd.bench_core import render_stage1
    from relightable3dgaussian_amd.fused_step import FusedStage1Step
    from relightable3dgaussian_amd.train_step import stage1_loss
    z = _load(1)
    cam = _camera(z)
    bg, gt, mask = (torch.from_numpy(z[k]).to(DEV) for k in ("bg", "gt", "mask"))
    it = int(z["iteration"])
    chk = _Checker()
    p = _params(z, False)
    outs = render_stage1(p, cam, bg)
    loss = stage1_loss(outs, gt, mask, None, it)
    loss.backward()
    assert outs[0] == int(z["num_rendered"])
    chk("render", outs[2], z["map_render"], 2e-5, 1e-5)
    chk("opacity", outs[3], z["map_opacity"], 2e-5, 1e-5)
    feat = outs[5] / outs[3].clamp_min(1e-5) * (outs[1] > 0)
    chk("normal map", feat[:3], z["map_normal"], 1e-4, 1e-5)
    chk("depth map", feat[3:4], z["map_depth"], 1e-4, 1e-5)
    chk("loss", loss.detach().reshape(1), np.array([z["loss"]], np.float32), 1e-5)
    names = {"xyz": p.xyz, "normal": p.normal, "scaling": p.scaling, "rotation": p.rotation, "opacity": p.opacity,
             "shs_dc": p.features_dc, "shs_rest": p.features_rest}
    for k, t in names.items():
        chk("autograd g_" + k, t.grad, z["g_" + k], 2e-3, 1e-9)
    fused = FusedStage1Step(_params(z, False))
    fused.iteration = it
    fo = fused.forward_backward(cam, bg, gt, mask)
    torch.cuda.synchronize()
    chk("fused render", fo[2], z["map_render"], 2e-5, 1e-5)
    chk("fused loss", fused.loss().reshape(1), np.array([z["loss"]], np.float32), 1e-5)
    g = fused.grads
    for k in ("xyz", "normal", "scaling", "rotation", "opacity"):
        chk("fused g_" + k, g[k], z["g_" + k], 2e-3, 1e-9)
    chk("fused g_shs", g["shs"], np.concatenate([z["g_shs_dc"], z["g_shs_rest"]], 1), 2e-3, 1e-9)
    chk("viewspace gradient", fused.viewspace_grad, z["g_viewspace"], 2e-3, 1e-9)
    chk.done()
