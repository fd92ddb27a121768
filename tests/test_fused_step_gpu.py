"""Fused stage-2 iteration (csrc/stage2_glue.hip + fused_step.py) against the plain-PyTorch/autograd restatement of the
reference's Python glue (train_step.Stage2Step: gaussian_renderer/neilf.py:15-318, scene/gaussian_model.py:183-232) and
torch.optim.Adam.  fp32 tolerances: the two paths evaluate the same formulas in different association orders, the
reductions inside the HIP ops use float atomics."""
import numpy as np
import pytest
import torch

from tests.helpers import report

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(P=4000, res=160, K=8, seed=3, weights=None, lrs=None):
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams, render_stage1
    from relightable3dgaussian_amd.train_step import Stage2Step
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    torch.manual_seed(1234)          # the teacher perturbations below use the global (device) generator
    scene = syn.make_scene(P=P, seed=seed, stage2=True, scale_log_mean=-3.2)
    cam = syn.orbit_cameras(8, width=res, height=res)[1].to(DEV)
    bg = torch.tensor([1.0, 0.6, 0.3], device=DEV)
    params = GaussianParams(scene, DEV, True)
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=P, seed=seed, stage2=False, scale_log_mean=-3.2), DEV, False)
        teacher.features_dc.add_(0.2 * torch.randn_like(teacher.features_dc))
        gt = render_stage1(teacher, cam, bg)[2].clone()
    ref = Stage2Step(params, scene, DEV, K, loss_weights=weights)
    fused = FusedStage2Step(params, K, loss_weights=weights, lrs=lrs)
    # identical visibility caches (the BVH inputs differ in the last bits between the two activation paths)
    fused.visibility, fused.incident_dirs, fused.incident_areas = ref.visibility, ref.incident_dirs, ref.incident_areas
    return params, ref, fused, cam, bg, gt


# default = the objective of script/run_nerf.sh:20-39 (only the render and pbr maps carry a loss: 3 active feature channels);
# {"normal": 0.01} adds the normal_render_depth term (6 active channels)
def _object_mask(res):
    """A soft object mask with values strictly between 0 and 1 somewhere (Camera.image_mask is the image's alpha channel)."""
    yy, xx = torch.meshgrid(torch.linspace(-1, 1, res), torch.linspace(-1, 1, res), indexing="ij")
    return (1.3 - 1.5 * (xx * xx + yy * yy).sqrt()).clamp(0, 1)[None].contiguous().to(DEV)


# default = the objective of script/run_nerf.sh:20-39 (only the render and pbr maps carry a loss: 3 active feature channels);
# {"normal": 0.01} adds the normal_render_depth term (6 active channels); run_syn4 = script/run_syn4.sh:34-36 / run_dtu.sh: the
# three edge-aware smoothness terms (13 active channels) -- alone and on top of the normal term (the light term's guide is
# the rendered normal: both write the normal maps' gradient), each with a non-trivial object mask
_SYN4 = dict(base_color_smooth=1.0, roughness_smooth=0.5, light_smooth=1.0)
@pytest.mark.parametrize("weights,masked", [(None, False), ({"normal": 0.01}, False), ({"normal": 0.01}, True), (_SYN4, True),
                                            (dict(_SYN4, normal=0.01), True), ({"roughness_smooth": 0.5}, False)],
                         ids=["run_nerf_stage2", "with_normal_term", "normal_term_masked", "run_syn4_objective",
                              "run_syn4_plus_normal", "one_smoothness_term"])
def test_fused_forward_backward_matches_autograd(weights, masked):
    params, ref, fused, cam, bg, gt = _setup(weights=weights)
    mask = _object_mask(gt.shape[-1]) if masked else None
    loss_ref, outs_ref = ref(cam, bg, gt, mask)
    loss_ref.backward()
    outs = fused.forward_backward(cam, bg, gt, image_mask=mask)
    torch.cuda.synchronize()
    msgs, ok_all = [], True

    def chk(name, got, want, rtol, atol=0.0, outliers=0.0):
        """`outliers`: fraction of elements allowed beyond rtol (each still within 20 % of the array's scale): the two
        activation paths differ in the last bit of exp()/sigmoid(), which flips a few borderline alpha >= 1/255
        decisions inside the rasterizer and moves the gradients of the handful of Gaussians involved."""
        nonlocal ok_all
        ok, msg = report(name, got, want, rtol, atol)
        if not ok and outliers > 0.0:
            gotn, wantn = got.detach().cpu().double(), want.detach().cpu().double()
            err = (gotn - wantn).abs()
            scale = wantn.abs().max()
            frac = float((err > atol + rtol * scale).double().mean())
            ok = frac <= outliers and float(err.max()) <= atol + 0.2 * scale
            msg += "  [outlier fraction %.2e allowed %.1e -> %s]" % (frac, outliers, "ok" if ok else "FAIL")
        msgs.append(msg)
        ok_all &= ok

    assert outs[0] == outs_ref[0]
    chk("image", outs[2], outs_ref[2], 1e-5, 1e-6)
    chk("feature", outs[5], outs_ref[5], 1e-4, 1e-6)
    chk("loss", fused.loss().reshape(1), loss_ref.detach().reshape(1), 1e-5)
    g = fused.grads
    chk("g xyz", g["xyz"], params.xyz.grad, 2e-4, outliers=4e-3)
    chk("g normal", g["normal"], params.normal.grad, 2e-4, outliers=4e-3)
    chk("g scaling", g["scaling"], params.scaling.grad, 2e-4, outliers=4e-3)
    chk("g rotation", g["rotation"], params.rotation.grad, 2e-4, outliers=4e-3)
    chk("g opacity", g["opacity"], params.opacity.grad, 2e-4, outliers=4e-3)
    chk("g shs", g["shs"], torch.cat([params.features_dc.grad, params.features_rest.grad], 1), 2e-4, outliers=4e-3)
    chk("g base_color", g["base_color"], params.base_color.grad, 2e-4, outliers=4e-3)
    chk("g roughness", g["roughness"], params.roughness.grad, 2e-4, outliers=4e-3)
    chk("g incidents", g["incidents"], torch.cat([params.incidents_dc.grad, params.incidents_rest.grad], 1), 2e-4,
        outliers=4e-3)
    chk("g env", g["env"], params.env.grad, 1e-3)        # sum over all Gaussians: inherits the outliers above
    print("\n".join(msgs))
    assert ok_all, "\n".join(msgs)
    for k in ("xyz", "shs", "incidents", "base_color"):
        assert float(g[k].abs().max()) > 0, k


def test_frozen_geometry_iteration_equals_the_full_one_on_the_groups_that_train():
    """script/run_syn4.sh:27-33 / run_dtu.sh: learning rate 0 on positions, normals, SH colour, opacity, scaling, rotation.
    The frozen-geometry iteration (feature-only tile backward, no per-Gaussian geometry backward, no geometry chain rule, one
    Adam launch over four groups) must produce the same loss and the same gradients for base colour, roughness, incident
    light and the environment texture as the full iteration, leave the frozen groups' gradients at zero -- and after a
    few steps the frozen parameters are bit-identical to where they started while the trained ones moved exactly as in a
    full iteration whose frozen groups have learning rate 0 in Adam only."""
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    frozen = dict(xyz=0.0, normal=0.0, scaling=0.0, rotation=0.0, opacity=0.0, shs=0.0, shs_rest=0.0, base_color=0.01,
                  roughness=0.01, incidents=0.001, incidents_rest=0.0001, env=0.1)
    params, ref, full, cam, bg, gt = _setup(weights=_SYN4)
    mask = _object_mask(gt.shape[-1])
    fr = FusedStage2Step(params, ref.K, loss_weights=_SYN4, lrs=frozen)
    assert fr.frozen_geometry and not full.frozen_geometry
    fr.visibility, fr.incident_dirs, fr.incident_areas = ref.visibility, ref.incident_dirs, ref.incident_areas
    full.forward_backward(cam, bg, gt, image_mask=mask)
    fr.forward_backward(cam, bg, gt, image_mask=mask)
    torch.cuda.synchronize()
    assert abs(float(fr.loss()) - float(full.loss())) <= 1e-6 * abs(float(full.loss()))
    for k in ("base_color", "roughness", "incidents", "env"):
        ok, msg = report("frozen g " + k, fr.grads[k], full.grads[k], 2e-5, 1e-9)       # (order of the float atomics only)
        assert ok, msg
        assert float(fr.grads[k].abs().max()) > 0
    for k in ("xyz", "normal", "scaling", "rotation", "opacity", "shs"):
        assert float(fr.grads[k].abs().max()) == 0.0, k
    # the full iteration with the SAME rates (Adam multiplies the frozen groups' updates by 0) as the trajectory to match
    slow = FusedStage2Step(params, ref.K, loss_weights=_SYN4, lrs=frozen)
    slow.frozen, slow.frozen_geometry = set(), False           # (force the full path; buckets are only used under DP)
    slow._groups_a, slow._groups_c, slow._groups_b = (5,), (0, 1, 2, 3, 4, 6, 7, 9), (8,)
    slow.visibility, slow.incident_dirs, slow.incident_areas = ref.visibility, ref.incident_dirs, ref.incident_areas
    fr2 = FusedStage2Step(params, ref.K, loss_weights=_SYN4, lrs=frozen)
    fr2.visibility, fr2.incident_dirs, fr2.incident_areas = ref.visibility, ref.incident_dirs, ref.incident_areas
    start = {k: getattr(fr2, k).clone() for k in ("xyz", "normal", "scaling", "rotation", "opacity", "shs")}
    for it in range(4):
        slow(cam, bg, gt, image_mask=mask)
        fr2(cam, bg, gt, image_mask=mask)
    torch.cuda.synchronize()
    for k, v in start.items():
        assert torch.equal(getattr(fr2, k), v), k
    for k in ("base_color", "roughness", "incidents", "env"):
        ok, msg = report("frozen param " + k, getattr(fr2, k), getattr(slow, k), 1e-4, 1e-6)
        assert ok, msg
        assert not torch.equal(getattr(fr2, k), getattr(params, k).detach() if k != "incidents" else
                               torch.cat([params.incidents_dc, params.incidents_rest], 1).detach())
    assert fr2.opt.step_count == 4 and fr2.poll_overflow() == 0


def test_fused_adam_matches_torch_adam():
    from relightable3dgaussian_amd.fused_step import FusedAdam
    gen = torch.Generator().manual_seed(0)
    shapes = [(1000, 3), (777, 16, 3), (5, 1), (1, 16, 32, 3), (4097,)]
    ps = [torch.randn(*s, generator=gen).to(DEV) for s in shapes]
    lrs = [1e-2, 2e-3, 1e-3, 5e-3, 1e-4]
    # torch reference: the [777,16,3] tensor is split into dc / rest parameter groups with different rates
    tp = [p.clone().requires_grad_(True) for p in ps]
    dc, rest = tp[1][:, :1].detach().clone().requires_grad_(True), tp[1][:, 1:].detach().clone().requires_grad_(True)
    groups = [dict(params=[tp[0]], lr=lrs[0]), dict(params=[dc], lr=lrs[1]), dict(params=[rest], lr=lrs[1] / 20),
              dict(params=[tp[2]], lr=lrs[2]), dict(params=[tp[3]], lr=lrs[3]), dict(params=[tp[4]], lr=lrs[4])]
    topt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
    mine = [p.clone() for p in ps]
    fopt = FusedAdam([dict(param=mine[0], lr=lrs[0]),
                      dict(param=mine[1], lr=lrs[1], lr_tail=lrs[1] / 20, period=48, split=3),
                      dict(param=mine[2], lr=lrs[2]), dict(param=mine[3], lr=lrs[3]), dict(param=mine[4], lr=lrs[4])],
                     eps=1e-15)
    for step in range(4):
        gs = [torch.randn(*s, generator=gen).to(DEV) * (0.1 + step) for s in shapes]
        tp[0].grad, tp[2].grad, tp[3].grad, tp[4].grad = gs[0], gs[2], gs[3], gs[4]
        dc.grad, rest.grad = gs[1][:, :1].contiguous(), gs[1][:, 1:].contiguous()
        topt.step()
        fopt.step(gs)
    torch.cuda.synchronize()
    want = [tp[0], torch.cat([dc, rest], 1), tp[2], tp[3], tp[4]]
    for i, (a, b) in enumerate(zip(mine, want)):
        ok, msg = report("adam p%d" % i, a, b.detach(), 2e-6, 1e-7)
        assert ok, msg


def test_fused_step_trains():
    """A few full fused iterations reduce the loss and keep everything finite."""
    params, ref, fused, cam, bg, gt = _setup(P=3000, res=128, K=8, seed=5)
    losses = []
    for it in range(6):
        fused(cam, bg, gt)
        losses.append(float(fused.loss()))
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
    for k in ("xyz", "shs", "incidents", "env"):
        assert torch.isfinite(getattr(fused, k)).all(), k


def test_fused_training_tracks_autograd_training_psnr():
    """North-star proxy ("PSNR within 0.1 dB of the reference pipeline"): 40 training iterations over 4 views with the
    fused iteration and with the autograd restatement of the reference's glue + torch.optim.Adam, from the same
    initialisation, end at the same view-averaged PSNR (|diff| < 0.1 dB)."""
    import math
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams, render_stage1
    from relightable3dgaussian_amd.train_step import Stage2Step
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    P, res, K, lr = 4000, 128, 8, 2e-3
    torch.manual_seed(1234)
    scene = syn.make_scene(P=P, seed=21, stage2=True, scale_log_mean=-3.0)
    cams = [c.to(DEV) for c in syn.orbit_cameras(8, width=res, height=res)[:4]]
    bg = torch.ones(3, device=DEV)
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=P, seed=21, stage2=False, scale_log_mean=-3.0), DEV, False)
        teacher.features_dc.add_(0.3 * torch.randn_like(teacher.features_dc))
        gts = [render_stage1(teacher, c, bg)[2].clone() for c in cams]

    def psnr(img, gt):
        return -10.0 * math.log10(float(((img - gt) ** 2).mean()))

    pa = GaussianParams(scene, DEV, True)
    ref = Stage2Step(pa, scene, DEV, K)
    opt = torch.optim.Adam(pa.parameters(), lr=lr, eps=1e-15)
    pb = GaussianParams(scene, DEV, True)
    fused = FusedStage2Step(pb, K, lr=lr)
    fused.visibility, fused.incident_dirs, fused.incident_areas = ref.visibility, ref.incident_dirs, ref.incident_areas
    first = None
    for it in range(40):
        i = it % 4
        loss, outs = ref(cams[i], bg, gts[i])
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=False)
        fused(cams[i], bg, gts[i])
        if first is None:
            first = psnr(outs[2].detach(), gts[i])
    with torch.no_grad():
        pa_db, pb_db = [], []
        for i in range(4):
            pa_db.append(psnr(ref.render(cams[i], bg)[0][2], gts[i]))
            pb_db.append(psnr(fused.forward_backward(cams[i], bg, gts[i])[2], gts[i]))
    print("PSNR autograd %s | fused %s" % (["%.2f" % v for v in pa_db], ["%.2f" % v for v in pb_db]))
    # the two trajectories differ by fp32 rounding amplified over 40 Adam steps: the view average is the stable quantity
    assert abs(sum(pa_db) / 4 - sum(pb_db) / 4) < 0.1, (pa_db, pb_db)
    assert all(abs(a - b) < 0.5 for a, b in zip(pa_db, pb_db)), (pa_db, pb_db)
    assert min(pb_db) > first - 1.0


@pytest.mark.parametrize("with_mask,iteration,weights", [(False, 0, None), (True, 7000, None),
                                                         (True, 3000, dict(depth_var=0.0, normal_smooth=0.05))],
                         ids=["run_nerf_stage1", "object_mask_iter7000", "no_depth_var"])
def test_fused_stage1_matches_autograd(with_mask, iteration, weights):
    """Stage-1 fused iteration vs bench_core.render_stage1 + train_step.stage1_loss (the reference's calculate_loss,
    render.py:137-223, restated in PyTorch) under autograd: L1+SSIM, mask entropy, normal_render_depth, edge-aware normal
    smoothness (Sobel stencil and its adjoint), depth variance with its iteration schedule."""
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams, render_stage1, loss_stage1
    from relightable3dgaussian_amd.fused_step import FusedStage1Step
    P, res = 4000, 160
    torch.manual_seed(1234)
    scene = syn.make_scene(P=P, seed=4, stage2=False, scale_log_mean=-3.2)
    cam = syn.orbit_cameras(8, width=res, height=res)[2].to(DEV)
    bg = torch.tensor([1.0, 0.7, 0.2], device=DEV)
    params = GaussianParams(scene, DEV, False)
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=P, seed=4, stage2=False, scale_log_mean=-3.2), DEV, False)
        teacher.features_dc.add_(0.2 * torch.randn_like(teacher.features_dc))
        gt = render_stage1(teacher, cam, bg)[2].clone()
    mask = None
    if with_mask:                      # object mask with a soft edge (the datasets' alpha channel is not binary)
        yy, xx = torch.meshgrid(torch.linspace(-1, 1, res, device=DEV), torch.linspace(-1, 1, res, device=DEV), indexing="ij")
        mask = (1.2 - 2.0 * (xx * xx + yy * yy).sqrt()).clamp(0, 1)[None].contiguous()
    outs_ref = render_stage1(params, cam, bg)
    loss_ref = loss_stage1(outs_ref, gt, mask, weights, iteration)
    loss_ref.backward()
    fused = FusedStage1Step(params, loss_weights=weights)
    fused.iteration = iteration
    outs = fused.forward_backward(cam, bg, gt, mask)
    torch.cuda.synchronize()
    assert outs[0] == outs_ref[0]
    msgs, ok_all = [], True

    def chk(name, got, want, rtol, atol=0.0, outliers=0.0):
        nonlocal ok_all
        ok, msg = report(name, got, want, rtol, atol)
        if not ok and outliers > 0.0:
            err = (got.detach().cpu().double() - want.detach().cpu().double()).abs()
            scale = float(want.abs().max())
            frac = float((err > atol + rtol * scale).double().mean())
            ok = frac <= outliers and float(err.max()) <= atol + 0.2 * scale
            msg += "  [outlier fraction %.2e -> %s]" % (frac, "ok" if ok else "FAIL")
        msgs.append(msg)
        ok_all &= ok

    chk("image", outs[2], outs_ref[2], 1e-5, 1e-6)
    chk("loss", fused.loss().reshape(1), loss_ref.detach().reshape(1), 1e-5)
    g = fused.grads
    for k in ("xyz", "normal", "scaling", "rotation", "opacity"):
        chk("g " + k, g[k], getattr(params, k).grad, 2e-4, outliers=4e-3)
    chk("g shs", g["shs"], torch.cat([params.features_dc.grad, params.features_rest.grad], 1), 2e-4, outliers=4e-3)
    assert ok_all, "\n".join(msgs)
    l0 = float(fused.loss())
    for _ in range(5):
        fused.forward_backward(cam, bg, gt, mask)
        fused.optimizer_step()
    assert float(fused.loss()) < l0


def test_fused_stage1_training_with_densification():
    """train.py:158-175 on the fused stage-1 iteration: statistics every step (checked against a torch restatement of
    add_densification_stats on this step's tensors), densify_and_prune / reset_opacity in between; the Adam moments of
    surviving rows travel with them and training goes on at the new size."""
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams, render_stage1
    from relightable3dgaussian_amd.fused_step import FusedStage1Step
    P, res = 6000, 160
    torch.manual_seed(99)
    scene = syn.make_scene(P=P, seed=6, stage2=False, scale_log_mean=-3.2)
    cams = [c.to(DEV) for c in syn.orbit_cameras(8, width=res, height=res)[:3]]
    bg = torch.tensor([1.0, 1.0, 1.0], device=DEV)
    params = GaussianParams(scene, DEV, False)
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=P, seed=6, stage2=False, scale_log_mean=-3.2), DEV, False)
        teacher.features_dc.add_(0.2 * torch.randn_like(teacher.features_dc))
        gts = [render_stage1(teacher, c, bg)[2].clone() for c in cams]
    step = FusedStage1Step(params, lr=2e-3)
    step.enable_densification()
    want = dict(xyz=torch.zeros(P, device=DEV), normal=torch.zeros(P, device=DEV), denom=torch.zeros(P, device=DEV),
                weights=torch.zeros(P, device=DEV), radii=torch.zeros(P, device=DEV))
    for i in range(6):
        outs = step.forward_backward(cams[i % 3], bg, gts[i % 3])
        radii, weights = outs[9], outs[8]
        vis = radii > 0
        want["weights"] += weights[:, 0]
        want["xyz"][vis] += step.viewspace_grad[vis, :2].norm(dim=-1)
        want["normal"][vis] += step.grads["normal"][vis].norm(dim=-1)
        want["denom"][vis] += 1
        want["radii"][vis] = torch.max(want["radii"][vis], radii[vis].float())
        step.optimizer_step()
    st = step.stats
    for got, ref in ((st.xyz_gradient_accum, want["xyz"]), (st.normal_gradient_accum, want["normal"]),
                     (st.denom, want["denom"]), (st.weights_accum, want["weights"]), (st.max_radii2D, want["radii"])):
        torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-12)
    assert float(st.denom.max()) > 0 and float(st.xyz_gradient_accum.max()) > 0
    # thresholds chosen from the statistics so that both branches fire
    mean_grad = (st.xyz_gradient_accum / st.denom.clamp_min(1)).cpu()
    thr = float(mean_grad[mean_grad > 0].median())
    exp_avg_before = step.opt.groups[0]["exp_avg"].clone()
    steps_before = step.opt.step_count
    gen = torch.Generator(device=DEV).manual_seed(3)
    info = step.densify_and_prune(thr, 0.005, 2.6, 20, 1e9, percent_dense=0.01, generator=gen)
    assert info["cloned"] > 0 and info["split"] > 0
    assert step.P == info["rows_out"] == step.xyz.shape[0] == step.opt.groups[0]["exp_avg"].shape[0] != P
    assert step.grads["shs"].shape == (step.P, 16, 3) and step.stats.P == step.P
    assert step.opt.step_count == steps_before
    kept = info["kind"] == -1
    src = info["src_row"][kept].long()
    assert torch.equal(step.opt.groups[0]["exp_avg"][:int(kept.sum())], exp_avg_before[src])
    assert float(step.opt.groups[0]["exp_avg"][int(kept.sum()):].abs().max()) == 0.0
    assert float(step.stats._slab.abs().sum()) == 0.0
    for i in range(4):
        step(cams[i % 3], bg, gts[i % 3])
    assert np.isfinite(float(step.loss()))
    assert float(step.stats.denom.max()) > 0
    P1 = step.P
    info = step.prune(0.005, 2.6, 20)
    assert 0 < step.P == info["rows_out"] <= P1 and step.stats.P == step.P
    step(cams[0], bg, gts[0])
    assert np.isfinite(float(step.loss()))
    step.reset_opacity()
    assert float(torch.sigmoid(step.opacity).max()) <= 0.01 + 1e-6
    g = step.opt.groups[step._opt_order.index("opacity")]
    assert float(g["exp_avg"].abs().max()) == 0.0 and float(g["exp_avg_sq"].abs().max()) == 0.0
    step(cams[1], bg, gts[1])
    assert np.isfinite(float(step.loss()))


@pytest.mark.parametrize("H,W", [(64, 64), (37, 50), (16, 200), (5, 7)])
def test_ssim_kernels_match_reference_formula(H, W):
    """r3dg_ssim_forward/backward vs the conv2d restatement of utils/loss_utils.py:20-63 under autograd."""
    import ctypes as C
    from relightable3dgaussian_amd import _lib
    from relightable3dgaussian_amd.train_step import ssim
    L = _lib.lib()
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.rand(3, H, W, generator=g).to(DEV)
    y = (x.cpu() * 0.7 + 0.3 * torch.rand(3, H, W, generator=g)).to(DEV)
    xr = x.clone().requires_grad_(True)
    val = ssim(xr, y)
    val.backward()
    part = torch.empty(3, 3, H, W, device=DEV)
    total = torch.zeros(32, device=DEV)              # R3DG_SUM_SLOTS floats per accumulator
    grad = torch.empty(3, H, W, device=DEV)
    st = _lib.current_stream()
    _lib.check(L.r3dg_ssim_forward(st, W, H, 3, x.data_ptr(), y.data_ptr(), part.data_ptr(), total.data_ptr()), "f")
    _lib.check(L.r3dg_ssim_backward(st, W, H, 3, x.data_ptr(), y.data_ptr(), part.data_ptr(), 1.0 / (3 * H * W),
                                    grad.data_ptr()), "b")
    torch.cuda.synchronize()
    assert abs(float(total.sum()) / (3 * H * W) - float(val)) < 2e-6
    ok, msg = report("ssim grad", grad, xr.grad, 2e-4, 1e-9)
    assert ok, msg


def test_bounded_iterations_equal_two_phase_iterations():
    """FusedStage2Step(bounded=True) -- from the second iteration on the rasterizer forward runs without the host
    read-back of num_rendered -- trains like bounded=False: same loss trajectory and parameters up to the order of the
    float atomics."""
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    P, res, K = 4000, 128, 8
    runs = {}
    for bounded in (False, True):
        params, ref, fused, cam, bg, gt = _setup(P=P, res=res, K=K, seed=11)
        step = FusedStage2Step(params, K, bounded=bounded)
        step.visibility, step.incident_dirs, step.incident_areas = ref.visibility, ref.incident_dirs, ref.incident_areas
        losses = []
        for it in range(5):
            outs = step(cam, bg, gt)
            losses.append(float(step.loss()))
        assert step.dropped_steps == 0
        counts = step.rendered_counts(5)
        assert len(counts) == 5 and all(c > 0 for c in counts)
        if bounded:
            assert outs[0] == step._capacity and outs[0] >= 2 * counts[0]
        else:
            assert outs[0] == counts[-1]
        runs[bounded] = (losses, step.xyz.clone(), step.shs.clone(), step.incidents.clone(), counts)
    la, lb = runs[False][0], runs[True][0]
    assert np.allclose(la, lb, rtol=2e-5), (la, lb)
    assert runs[False][4] == runs[True][4], "num_rendered differs"
    for i in (1, 2, 3):
        ok, msg = report("param %d" % i, runs[True][i], runs[False][i], 1e-4, 1e-6)
        assert ok, msg


def test_every_schedule_of_the_whole_iteration_trains_alike(monkeypatch):
    """A whole iteration picks one of several schedules for its two large parameter groups: the SH group's Adam under the shading
    backward or with the others (R3DG_EARLY_ADAM: the size rule takes the second above a million Gaussians), the incident-light
    chain as one kernel, as three launches, or not at all (round 6: also WITHOUT the early Adam, what a 2M-Gaussian scene runs).
    Same losses and parameters after five iterations up to the order of the float atomics and the chain kernel's fast-math Adam."""
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    P, res, K = 4000, 128, 16
    settings = {"default": {}, "no early Adam, chain kernel": {"R3DG_EARLY_ADAM": "0"},
                "no early Adam, no chain": {"R3DG_EARLY_ADAM": "0", "R3DG_CHAIN_WITHOUT_EARLY_ADAM": "0"},
                "early Adam, three launches": {"R3DG_INCIDENT_CHAIN_KERNEL": "0"}}
    runs = {}
    for name, env in settings.items():
        for k in ("R3DG_EARLY_ADAM", "R3DG_CHAIN_WITHOUT_EARLY_ADAM", "R3DG_INCIDENT_CHAIN_KERNEL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        params, ref, fused, cam, bg, gt = _setup(P=P, res=res, K=K, seed=11)
        step = FusedStage2Step(params, K, lr=1e-3)
        losses = []
        for it in range(5):
            step(cam, bg, gt)
            losses.append(float(step.loss()))
        assert step._frs is not None, "the fixed-ray-set path must be the one under test"
        if name == "no early Adam, chain kernel":
            assert step._pre_rotated is not None, "the chain kernel did not run"
        runs[name] = (losses, step.shs.clone(), step.incidents.clone(), step.base_color.clone(), step.xyz.clone())
    base = runs["default"]
    for name, r in runs.items():
        assert np.allclose(r[0], base[0], rtol=2e-5), (name, r[0], base[0])
        for i in (1, 2, 3, 4):
            ok, msg = report("%s: param %d" % (name, i), r[i], base[i], 1e-4, 1e-6)
            assert ok, msg


def test_switching_schedules_between_iterations_keeps_the_rotated_coefficients_current(monkeypatch):
    """The chain kernel leaves the rotation of the NEW incident-light coefficients behind for the next forward (`_pre_rotated`).  An
    iteration that updates the coefficients through the plain Adam launch instead must invalidate that record -- the kernel writes
    through the raw pointer, the tensor's version counter does not move -- or the iteration after it shades with coefficients that
    are one update old (a latent bug until round 6: bench.py's one-stream pass toggled schedules, nothing compared results across
    the toggle).  Alternating schedules must train like one schedule throughout."""
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    P, res, K = 4000, 128, 16
    plain = {"R3DG_EARLY_ADAM": "0", "R3DG_CHAIN_WITHOUT_EARLY_ADAM": "0"}
    runs = {}
    for name, pattern in (("default", [{}] * 6), ("alternating", [{}, plain, {}, plain, plain, {}])):
        params, ref, fused, cam, bg, gt = _setup(P=P, res=res, K=K, seed=11)
        step = FusedStage2Step(params, K, lr=2e-3)
        losses = []
        for env in pattern:
            for k in plain:
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            step(cam, bg, gt)
            losses.append(float(step.loss()))
        runs[name] = (losses, step.incidents.clone(), step.base_color.clone())
    for k in plain:
        monkeypatch.delenv(k, raising=False)
    assert np.allclose(runs["alternating"][0], runs["default"][0], rtol=2e-5), (runs["alternating"][0], runs["default"][0])
    for i in (1, 2):
        ok, msg = report("param %d" % i, runs["alternating"][i], runs["default"][i], 1e-4, 1e-6)
        assert ok, msg


def test_feature_rows_written_in_place_equal_the_packed_rows(monkeypatch):
    """Without r3dg_stage2_pack_features (the default when the fixed-ray-set kernels run): r3dg_stage2_activate writes the nine
    columns of the [P,16] feature rows that do not wait for the shading integral, r3dg_shade_frs_forward (main and listed
    kernels) the other seven, and r3dg_stage2_unpack_gradients adds the light-smoothness sum.  The rows are the packed rows bit
    for bit (neilf.py:115-122), the loss trajectory and the first iteration's gradients the same up to the order of float sums."""
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    P, res, K = 4000, 128, 8
    runs = {}
    for direct in ("0", "1"):
        monkeypatch.setenv("R3DG_DIRECT_ROWS", direct)
        params, ref, fused, cam, bg, gt = _setup(P=P, res=res, K=K, seed=11, weights={"light": 0.01})
        step = FusedStage2Step(params, K, loss_weights={"light": 0.01})      # (its own visibility update: the Fibonacci ray set)
        step.features.fill_(float("nan"))                      # every element has to be written by somebody
        step(cam, bg, gt)
        assert step._frs is not None and step._frs.n_invalid > 0, "the test wants the rotated path with listed Gaussians"
        rows = step.features.clone()
        grads = {k: step.grads[k].clone() for k in ("incidents", "base_color", "roughness", "xyz", "env")}
        losses = [float(step.loss())]
        for it in range(3):
            step(cam, bg, gt)
            losses.append(float(step.loss()))
        runs[direct] = (rows, losses, grads)
    assert torch.equal(runs["0"][0], runs["1"][0]), "feature rows differ: max %g" % float((runs["0"][0] - runs["1"][0]).abs().max())
    assert np.allclose(runs["0"][1], runs["1"][1], rtol=2e-6), (runs["0"][1], runs["1"][1])
    # (gradients, not parameters: Adam turns a gradient of 1e-14 -- the noise floor of the float atomics, which differs between
    # two runs of the SAME configuration -- into a step of the size of the learning rate)
    for k, g0 in runs["0"][2].items():
        g1 = runs["1"][2][k]
        tol = 1e-4 * float(g0.abs().max()) + 1e-12
        assert float((g0 - g1).abs().max()) <= tol, (k, float((g0 - g1).abs().max()), tol)


def test_folded_launches_equal_the_launches_they_replace():
    """Round 5: r3dg_stage2_activate_with (softplus of the texture + a zero fill as extra workgroups), r3dg_stage2_activate_backward_with
    (the texture's chain rule as extra workgroups) and r3dg_stage2_normals_srgb (the rasterizer forward's pseudo-normal pass + the
    sRGB-mapped PBR image, pixel by pixel) against the separate launches / torch: same bits from the shared device code, the softplus
    within an ulp of torch.nn.functional.softplus (direct_light_map.py:18-23)."""
    import torch.nn.functional as F
    from relightable3dgaussian_amd import _lib, rasterizer_ops
    params, ref, fused, cam, bg, gt = _setup(P=3001, res=150, K=8, seed=5)
    L = _lib.lib()
    S = lambda: _lib.current_stream()
    P = fused.P
    # ---- activations + side jobs
    with torch.no_grad():
        fused.env.copy_(torch.linspace(-30.0, 30.0, fused.env.numel(), device=DEV).view_as(fused.env))     # both branches of softplus
    fused.refresh_activations(cam)
    plain = [t.clone() for t in (fused.a_scales, fused.a_rot, fused.a_opacity, fused.a_normal, fused.a_base, fused.a_rough,
                                 fused.a_viewdirs, fused.features)]
    for t in (fused.a_scales, fused.a_rot, fused.a_opacity, fused.a_normal, fused.a_base, fused.a_rough, fused.a_viewdirs):
        t.fill_(float("nan"))
    env_out = torch.full(tuple(fused.env.shape[1:]), float("nan"), device=DEV)
    junk = torch.full((1000,), 7.0, device=DEV)
    fused.refresh_activations(cam, env_out=env_out, zero=junk[:777])
    after = (fused.a_scales, fused.a_rot, fused.a_opacity, fused.a_normal, fused.a_base, fused.a_rough, fused.a_viewdirs, fused.features)
    for a, b in zip(plain, after):
        assert torch.equal(a, b)
    want = F.softplus(fused.env)[0]
    assert float(((env_out - want).abs() / want.abs().clamp_min(1e-30)).max()) < 4e-7
    assert float(junk[:777].abs().max()) == 0.0 and float(junk[777:].min()) == 7.0
    # ---- chain rule + the texture's
    fused.forward_backward(cam, bg, gt)
    torch.cuda.synchronize()
    He, We = fused.env.shape[1], fused.env.shape[2]
    env_c = F.softplus(fused.env)[0].contiguous()
    d_env = torch.randn_like(env_c)
    g = [torch.randn(P, n, device=DEV) for n in (16, 3, 1, 3, 3, 4, 1, 3)]       # dL_dfeatures, d_base, d_rough, d_view, dscales, drot, dop, dmeans
    vm, campos = cam.world_view_transform.contiguous(), cam.camera_center.contiguous()
    def run(entry, extra, outs, d_env_buf, g_env, tv):
        args = [S(), P, fused.xyz.data_ptr(), fused.scaling.data_ptr(), fused.rotation.data_ptr(), fused.opacity.data_ptr(),
                fused.normal.data_ptr(), fused.base_color.data_ptr(), fused.roughness.data_ptr(), vm.data_ptr(), campos.data_ptr()]
        args += [t.data_ptr() for t in g] + [o.data_ptr() for o in outs]
        _lib.check(getattr(L, entry)(*args, *extra), entry)
    shapes = ((P, 3), (P, 3), (P, 4), (P, 1), (P, 3), (P, 3), (P, 1))
    outs_a = [torch.full(s, float("nan"), device=DEV) for s in shapes]
    outs_b = [torch.full(s, float("nan"), device=DEV) for s in shapes]
    da, db = d_env.clone(), d_env.clone()
    ga, gb = torch.full_like(env_c, float("nan")), torch.full_like(env_c, float("nan"))
    tva, tvb = torch.zeros(32, device=DEV), torch.zeros(32, device=DEV)
    run("r3dg_stage2_activate_backward", (), outs_a, None, None, None)
    _lib.check(L.r3dg_stage2_env_backward(S(), He, We, fused.env.data_ptr(), env_c.data_ptr(), da.data_ptr(), 0.37, ga.data_ptr(),
                                          tva.data_ptr(), 1), "env_backward")
    run("r3dg_stage2_activate_backward_with", (He, We, fused.env.data_ptr(), env_c.data_ptr(), db.data_ptr(), 0.37, gb.data_ptr(),
                                                tvb.data_ptr(), 1), outs_b, None, None, None)
    for a, b in zip(outs_a, outs_b):
        assert torch.equal(a, b)
    assert torch.equal(ga, gb) and float(db.abs().max()) == 0.0 and float(da.abs().max()) == 0.0
    assert abs(float(tva.sum()) - float(tvb.sum())) <= 1e-6 * abs(float(tva.sum())) and float(tva.sum()) > 0
    # ---- pseudo normals + sRGB map
    empty = torch.Tensor([])
    fw = rasterizer_ops.rasterize_gaussians(
        bg, fused.xyz, fused.features, empty, fused.a_opacity, fused.a_scales, fused.a_rot, 1.0, empty, vm,
        cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.cx, cam.cy, cam.image_height, cam.image_width, fused.shs, 3, campos,
        False, True, False)
    R, n_contrib, image, opacity, depth, feature, pseudo_normal, sxyz = fw[:8]
    H, W = cam.image_height, cam.image_width
    srgb = torch.empty(3, H, W, device=DEV)
    _lib.check(L.r3dg_stage2_pbr_srgb(S(), W, H, opacity.data_ptr(), feature.data_ptr(), n_contrib.data_ptr(), bg.data_ptr(),
                                      srgb.data_ptr()), "pbr_srgb")
    n2, x2, s2 = (torch.full((3, H, W), float("nan"), device=DEV) for _ in range(3))
    _lib.check(L.r3dg_stage2_normals_srgb(S(), W, H, vm.data_ptr(), float(cam.tanfovx), float(cam.tanfovy), float(cam.cx),
                                          float(cam.cy), opacity.data_ptr(), depth.data_ptr(), n2.data_ptr(), x2.data_ptr(),
                                          feature.data_ptr(), n_contrib.data_ptr(), bg.data_ptr(), s2.data_ptr()), "normals_srgb")
    assert torch.equal(s2, srgb) and torch.equal(x2, sxyz)
    assert float((n2 - pseudo_normal).abs().max()) <= 1e-6 and float(pseudo_normal.abs().max()) > 0.5
    print("folded launches: normals max diff %.2e" % float((n2 - pseudo_normal).abs().max()))


def test_texture_resize_keeps_the_ray_set_and_rebuilds_only_its_lookup_records(monkeypatch):
    """only the 8-byte lookup records depend on the environment texture's size.  Swapping the texture for one of another
    size between two iterations must not regenerate and re-classify the P x K directions (FixedRaySet.try_build: a host read-back);
    the iteration after the swap equals the iteration of a step object that was built with the new texture."""
    from relightable3dgaussian_amd import shading_ops
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    params, ref, fused, cam, bg, gt = _setup(P=3000, res=128, K=16, seed=9)
    step = FusedStage2Step(params, 16)
    step(cam, bg, gt)
    assert step._frs is not None
    built = []
    orig = shading_ops.FixedRaySet.try_build
    monkeypatch.setattr(shading_ops.FixedRaySet, "try_build", classmethod(lambda cls, *a, **k: built.append(1) or orig(*a, **k)))
    ray_set = step._frs
    small = torch.nn.functional.interpolate(step.env.permute(0, 3, 1, 2), size=(8, 16), mode="bilinear").permute(0, 2, 3, 1).contiguous()
    fresh = FusedStage2Step(params, 16)
    for s_ in (step, fresh):
        s_.env = small.clone()
        s_.grads["env"] = torch.zeros_like(s_.env)
    # (the texture's Adam group follows the tensor: rebuild it for both the same way)
    for s_ in (step, fresh):
        g = s_.opt.groups[s_._opt_order.index("env")]
        g["param"], g["exp_avg"], g["exp_avg_sq"] = s_.env, torch.zeros_like(s_.env), torch.zeros_like(s_.env)
    fresh.xyz.copy_(step.xyz); fresh.normal.copy_(step.normal); fresh.scaling.copy_(step.scaling); fresh.rotation.copy_(step.rotation)
    fresh.opacity.copy_(step.opacity); fresh.shs.copy_(step.shs); fresh.base_color.copy_(step.base_color)
    fresh.roughness.copy_(step.roughness); fresh.incidents.copy_(step.incidents)
    fresh.visibility, fresh.incident_dirs, fresh.incident_areas = step.visibility, step.incident_dirs, step.incident_areas
    fresh._ray_normals = step._ray_normals
    n_before = len(built)
    step.forward_backward(cam, bg, gt)
    assert len(built) == n_before and step._frs is ray_set, "the ray set was rebuilt for a texture resize"
    fresh.forward_backward(cam, bg, gt)
    torch.cuda.synchronize()
    assert abs(float(step.loss()) - float(fresh.loss())) <= 1e-6 * abs(float(fresh.loss()))
    for k in ("env", "base_color", "incidents"):
        a, b = step.grads[k], fresh.grads[k]
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12, k


def test_iterations_leave_no_device_memory_to_the_garbage_collector():
    """A frame's scratch buffers (geometry / binning / image state) are freed by reference counting when the iteration is over:
    no reference cycle holds a device tensor (the resize callbacks used to be closures over the object that owns them and the
    buffers -- hundreds of MB per frame that only a generation-2 collection released)."""
    import gc
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    params, ref, fused, cam, bg, gt = _setup(P=4000, res=128, K=8, seed=11)
    step = FusedStage2Step(params, 8)
    for it in range(3):
        step(cam, bg, gt)
    del ref, fused
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    gc.set_debug(gc.DEBUG_SAVEALL)
    try:
        for it in range(6):
            step(cam, bg, gt)
        torch.cuda.synchronize()
        gc.collect()
        held = [o for o in gc.garbage if isinstance(o, torch.Tensor) and o.is_cuda]
        nbytes = sum(t.numel() * t.element_size() for t in held)
    finally:
        gc.garbage.clear()
        gc.set_debug(0)
        if was:
            gc.enable()
    assert not held, "%d device tensors (%.1f MB) were only reachable through reference cycles" % (len(held), nbytes / 2**20)


def test_clock_probe_reports_a_plausible_shader_clock():
    """r3dg_clock_probe (bench.py's `device_clock`): every wave of a device-filling FMA-only grid reports, and the ratio of the
    shader-clock counter to the constant-rate wall clock is a clock an MI355X can run at."""
    from relightable3dgaussian_amd import _lib
    ghz, waves = _lib.shader_clock_ghz(DEV, 2000)
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert waves == cus * 8 * 4
    assert 0.8 < ghz < 3.2, ghz


def test_bounded_iteration_that_overflows_is_dropped_not_applied():
    """A view that needs more instance slots than the bounded forward has: the iteration's Adam launches update nothing,
    poll_overflow() reports it, takes the step count back and doubles the capacity; the next iteration trains again."""
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    params, ref, fused, cam, bg, gt = _setup(P=4000, res=128, K=8, seed=12)
    step = FusedStage2Step(params, 8, bounded=True)
    step.visibility, step.incident_dirs, step.incident_areas = ref.visibility, ref.incident_dirs, ref.incident_areas
    step(cam, bg, gt)                                   # learns the count
    n = step.rendered_counts(1)[0]
    step._capacity = n - 5                              # the next view will not fit
    before = {k: getattr(step, k).clone() for k in ("xyz", "shs", "incidents", "env", "opacity")}
    moments = [g["exp_avg"].clone() for g in step.opt.groups]
    count_before = step.opt.step_count
    step(cam, bg, gt)
    torch.cuda.synchronize()
    for k, v in before.items():
        assert torch.equal(getattr(step, k), v), "%s was updated by a dropped iteration" % k
    for g, m in zip(step.opt.groups, moments):
        assert torch.equal(g["exp_avg"], m)
    assert step.poll_overflow() == 1 and step.dropped_steps == 1
    assert step.opt.step_count == count_before and step._capacity >= 2 * n
    step(cam, bg, gt)
    torch.cuda.synchronize()
    assert step.poll_overflow() == 0
    assert not torch.equal(step.xyz, before["xyz"])
    assert step.rendered_counts(1)[0] > 0


def test_bounded_iteration_that_overflows_is_replayed():
    """VERDICT r4 missing 5: the reference trains on every view (it sizes its binning state from the count it reads back,
    rasterizer_impl.cu:291).  A bounded iteration that overflows is dropped on the device; poll_overflow() says WHICH
    iterations were (their slots of the flag ring), and replay_dropped() trains on those views through the exact two-phase
    forward.  With a poll behind every iteration the result is the two-phase loop's: same parameters (up to the order of the
    float atomics), same Adam step count, nothing left dropped."""
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    params, ref, fused, cam, bg, gt = _setup(P=4000, res=128, K=8, seed=12)
    cams = [c.to(DEV) for c in syn.orbit_cameras(6, width=128, height=128)]
    a = FusedStage2Step(params, 8, bounded=True)
    b = FusedStage2Step(params, 8, bounded=False)
    for s_ in (a, b):
        s_.visibility, s_.incident_dirs, s_.incident_areas = ref.visibility, ref.incident_dirs, ref.incident_areas
    inputs = {}
    replayed = []
    for i in range(6):
        b(cams[i], bg, gt)
        if i in (2, 4):
            a._capacity = 1000                                      # this view will not fit
        inputs[a._iter + 1] = (cams[i], bg, gt)
        a(cams[i], bg, gt)

        def inputs_of(it):
            inputs[a._iter + 1] = inputs[it]
            return inputs[it]
        replayed += a.replay_dropped(inputs_of)
    a.flush()
    torch.cuda.synchronize()
    assert len(replayed) == 2 and a.dropped_steps == 0 and a.dropped_iterations == []
    assert a.opt.step_count == b.opt.step_count == 6
    for k in ("xyz", "scaling", "rotation", "opacity", "shs", "base_color", "roughness", "incidents", "env"):
        ok, msg = report(k, getattr(a, k), getattr(b, k), 2e-5, 1e-7)
        assert ok, msg
    assert a._capacity is not None and a._capacity > 1000            # re-learned from the replayed view's count


def test_bounded_stage1_iterations_equal_two_phase_iterations():
    """FusedStage1Step(bounded=True) vs bounded=False: same losses, same parameters (up to the order of float atomics), same
    densification statistics; a view that does not fit updates neither parameters nor statistics."""
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams, render_stage1
    from relightable3dgaussian_amd.fused_step import FusedStage1Step
    P, res = 5000, 128
    cam = syn.orbit_cameras(8, width=res, height=res)[3].to(DEV)
    bg = torch.tensor([0.2, 0.9, 0.4], device=DEV)
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=P, seed=21, stage2=False, scale_log_mean=-3.2), DEV, False)
        teacher.features_dc.add_(0.2 * torch.randn_like(teacher.features_dc))
        gt = render_stage1(teacher, cam, bg)[2].clone()
    runs = {}
    for bounded in (False, True):
        params = GaussianParams(syn.make_scene(P=P, seed=21, stage2=False, scale_log_mean=-3.2), DEV, False)
        step = FusedStage1Step(params, bounded=bounded)
        step.enable_densification()
        losses = []
        for it in range(5):
            outs = step(cam, bg, gt)
            losses.append(float(step.loss()))
        assert step.dropped_steps == 0
        counts = step.rendered_counts(5)
        assert (outs[0] == step._capacity) if bounded else (outs[0] == counts[-1])
        runs[bounded] = (losses, step.xyz.clone(), step.shs.clone(), step.stats.denom.clone(),
                         step.stats.xyz_gradient_accum.clone(), counts)
    assert np.allclose(runs[False][0], runs[True][0], rtol=2e-5), (runs[False][0], runs[True][0])
    assert runs[False][5] == runs[True][5]
    assert torch.equal(runs[False][3], runs[True][3])
    for i in (1, 2, 4):
        ok, msg = report("stage-1 tensor %d" % i, runs[True][i], runs[False][i], 1e-4, 1e-7)
        assert ok, msg
    # overflow: nothing moves
    step._capacity = runs[True][5][-1] - 3
    before = (step.xyz.clone(), step.shs.clone(), step.stats.denom.clone(), step.stats.weights_accum.clone())
    n_steps = step.opt.step_count
    step(cam, bg, gt)
    torch.cuda.synchronize()
    assert torch.equal(step.xyz, before[0]) and torch.equal(step.shs, before[1])
    assert torch.equal(step.stats.denom, before[2]) and torch.equal(step.stats.weights_accum, before[3])
    assert step.poll_overflow() == 1 and step.opt.step_count == n_steps and step._capacity >= 2 * runs[True][5][-1]
    step(cam, bg, gt)
    torch.cuda.synchronize()
    assert step.poll_overflow() == 0 and not torch.equal(step.xyz, before[0])


def test_option_contexts_keep_two_objects_apart():
    """include/r3dg_hip.h "option contexts" (r3dg_set_option alone is process-global state that two step
    objects race on).  A context's values are seen by launches of the thread that made it current, only where the context sets
    them, only while it is current; contexts nest; another thread is not affected.  Then two step objects with DIFFERENT
    instance-ordering formulations in one process, interleaved: each trains exactly as the same object does alone."""
    import threading
    from relightable3dgaussian_amd import _lib
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    assert _lib.get_option("CULL") == 1 and _lib.get_option("TILE_BINNING") == 2
    a, b = _lib.OptionContext(CULL=0), _lib.OptionContext(TILE_BINNING=0, RESERVE_CUS=8)
    with a:
        assert (_lib.get_option("CULL"), _lib.get_option("TILE_BINNING"), _lib.get_option("RESERVE_CUS")) == (0, 2, 0)
        with b:
            assert (_lib.get_option("CULL"), _lib.get_option("TILE_BINNING"), _lib.get_option("RESERVE_CUS")) == (1, 0, 8)
            seen = []
            t = threading.Thread(target=lambda: seen.append((_lib.get_option("TILE_BINNING"), _lib.get_option("RESERVE_CUS"))))
            t.start()
            t.join()
            assert seen == [(2, 0)], "another thread sees this thread's context"
        assert _lib.get_option("CULL") == 0 and _lib.get_option("TILE_BINNING") == 2
    assert _lib.get_option("CULL") == 1
    with pytest.raises(RuntimeError):
        a.set("CULL", 7)

    P, res, K = 4000, 128, 8

    def make(binning):
        params, ref, fused, cam, bg, gt = _setup(P=P, res=res, K=K, seed=11)
        step = FusedStage2Step(params, K)
        step._ctx.set("TILE_BINNING", binning)          # 0: the reference's global radix sort (two-phase forward), 2: direct binning
        step.visibility, step.incident_dirs, step.incident_areas = ref.visibility, ref.incident_dirs, ref.incident_areas
        return step, cam, bg, gt
    alone = {}
    for binning in (0, 2):
        step, cam, bg, gt = make(binning)
        for it in range(4):
            step(cam, bg, gt)
        alone[binning] = (float(step.loss()), step.xyz.clone(), step.shs.clone(), step._capacity)
    (s0, cam, bg, gt), (s2, _, _, _) = make(0), make(2)
    for it in range(4):                                  # interleaved in ONE process: each call runs inside its own context
        s0(cam, bg, gt)
        s2(cam, bg, gt)
    assert _lib.get_option("TILE_BINNING") == 2, "a step object leaked its option into the process"
    for binning, step in ((0, s0), (2, s2)):
        want = alone[binning]
        assert abs(float(step.loss()) - want[0]) <= 2e-5 * abs(want[0])
        for name, x, y in (("xyz", step.xyz, want[1]), ("shs", step.shs, want[2])):
            ok, msg = report("binning %d %s" % (binning, name), x, y, 1e-4, 1e-6)
            assert ok, msg
    # the bounded forward exists only for the direct binning: the object that selected the global sort never took it
    assert s0._use_bounded.__self__ is s0
    with s0._ctx:
        assert not s0._use_bounded(res, res)
    with s2._ctx:
        assert s2._use_bounded(res, res)


@pytest.mark.parametrize("H,W", [(64, 64), (37, 50), (5, 7), (120, 161), (70, 121), (3, 61)])
@pytest.mark.parametrize("weights,acc_normal,masked", [((1.0, 0.5, 1.0), 0, True), ((1.0, 0.0, 0.0), 0, False),
                                                       ((0.0, 0.5, 1.0), 1, True), ((0.0, 0.0, 1.0), 1, False)])
def test_fused_smoothness_kernel_equals_the_three_pass_formulation(H, W, weights, acc_normal, masked):
    """r3dg_stage2_smooth_fused (one kernel, the image streamed through registers in 60-column strips: maps, stencils and adjoint
    never touch HBM or LDS) against
    r3dg_stage2_smooth_forward + _backward -- the three-pass formulation that is pinned to the reference's calculate_loss
    (tests/test_reference_pipeline_gpu.py::test_stage2_syn4_objective_matches_the_reference_python).  The per-pixel arithmetic
    is the same, expression for expression: gradients bit-identical, the three sums up to the order of their float atomics;
    image sizes that are not multiples of the strip (one strip, three strips, a last strip of one column), every border, every subset of the three terms."""
    from relightable3dgaussian_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(H * 1000 + W)
    N = H * W
    opacity = torch.rand(1, H, W, generator=g).to(DEV)
    opacity[0, : H // 4] *= 1e-6                                             # (the clamp of the division at 1e-5)
    feature = (torch.rand(16, H, W, generator=g) * 1.5 - 0.2).to(DEV) * opacity
    n_contrib = (torch.rand(H, W, generator=g) > 0.15).to(torch.int32).to(DEV)
    gt = torch.rand(3, H, W, generator=g).to(DEV)
    mask = torch.rand(1, H, W, generator=g).to(DEV) if masked else None
    wb, wr, wl = (w / (3.0 * N) for w in weights)
    s = _lib.current_stream()
    outs = []
    for fused in (False, True):
        d_op = torch.full((1, H, W), 0.25, device=DEV)
        d_f = torch.full((16, H, W), -3.0, device=DEV)
        sums = torch.zeros(3, 32, device=DEV)
        args = (W, H, opacity.data_ptr(), feature.data_ptr(), n_contrib.data_ptr())
        if fused:
            _lib.check(L.r3dg_stage2_smooth_fused(s, *args, gt.data_ptr(), _lib.ptr(mask), wb, wr, wl, acc_normal, d_op.data_ptr(),
                                                  d_f.data_ptr(), sums.data_ptr()), "smooth_fused")
        else:
            scratch = torch.empty(30 * N, device=DEV)
            _lib.check(L.r3dg_stage2_smooth_forward(s, *args, gt.data_ptr(), _lib.ptr(mask), wb, wr, wl, scratch.data_ptr(),
                                                    sums.data_ptr()), "smooth_forward")
            _lib.check(L.r3dg_stage2_smooth_backward(s, *args, _lib.ptr(mask), scratch.data_ptr(), wb, wr, wl, acc_normal,
                                                     d_op.data_ptr(), d_f.data_ptr()), "smooth_backward")
        torch.cuda.synchronize()
        outs.append((d_op, d_f, sums.sum(1)))
    (o0, f0, s0), (o1, f1, s1) = outs
    assert torch.equal(f0, f1), "feature gradient maps differ: max %g" % float((f0 - f1).abs().max())
    assert torch.equal(o0, o1), "opacity gradient differs"
    assert torch.allclose(s0, s1, rtol=2e-5, atol=1e-6), (s0, s1)
    assert float(f1.abs().max()) > 0 and bool((f1[0:5] == -3.0).all())       # untouched maps stay untouched
