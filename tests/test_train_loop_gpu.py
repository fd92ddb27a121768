"""From-scratch stage-1 training through train_loop.train_stage1: random-init points (distCUDA2 scales), the fused
stage-1 iteration, and the reference's densification schedule (train.py:158-175) on a synthetic teacher scene."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _psnr(a, b):
    return -10.0 * math.log10(float((a - b).square().mean()) + 1e-12)


def test_from_scratch_training_with_densification_schedule():
    from relightable3dgaussian_amd import synthetic as syn, train_loop
    from relightable3dgaussian_amd.bench_core import GaussianParams, render_stage1
    torch.manual_seed(7)
    res, n_views = 128, 6
    cams = [c.to(DEV) for c in syn.orbit_cameras(n_views, width=res, height=res)]
    bg = torch.ones(3, device=DEV)
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=3000, seed=9, stage2=False, scale_log_mean=-2.4), DEV, False)
        gts = [render_stage1(teacher, c, bg)[2].clone() for c in cams]
    g = torch.Generator().manual_seed(3)
    P0 = 1500
    pts = torch.rand(P0, 3, generator=g) * 2.6 - 1.3
    cols = torch.rand(P0, 3, generator=g)
    init = train_loop.create_from_points(pts, cols, device=DEV)
    assert init.scaling.shape == (P0, 3) and torch.isfinite(init.scaling).all()
    assert torch.equal(init.scaling[:, 0], init.scaling[:, 1])                    # isotropic

    sch = train_loop.Schedule(densify_from_iter=40, densification_interval=20, densify_until_iter=200,
                              opacity_reset_interval=120, percent_dense=0.01, densify_grad_threshold=1e-4)
    seen = {}

    def watch(it, step):
        if it in (1, 240):
            psnrs = []
            for c, gt in zip(cams, gts):
                step.forward_backward(c, bg, gt)        # forward only matters here; the gradients are overwritten next
                step._drain()
                psnrs.append(_psnr(step.last_outs[2], gt))
            seen[it] = sum(psnrs) / len(psnrs)

    # the watch hook must not feed the statistics: evaluate through a copy of the hook that pauses them
    def on_iteration(it, step):
        st, step.stats = step.stats, None
        try:
            watch(it, step)
        finally:
            step.stats = st

    step, history = train_loop.train_stage1(init, cams, gts, bg, extent=2.6, schedule=sch, iterations=240, seed=5,
                                            on_iteration=on_iteration)
    events = [e for _, e, _ in history]
    its = [i for i, e, _ in history if e == "densify"]
    assert its == list(range(60, 200, 20)), its                       # > from_iter, multiples of the interval, < until
    assert [i for i, e, _ in history if e == "reset_opacity"] == [40, 120]
    assert "densify" in events and any(rows != P0 for _, e, rows in history if e == "densify")
    assert step.stats is None                                         # statistics stop at densify_until_iter
    assert step.P == step.xyz.shape[0] == step.opt.groups[0]["exp_avg"].shape[0] > 0
    for k in ("xyz", "scaling", "rotation", "opacity", "shs", "normal"):
        assert torch.isfinite(getattr(step, k)).all(), k
    assert seen[240] > seen[1] + 2.0, seen                            # it learns (dB, view-averaged)
