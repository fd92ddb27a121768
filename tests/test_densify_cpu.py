"""CPU: the densification oracle (oracle/densify.py) against the outputs of the reference's own GaussianModel
(tests/golden/densify_reference_*.npz, made by tests/golden/make_densify_golden.py)."""
import glob
import os

import numpy as np
import pytest

from oracle import densify as od

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(glob.glob(os.path.join(GOLDEN, "densify_reference_*.npz")))
STATS = ("weights_accum", "xyz_gradient_accum", "normal_gradient_accum", "denom", "max_radii2D")


def load_case(path):
    z = np.load(path)
    names = [str(n) for n in z["group_names"]]
    return z, names


def oracle_model(z, names, stats_prefix):
    return od.Model({n: z["pre_" + n] for n in names}, {n: z["pre_%s_exp_avg" % n] for n in names},
                    {n: z["pre_%s_exp_avg_sq" % n] for n in names}, {s: z["%s_%s" % (stats_prefix, s)] for s in STATS})


def run_op(m, z):
    op = str(z["op"])
    mss = float(z["max_screen_size"]) or None
    if op == "densify_and_prune":
        m.densify_and_prune(float(z["grad_threshold"]), float(z["min_opacity"]), float(z["extent"]), mss,
                            float(z["grad_normal_threshold"]), float(z["percent_dense"]), z["normal_table"],
                            float(z["weights_threshold"]))
    elif op == "prune":
        m.prune(float(z["min_opacity"]), float(z["extent"]), mss, float(z["weights_threshold"]))
    else:
        m.reset_opacity()


def test_fixtures_present():
    assert len(CASES) >= 5


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[18:-4] for p in CASES])
def test_accumulate_matches_reference(path):
    z, names = load_case(path)
    m = oracle_model(z, names, "pre")
    for v in range(int(z["views"])):
        od.accumulate(m.s, z["view%d_viewspace_grad" % v], z["view%d_normal_grad" % v], z["view%d_radii" % v],
                      z["view%d_weights" % v])
    for s in STATS:
        np.testing.assert_allclose(m.s[s], z["in_" + s], rtol=2e-7, atol=0, err_msg=s)


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[18:-4] for p in CASES])
def test_operation_matches_reference(path):
    z, names = load_case(path)
    m = oracle_model(z, names, "in")
    run_op(m, z)
    assert m.min_margin > 1e-5, "fixture has a value on a threshold (margin %.1e)" % m.min_margin
    assert m.P == z["out_xyz"].shape[0]
    generated = ("xyz", "scaling", "opacity")          # touched by float math (children / reset); the rest is moved
    for n in names:
        tol = dict(rtol=3e-6, atol=1e-6) if n in generated else dict(rtol=0, atol=0)
        np.testing.assert_allclose(m.p[n], z["out_" + n], err_msg=n, **tol)
        np.testing.assert_array_equal(m.m[n], z["out_%s_exp_avg" % n], err_msg=n + " exp_avg")
        np.testing.assert_array_equal(m.v[n], z["out_%s_exp_avg_sq" % n], err_msg=n + " exp_avg_sq")
    for s in STATS:
        np.testing.assert_array_equal(m.s[s], z["out_" + s], err_msg=s)


def test_host_mirror_structs_and_argument_checks():
    """Host logic that needs no GPU: the ctypes mirrors of r3dg_densify_config / r3dg_densify_group have the C layout,
    and the mirror refuses CPU tensors (there is no fallback path) and incomplete group sets before touching the library."""
    import collections
    import ctypes as C
    import torch
    from relightable3dgaussian_amd import densify as D
    assert C.sizeof(D.DensifyConfig) == 2 * 4 + 8 * 4
    assert C.sizeof(D.DensifyGroup) == 6 * C.sizeof(C.c_void_p) + 2 * 4
    assert D.MAX_GROUPS + 4 <= 24            # R3DG_DENSIFY_MAX_GROUPS, prune() appends four statistics rows
    P = 8
    groups = collections.OrderedDict(
        (n, dict(param=torch.zeros((P,) + s), exp_avg=None, exp_avg_sq=None))
        for n, s in (("xyz", (3,)), ("scaling", (3,)), ("rotation", (4,)), ("opacity", (1,))))
    st = D.DensificationStats(P, torch.device("cpu"))
    assert st.column("denom").shape == (P, 1) and st.max_radii2D.shape == (P,)
    with pytest.raises(RuntimeError, match="device tensor"):
        D.densify_and_prune(groups, st, 2e-4, 0.005, 4.0, 20, 2e-9, 0.01)
    with pytest.raises(RuntimeError, match="device tensor"):
        D.reset_opacity(groups["opacity"]["param"])
    with pytest.raises(RuntimeError, match="device tensor"):
        st.add(torch.zeros(P, 3), None, torch.zeros(P, dtype=torch.int32), torch.zeros(P, 1))
    del groups["rotation"]
    with pytest.raises(RuntimeError, match="rotation"):
        D.prune(groups, st, 0.005, 4.0, 20)


def test_training_loop_follows_the_reference_schedule(monkeypatch):
    """train_loop.train_stage1 against train.py:158-175, with the fused iteration replaced by a recorder (no GPU): which
    iterations collect statistics, densify (with which size / normal thresholds), reset the opacity, and skip the
    optimizer step (the reference's step() finds .grad = None on the freshly replaced parameters)."""
    import types
    import torch
    from relightable3dgaussian_amd import train_loop

    log = []

    class FakeStep:
        _opt_order = ("xyz", "normal")

        def __init__(self, init, lr, lr_rest_scale, process_group, lrs, loss_weights=None):
            self.dev, self.P, self.stats, self.lrs = torch.device("cpu"), 10, None, lrs
            self.opt = types.SimpleNamespace(groups=[dict(lr=lrs["xyz"]), dict(lr=lrs["normal"])])
            self.xyz_lrs = []
            self._iter, self.dp = 0, False            # (the step object counts its forward passes; single GPU)

        def enable_densification(self):
            self.stats = object()

        def forward_backward(self, cam, bg, gt, image_mask=None):
            self._iter += 1
            log.append(("fb", cam, self.stats is not None))
            self.xyz_lrs.append(self.opt.groups[0]["lr"])

        def replay_dropped(self, inputs_of):        # (FusedStage1Step.replay_dropped: the dropped forward passes, trained on again)
            cam = inputs_of(10)[0]
            self._iter += 1
            log.append(("replay", 10, cam))
            return [10]

        def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, max_grad_normal, percent_dense,
                              generator):
            log.append(("densify", max_grad, min_opacity, extent, max_screen_size, max_grad_normal, percent_dense))
            self.P += 1
            return dict(rows_out=self.P)

        def reset_opacity(self):
            log.append(("reset",))

        def optimizer_step(self):
            log.append(("step",))

        def poll_overflow(self):                    # (the bounded forward dropped one view around iteration 10)
            n_fb = sum(1 for x in log if x[0] == "fb")
            log.append(("poll", n_fb))
            if n_fb >= 10 and not getattr(self, "_told", False):
                self._told = True
                return 1
            return 0

    monkeypatch.setattr(train_loop, "FusedStage1Step", FakeStep)
    monkeypatch.setattr(torch, "Generator", lambda device=None: types.SimpleNamespace(manual_seed=lambda s: None))
    sch = train_loop.Schedule(densify_from_iter=4, densification_interval=3, densify_until_iter=14,
                              opacity_reset_interval=8, normal_densify_from_iter=7)
    step, history = train_loop.train_stage1(None, ["c0", "c1", "c2"], [0, 1, 2], None, extent=2.0, schedule=sch,
                                            iterations=16, white_background=True, poll_interval=5)
    # dropped views are asked for every poll_interval iterations, before every densify (6, 9, 12), on the two iterations behind
    # one (the instance count has just grown with P) and at the end, and reported
    assert [n for tag, n in (x for x in log if x[0] == "poll")] == [5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]
    assert [h for h in history if h[1] == "dropped_views"] == [(10, "dropped_views", 1)]
    # ... and trained on again right there: the view of forward pass 10 (c0), behind that iteration's own
    # optimizer step, in front of the next iteration's forward
    assert [h for h in history if h[1] == "replayed_views"] == [(10, "replayed_views", [10])]
    ir = log.index(("replay", 10, "c0"))
    assert log[ir - 1] == ("step",) and log[ir + 1][0] == "fb"
    history = [h for h in history if h[1] not in ("dropped_views", "replayed_views")]
    # reference: for it in 1..16: stats while it < 14; densify if it > 4 and it % 3 == 0 (and it < 14): 6, 9, 12;
    # reset if it % 8 == 0 or (white and it == 4), only while it < 14: 4, 8
    assert [(i, e) for i, e, _ in history] == [(4, "reset_opacity"), (6, "densify"), (8, "reset_opacity"),
                                               (9, "densify"), (12, "densify")]
    fb = [x for x in log if x[0] == "fb"]
    assert [c for _, c, _ in fb] == ["c0", "c1", "c2"] * 5 + ["c0"]              # round robin over the views
    assert [s for _, _, s in fb] == [True] * 13 + [False] * 3                    # statistics stop at densify_until_iter
    dens = [x for x in log if x[0] == "densify"]
    # size threshold 20 only after the first opacity-reset interval; normal threshold 99999 until normal_densify_from_iter
    assert [(d[4], d[5]) for d in dens] == [(None, 99999), (20, 2e-9), (20, 2e-9)]
    assert all(d[1:4] == (0.0002, 0.005, 2.0) and d[6] == 0.001 for d in dens)
    # 16 iterations, 3 of them densify -> 13 optimizer steps; a reset alone does not skip the step
    assert sum(1 for x in log if x == ("step",)) == 13
    i6 = log.index(dens[0])
    assert log[i6 + 1][0] == "fb"                                                # no optimizer step after the densify
    assert step.lrs["xyz"] == sch.position_lr_init * 2.0 and step.lrs["opacity"] == 0.05
    # update_learning_rate(iteration) before every iteration: log-linear decay of the position rate, scaled by the extent
    want = [train_loop.position_lr(i, 0.00016 * 2.0, 0.0000016 * 2.0, 0.01, 30000) for i in range(1, 17)]
    assert step.xyz_lrs == want and want[0] > want[-1] > 0


def test_position_lr_matches_the_reference_schedule():
    """train_loop.position_lr against values of the reference's get_expon_lr_func (tests/golden/lr_schedule_reference.npz,
    written by tests/golden/make_densify_golden.py)."""
    from relightable3dgaussian_amd import train_loop
    z = np.load(os.path.join(GOLDEN, "lr_schedule_reference.npz"))
    for name in ("default", "delayed", "disabled"):
        lr_init, lr_final, delay_mult, max_steps, delay_steps = z[name + "_args"]
        got = [train_loop.position_lr(int(s), lr_init, lr_final, delay_mult, int(max_steps), int(delay_steps))
               for s in z["steps"]]
        np.testing.assert_allclose(got, z[name], rtol=1e-12, atol=0, err_msg=name)


def test_c_abi_argument_validation_without_a_gpu():
    """The new entry points reject bad arguments before any HIP call (R3DG_EINVAL + a message), so this runs on the
    CPU-only box: error behaviour is part of the boundary."""
    import ctypes as C
    from relightable3dgaussian_amd import _lib, densify as D
    L = _lib.lib()
    EINVAL = -1
    cfg = D.DensifyConfig(0, 9, 2e-4, 2e-9, 0.005, 1e-4, 0.04, 0.4, 1.6, 20.0)
    counts = (C.c_int32 * 8)()
    assert L.r3dg_densify_plan(None, 16, C.addressof(cfg), 1, 1, 1, 1, 1, 1, 1, 1, 1, C.addressof(counts), 1) == EINVAL
    assert b"n_split" in L.r3dg_last_error()
    cfg.n_split, cfg.mode = 2, 7
    assert L.r3dg_densify_plan(None, 16, C.addressof(cfg), 1, 1, 1, 1, 1, 1, 1, 1, 1, C.addressof(counts), 1) == EINVAL
    cfg.mode = 0
    assert L.r3dg_densify_plan(None, 16, C.addressof(cfg), None, 1, 1, 1, 1, 1, 1, 1, 1, C.addressof(counts), 1) == EINVAL
    assert b"null" in L.r3dg_last_error()
    assert L.r3dg_densify_plan(None, -1, C.addressof(cfg), 1, 1, 1, 1, 1, 1, 1, 1, 1, C.addressof(counts), 1) == EINVAL
    assert L.r3dg_densify_accumulate(None, 8, None, None, 1, 1, 1, 1, 1, 1, 1, None) == EINVAL
    assert L.r3dg_densify_accumulate(None, 0, None, None, None, None, None, None, None, None, None, None) == 0      # P == 0
    grp = (D.DensifyGroup * 1)(D.DensifyGroup(1, None, None, 1, None, None, 4, 1))     # role xyz with 4-float rows
    assert L.r3dg_densify_gather(None, 8, 1, 1, 1, C.cast(grp, C.c_void_p), 1, 1, 1, None, 1.6) == EINVAL
    assert b"xyz rows" in L.r3dg_last_error()
    grp[0] = D.DensifyGroup(1, 1, None, 1, 1, 1, 3, 0)                                   # exp_avg without exp_avg_sq
    assert L.r3dg_densify_gather(None, 8, 1, 1, 1, C.cast(grp, C.c_void_p), 1, 1, 1, None, 1.6) == EINVAL
    assert L.r3dg_densify_gather(None, 8, 1, 1, 25, C.cast(grp, C.c_void_p), 1, 1, 1, None, 1.6) == EINVAL
    assert L.r3dg_densify_gather(None, 0, None, None, 0, None, None, None, None, None, 1.6) == 0               # nothing to do
    assert L.r3dg_reset_opacity(None, 4, None, None, None) == EINVAL
    assert L.r3dg_densify_temp_bytes(300000) >= 300000 + 4 * 4 * ((300000 + 255) // 256)
    # relight glue
    assert L.r3dg_relight_pack_features(None, 4, 1, 1, 1, 1, 1, 1, None) == EINVAL
    assert L.r3dg_relight_pack_features(None, 4, 16, 16, 16, 16, 16, 16, 20) == EINVAL                      # misaligned rows
    assert b"aligned" in L.r3dg_last_error()
    assert L.r3dg_relight_compose(None, 8, 8, 0.0, 10.0, 4.0, 4.0, 1, None, 1, 16, 32, None, 1, 1, 1, 1, None,
                                  None) == EINVAL                                                          # focal 0
    assert L.r3dg_relight_compose(None, 8, 8, 10.0, 10.0, 4.0, 4.0, 1, None, 1, 16, 32, None, 1, 1, 1, None, 1,
                                  None) == EINVAL                                                          # render_env w/o image
    assert L.r3dg_relight_compose(None, 0, 8, 10.0, 10.0, 4.0, 4.0, None, None, None, 16, 32, None, None, None, None,
                                  None, None, None) == 0                                                    # empty image
    assert L.r3dg_ssim_forward_pair(None, 8, 8, 3, 1, 1, 1, 1, None, None, None) == EINVAL                  # image 1 w/o partials


def test_scene_composition_matches_reference():
    """relight.compose_scenes (set_transform + create_from_gaussians + incident reset, relighting.py:28-52) against the
    reference's own GaussianModel on two objects under rotation x scale x translation (tests/golden/composition_reference.npz)."""
    import collections
    import torch
    from relightable3dgaussian_amd import relight
    z = np.load(os.path.join(GOLDEN, "composition_reference.npz"))
    names = [str(n) for n in z["group_names"]]
    objs = [collections.OrderedDict((n, torch.from_numpy(z["obj%d_%s" % (j, n)])) for n in names) for j in range(2)]
    trs = [torch.from_numpy(z["obj%d_transform" % j]) for j in range(2)]
    out = relight.compose_scenes(objs, trs)
    assert list(out.keys()) == names and out["xyz"].shape[0] == 65
    for n in names:
        tol = dict(rtol=2e-6, atol=2e-6) if n in relight.TRANSFORMED else dict(rtol=0, atol=0)
        np.testing.assert_allclose(out[n].numpy(), z["out_" + n], err_msg=n, **tol)
    assert float(out["incidents_dc"].abs().max()) == 0.0 and float(out["incidents_rest"].abs().max()) == 0.0
    assert float(out["base_color"].abs().max()) > 0.0
    with pytest.raises(RuntimeError):
        relight.compose_scenes(objs, trs[:1])
