"""CPU tests of the host-side mirrors against outputs of the reference's own Python (fixtures under tests/golden/, written by
tests/golden/make_densify_golden.py and make_golden.py): checkpoint format (train.py:190-203, GaussianModel.capture /
restore), learning rates, loss terms (SSIM, L1, TV, clipped sRGB), visibility inputs, and the committed bench line."""
import os
import types

import pytest
import torch

from relightable3dgaussian_amd import checkpoint as ck

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(autouse=True)
def _no_side_streams_shared_between_tests():
    """fused_step keeps ONE set of side streams per process (shared_stream); the tests below hand it fake stream objects, which
    must not outlive the test that made them."""
    from relightable3dgaussian_amd import fused_step
    saved = dict(fused_step._STREAMS)
    fused_step._STREAMS.clear()
    yield
    fused_step._STREAMS.clear()
    fused_step._STREAMS.update(saved)


def _load(stage):
    path = os.path.join(GOLDEN, "checkpoint_reference_stage%d.pth" % stage)
    return path, torch.load(path, map_location="cpu", weights_only=False)


@pytest.mark.parametrize("stage", [1, 2])
def test_restore_reads_the_reference_file(stage):
    path, (captured, iteration) = _load(stage)
    r = ck.restore(path)
    assert r.iteration == iteration == 777 and r.active_sh_degree == captured[0] == 3
    assert r.spatial_lr_scale == captured[14] and r.adam_steps == 2
    for name, idx in (("xyz", 1), ("normal", 2), ("features_dc", 3), ("features_rest", 4), ("scaling", 5), ("rotation", 6),
                      ("opacity", 7)):
        assert torch.equal(getattr(r, name), captured[idx].data) and not getattr(r, name).requires_grad
    P = r.xyz.shape[0]
    assert r.features_dc.shape == (P, 1, 3) and r.features_rest.shape == (P, 15, 3)
    assert torch.equal(r.max_radii2D, captured[8]) and float(r.max_radii2D.max()) > 0
    for i, n in enumerate(ck.STAT_NAMES):
        assert torch.equal(r.stats[n], captured[9 + i].reshape(-1))
    names = list(ck.STAGE1_GROUPS) + (list(ck.PBR_GROUPS) if stage == 2 else [])
    assert r.group_names == names and sorted(r.moments) == sorted(names)
    opt = captured[13]
    for g in opt["param_groups"]:
        st = opt["state"][g["params"][0]]
        assert torch.equal(r.moments[g["name"]][0], st["exp_avg"]) and torch.equal(r.moments[g["name"]][1], st["exp_avg_sq"])
        assert float(r.moments[g["name"]][0].abs().max()) > 0
    if stage == 2:
        assert r.base_color.shape == (P, 3) and r.incidents_rest.shape == (P, 15, 3) and r.visibility_rest.shape == (P, 15, 1)
    with pytest.raises(RuntimeError):
        ck.restore((captured[:7], 1))


def test_capture_round_trip_through_the_fused_stage1_step():
    """restore -> FusedStage1Step (CPU tensors; construction launches nothing) -> load_moments -> capture reproduces the
    reference's object entry by entry, including the optimizer state_dict layout."""
    from relightable3dgaussian_amd.densify import DensificationStats
    from relightable3dgaussian_amd.fused_step import FusedStage1Step
    path, (captured, iteration) = _load(1)
    r = ck.restore(path)
    step = FusedStage1Step(r)
    ck.load_moments(step, r)
    assert step.opt.step_count == 2 and step.shs.shape == (r.xyz.shape[0], 16, 3)
    step.stats = DensificationStats(r.xyz.shape[0], torch.device("cpu"))
    for n in ck.STAT_NAMES:
        getattr(step.stats, n).copy_(r.stats[n])
    step.stats.max_radii2D.copy_(r.max_radii2D)
    lrs = {g["name"]: g["lr"] for g in captured[13]["param_groups"]}
    ours, it = ck.capture(step, iteration, spatial_lr_scale=r.spatial_lr_scale, learning_rates=lrs)
    assert it == iteration and len(ours) == len(captured) == 15 and ours[0] == captured[0] and ours[14] == captured[14]
    for idx in range(1, 13):
        a, b = ours[idx], captured[idx]
        assert a.shape == b.shape and torch.equal(a.detach(), b.detach()), idx
    for idx in range(1, 8):
        assert isinstance(ours[idx], torch.nn.Parameter) and ours[idx].requires_grad          # what restore() assigns
    oa, ob = ours[13], captured[13]
    assert [g["name"] for g in oa["param_groups"]] == [g["name"] for g in ob["param_groups"]]
    for ga, gb in zip(oa["param_groups"], ob["param_groups"]):
        assert ga["lr"] == gb["lr"] and ga["eps"] == gb["eps"] == 1e-15 and ga["betas"] == gb["betas"]
        sa, sb = oa["state"][ga["params"][0]], ob["state"][gb["params"][0]]
        assert float(sa["step"]) == float(sb["step"])
        assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])


def test_capture_of_a_stage2_holder_has_the_reference_layout():
    """Stage-2 layout (21 entries) from a duck-typed holder with this repo's joined [P,16,3] tensors: dc / rest halves land
    in the reference's groups, the untrained baked-visibility groups are zeros of the reference's shapes."""
    _, (captured, _) = _load(2)
    r = ck.restore((captured, 5))
    P = r.xyz.shape[0]
    order = ("xyz", "normal", "scaling", "rotation", "opacity", "shs", "base_color", "roughness", "incidents", "env")
    joined = dict(shs=torch.cat([r.features_dc, r.features_rest], 1), incidents=torch.cat([r.incidents_dc, r.incidents_rest], 1),
                  env=torch.zeros(1, 4, 8, 3))
    tensors = {k: joined.get(k, getattr(r, k, None)) for k in order}
    g = torch.Generator().manual_seed(0)
    groups = [dict(exp_avg=torch.randn(tensors[k].shape, generator=g), exp_avg_sq=torch.rand(tensors[k].shape, generator=g))
              for k in order]
    holder = types.SimpleNamespace(_opt_order=order, opt=types.SimpleNamespace(groups=groups, step_count=9), stats=None,
                                   **tensors)
    ours, it = ck.capture(holder, 40000, spatial_lr_scale=2.5)
    assert it == 40000 and len(ours) == 21 and ours[14] == 2.5
    for idx in range(15, 21):
        assert ours[idx].shape == captured[idx].shape
    assert torch.equal(ours[17].detach(), r.incidents_dc) and torch.equal(ours[18].detach(), r.incidents_rest)
    assert float(ours[19].detach().abs().max()) == 0.0 and float(ours[20].detach().abs().max()) == 0.0
    names = [gr["name"] for gr in ours[13]["param_groups"]]
    assert names == list(ck.STAGE1_GROUPS + ck.PBR_GROUPS)
    st = ours[13]["state"]
    i_rest = names.index("incidents_rest")
    assert torch.equal(st[i_rest]["exp_avg"], groups[order.index("incidents")]["exp_avg"][:, 1:]) and float(st[i_rest]["step"]) == 9
    back = ck.restore((ours, it))
    assert back.adam_steps == 9 and torch.equal(back.base_color, r.base_color) and float(back.stats["denom"].sum()) == 0.0


def test_fused_stage2_step_takes_the_reference_learning_rates(monkeypatch):
    """Constructor logic only (the visibility trace and the activation kernel are stubbed, nothing runs on a GPU): per-group
    rates as run_nerf.sh sets them for stage 2, dc / rest split of the joined SH tensors, one shared slab for the gradients."""
    from relightable3dgaussian_amd import fused_step
    _, (captured, _) = _load(2)
    r = ck.restore((captured, 30000))
    r.env = torch.zeros(1, 16, 32, 3)
    P = r.xyz.shape[0]
    monkeypatch.setattr(fused_step, "update_visibility", lambda *a, **k: (None, None, None, None))
    monkeypatch.setattr(fused_step.FusedStage2Step, "refresh_activations", lambda self, cam=None: None)
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: None)
    lrs = dict(xyz=0.000016 * 2.6, normal=0.001, shs=0.00025, opacity=0.005, scaling=0.0005, rotation=0.0001,
               base_color=0.01, roughness=0.01, incidents=0.001, incidents_rest=0.0001, env=0.1)
    step = fused_step.FusedStage2Step(r, 8, lr=1e-4, lr_rest_scale=1.0 / 20.0, lrs=lrs)
    g = {k: step.opt.groups[i] for i, k in enumerate(step._opt_order)}
    assert g["xyz"]["lr"] == lrs["xyz"] and g["opacity"]["lr"] == 0.005 and g["env"]["lr"] == 0.1
    assert g["shs"]["lr"] == 0.00025 and g["shs"]["lr_tail"] == 0.00025 / 20.0 and g["shs"]["period"] == 48 and g["shs"]["split"] == 3
    assert g["incidents"]["lr"] == 0.001 and g["incidents"]["lr_tail"] == 0.0001
    assert step.shs.shape == (P, 16, 3) and torch.equal(step.features_rest, r.features_rest)
    # gradient slab: [shs, overflow flag | per-Gaussian groups + env | incidents], every group 16-byte aligned, env inside
    # bucket C, the bounded forward's flag (4 floats) at the end of bucket A
    total = 4 + sum((x.numel() + 3) // 4 * 4 for x in (step.shs, step.xyz, step.normal, step.scaling, step.rotation,
                                                       step.opacity, step.base_color, step.roughness, step.env, step.incidents))
    assert step.grad_flat.numel() == total
    assert step._bucket_a.numel() == step.shs.numel() + 4 and step._bucket_b.numel() == step.incidents.numel()
    assert step._flag.data_ptr() == step._bucket_a.data_ptr() + 4 * step.shs.numel() and step._flag.numel() == 4
    assert step._bucket_a.numel() + step._bucket_b.numel() + step._bucket_c.numel() == total
    for k, gr in step.grads.items():
        assert gr.shape == getattr(step, k).shape and gr.data_ptr() % 16 == 0, k
    lo, hi = step._bucket_c.data_ptr(), step._bucket_c.data_ptr() + 4 * step._bucket_c.numel()
    assert lo <= step.grads["env"].data_ptr() < hi and lo <= step.grads["xyz"].data_ptr() < hi
    # lookup cache of the incident directions: rebuilt when the direction tensor is REPLACED (even by one that the allocator
    # hands the old address) or changed in place, not otherwise
    from relightable3dgaussian_amd import shading_ops
    builds = []
    monkeypatch.setattr(shading_ops, "build_taps", lambda dirs, He, We, *a, **k: builds.append(dirs) or torch.zeros(3))
    step.incident_dirs, step.incident_areas = torch.ones(P, 8, 3), torch.full((P, 8, 1), 2.0)
    t0 = step.taps(16, 32)
    assert step.taps(16, 32) is t0 and len(builds) == 1 and step._uniform_area == 2.0
    step.incident_dirs = None
    step.incident_dirs = torch.zeros(P, 8, 3)            # whatever address the allocator hands out: a new cache
    step.taps(16, 32)
    assert len(builds) == 2 and builds[1] is step.incident_dirs
    step.incident_dirs.add_(1.0)
    step.taps(16, 32)
    step.taps(8, 16)
    assert len(builds) == 4
    ck.load_moments(step, r)
    assert step.opt.step_count == 2
    assert torch.equal(step.opt.groups[step._opt_order.index("incidents")]["exp_avg"][:, 1:], r.moments["incidents_rest"][0])


def test_relight_renderer_host_logic_with_a_recording_library(monkeypatch):
    """RelightRenderer.frame's host side only (every C-ABI entry point replaced by a recorder, nothing runs on a GPU): which
    shading entry point a frame takes and with which cache -- transport cache (default), radiance cache, and no cache once
    the light turns with every frame."""
    import types
    from relightable3dgaussian_amd import _lib, rasterizer_ops, relight, shading_ops
    calls = []

    class Recorder:
        def __getattr__(self, name):
            def fn(*args):
                calls.append((name, args))
                return 0
            return fn

    class DeviceTensor(torch.Tensor):            # a CPU tensor that claims to be a device tensor
        is_cuda = property(lambda self: True)

    dt = lambda t: t.as_subclass(DeviceTensor)
    P, K = 7, 8
    monkeypatch.setattr(_lib, "lib", lambda: Recorder())
    monkeypatch.setattr(_lib, "current_stream", lambda: 0)
    monkeypatch.setattr(shading_ops, "_c", lambda t: t.contiguous())          # (the renderer's own buffers are CPU tensors here)
    import contextlib
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    monkeypatch.setattr(relight, "update_visibility", lambda *a, **k: (torch.ones(P, K, 1), torch.ones(P, K, 3),
                                                                       torch.full((P, K, 1), 2.0), None))
    built = []
    monkeypatch.setattr(shading_ops, "build_taps", lambda dirs, He, We, tr=None, radiance_of=None:
                        built.append(tr) or torch.zeros(P * K * 3))
    z = torch.zeros
    class _Pending:
        def finish(self, ordering_stream=None):
            return (3, z(1), z(3, 4, 4), z(1, 4, 4), z(1, 4, 4), z(28, 4, 4), z(3, 4, 4), z(3, 4, 4), None, z(P))
    wanted = []
    monkeypatch.setattr(rasterizer_ops, "rasterize_gaussians_begin",
                        lambda *a, want_weights=True, **k: wanted.append(want_weights) or _Pending())
    model = types.SimpleNamespace(xyz=dt(z(P, 3)), normal=z(P, 3), scaling=z(P, 3), rotation=z(P, 4), opacity=z(P, 1),
                                  base_color=z(P, 3), roughness=z(P, 1), shs=z(P, 16, 3), incidents=z(P, 16, 3))
    cam = types.SimpleNamespace(image_height=4, image_width=4, world_view_transform=torch.eye(4), full_proj_transform=torch.eye(4),
                                camera_center=z(3), tanfovx=0.5, tanfovy=0.5, cx=2.0, cy=2.0)
    env = dt(z(8, 16, 3))
    names = lambda: [c[0] for c in calls]
    # radiance cache, built once for a fixed light
    r = relight.RelightRenderer(model, env, K, cache="radiance")
    calls.clear()
    for _ in range(3):
        out = r.frame(cam, z(3))
    assert len(built) == 1 and names().count("r3dg_shade_forward_cached") == 3 and "r3dg_shade_forward_transport" not in names()
    cached = [c for c in calls if c[0] == "r3dg_shade_forward_cached"][-1][1]
    assert cached[-2] == 2 and cached[-3] is not None and cached[16] == 2.0 and cached[15] is None     # radiance taps, uniform area
    assert set(out) >= {"render", "feature", "pbr_env", "num_rendered"} and out["num_rendered"] == 3
    assert wanted and not any(wanted)                       # frames never ask for the per-Gaussian blend weights
    # a light that turns with every frame: cache on the first change only, afterwards the split transport (built once) with the
    # lookup inside the per-frame kernel
    calls.clear()
    del built[:]
    import math

    def rot(i):
        a = 0.3 * i
        return torch.tensor([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0], [0.0, 0.0, 1.0]])
    for i in range(4):
        r.frame(cam, z(3), env_transform=rot(i))
    flags = [c[1][-2] for c in calls if c[0] == "r3dg_shade_forward_cached"]
    assert len(built) == 1 and flags == [2], (len(built), flags)
    assert names().count("r3dg_shade_build_split") == 1 and names().count("r3dg_shade_forward_split") == 3
    sp = [c for c in calls if c[0] == "r3dg_shade_forward_split"][-1][1]
    assert len(sp) == 17 and sp[1:3] == (P, K) and sp[12] is not None and sp[14:16] == (8, 16)      # the rotation; He, We
    # the same with DEVICE matrices built per frame (relighting.py:162-163): keyed by storage identity, and the renderer keeps
    # the tensors it keyed on alive, so a recycled address cannot pass for "the light did not move"
    r2 = relight.RelightRenderer(model, env, K, cache="radiance")
    calls.clear()
    del built[:]
    for i in range(6):
        r2.frame(cam, z(3), env_transform=dt(rot(i)))
    flags = [c[1][-2] for c in calls if c[0] == "r3dg_shade_forward_cached"]
    assert len(built) == 1 and flags == [2] and names().count("r3dg_shade_forward_split") == 5, (len(built), flags)
    fixed = dt(rot(9))
    for i in range(3):
        r2.frame(cam, z(3), env_transform=fixed)
    flags = [c[1][-2] for c in calls if c[0] == "r3dg_shade_forward_cached"][1:]
    # stopped: its first frame is still a change (split kernel), cached again from the second one
    assert len(built) == 2 and flags == [2, 2] and names().count("r3dg_shade_forward_split") == 6, (len(built), flags)
    assert names().count("r3dg_shade_build_split") == 1
    # the split cache is keyed like the lookup cache.  A map edited in place / swapped in between two frames of a
    # turning light rebuilds the footprints (and only them); swapped visibility rebuilds the sample half; a HOST transform that
    # is not a rotation (scaled) never takes the split kernel, which evaluates the lobe in the light's frame
    r2.frame(cam, z(3), env_transform=dt(rot(18)))         # (the light turns again: the first change still builds a lookup cache)
    calls.clear()
    r2.frame(cam, z(3), env_transform=dt(rot(20)))
    r2.frame(cam, z(3), env_transform=dt(rot(21)))
    assert names().count("r3dg_shade_env_footprints") == 0 and names().count("r3dg_shade_forward_split") == 2
    r2.envmap.add_(1.0)                                    # edited in place: version counter
    r2.frame(cam, z(3), env_transform=dt(rot(22)))
    assert names().count("r3dg_shade_env_footprints") == 1 and names().count("r3dg_shade_build_split") == 0
    r2.envmap = dt(z(8, 16, 3))                            # swapped
    r2.frame(cam, z(3), env_transform=dt(rot(23)))
    assert names().count("r3dg_shade_env_footprints") == 2 and names().count("r3dg_shade_build_split") == 0
    r2.visibility = torch.ones(P, K, 1)
    r2.frame(cam, z(3), env_transform=dt(rot(24)))
    assert names().count("r3dg_shade_build_split") == 1 and names().count("r3dg_shade_env_footprints") == 2
    n_split = names().count("r3dg_shade_forward_split")
    r2.frame(cam, z(3), env_transform=torch.eye(3) * 2.0)              # scaled, given on the host: general kernel
    assert names().count("r3dg_shade_forward_split") == n_split and calls[-4][0] != "r3dg_shade_forward_split"
    assert [c for c in calls if c[0] == "r3dg_shade_forward_cached"][-1][1][-2] == 0
    r2.envmap = dt(z(4096, 8, 3))                          # taller than the footprint records address: general kernel, no error
    r2.frame(cam, z(3), env_transform=dt(rot(25)))
    assert names().count("r3dg_shade_forward_split") == n_split
    # the default, transport cache: one build (radiance -> transport in place + constants), then the transport kernel per frame
    r = relight.RelightRenderer(model, env, K)
    assert r.cache == "transport" and r.xyz is not model.xyz            # (works on a snapshot of the parameters)
    calls.clear()
    for _ in range(3):
        r.frame(cam, z(3))
    n = names()
    assert n.count("r3dg_shade_build_transport") == 1 and n.count("r3dg_shade_forward_transport") == 3
    assert "r3dg_shade_forward_cached" not in n
    tr_args = [c for c in calls if c[0] == "r3dg_shade_forward_transport"][0][1]
    assert len(tr_args) == 12 and tr_args[1:3] == (P, K) and tr_args[10] is None and tr_args[9] == r._zsamples.data_ptr()
    assert r._zsamples.shape == (K, 3) and r._consts.shape == (P, 16)
    b_args = [c for c in calls if c[0] == "r3dg_shade_build_transport"][0][1]
    assert len(b_args) == 12 and b_args[1:4] == (P, K, 16) and b_args[8] is None and b_args[9] == 2.0
    with pytest.raises(RuntimeError):
        relight.RelightRenderer(model, env, K, cache="everything")


def test_committed_bench_line_follows_the_contract():
    """profiles/r02_bench_default.json is the line `python bench.py` printed on the MI355X: the keys the driver and the judge
    read are there, with the types and relations the contract states."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = json.load(open(os.path.join(root, "profiles", "r02_bench_default.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "iters/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["n_gpus"] == 1 and "workload" in d["config"]
    assert abs(d["value"] - d["n_gpus"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-3
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] and c["sample"]
    assert c["extra"]["c_port"]["unit"] == "iters/s" and c["extra"]["c_port"]["cores"] == 1     # the scalar C port, as round 1
    assert d["value"] >= 40.0 and d["relight"]["relight_fps"] >= 60.0            # BASELINE.json targets on 1x MI355X
    rr = d["roofline_relight"]
    assert rr["unit"] == "GB/s" and abs(rr["frac"] - rr["achieved"] / rr["peak"]) < 1e-3
    sp = d["spread_iters_per_s"]
    assert sp["min"] <= sp["median"] <= sp["max"] and sp["blocks"] >= 3


def test_compact_bench_line_stays_small_and_strict():
    """The LAST stdout line of bench.py is what the driver parses (round 4: a 23 KB line left BENCH_r04.parsed null): the
    compact form of every committed full document is strict JSON under 4 KB and still carries the contract's keys, `roofline`
    and `cpu_baseline`; and `bench.py --plumbing-only` really prints exactly one such line on stdout."""
    import glob
    import json
    import subprocess
    import sys
    from relightable3dgaussian_amd import bench_core
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    docs = sorted(glob.glob(os.path.join(root, "profiles", "r0[2-9]_bench_default.json")))
    assert docs
    for path in docs:
        d = json.load(open(path))
        if "kernels" not in d:               # (round 5 on: the committed file may already be a compact line)
            continue
        line = bench_core.compact(d)
        assert len(line.encode()) < bench_core.COMPACT_LIMIT <= 4096 and "\n" not in line, (path, len(line))
        c = json.loads(line, parse_constant=lambda x: (_ for _ in ()).throw(ValueError(x)))        # strict: no NaN / Infinity
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in c, (path, k)
        assert c["value"] == d["value"] and c["ms_per_step"] == d["ms_per_step"] and "workload" in c["config"]
        r = c["roofline"]
        assert r["kernel"] and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert c["cpu_baseline"]["cores"] >= 1 and c["cpu_baseline"]["kind"] == "port"
        assert c["relight_fps"] == d["relight"]["relight_fps"]
    # a pathological document (huge strings everywhere) still yields a line under the limit
    big = json.load(open(docs[-1]))
    big["config"]["workload"] = "x" * 5000
    big.setdefault("cpu_baseline", {})["sample"] = "y" * 5000
    big["other_configs"] = {("k%d" % i) * 10: {"iters_per_s": 1.0 * i} for i in range(200)}
    assert len(bench_core.compact(big)) < bench_core.COMPACT_LIMIT
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--plumbing-only", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=240, env=e, stdin=subprocess.DEVNULL, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    out = [x for x in r.stdout.splitlines() if x.strip()]
    assert len(out[-1]) < 4096 and sum(1 for x in out if x.startswith("{")) == 1
    last = json.loads(out[-1])
    assert last["plumbing_only"] is True and last["n_gpus"] == 1 and last["steps"] == 2
    assert "bench_full: {" in r.stderr


def test_ssim_and_image_loss_match_the_reference():
    """train_step.ssim / image_loss -- the parity targets of the HIP SSIM kernels and of the fused iterations' image terms --
    against the reference's utils/loss_utils.ssim + l1_loss and their gradient (tests/golden/ssim_reference.npz)."""
    import numpy as np
    from relightable3dgaussian_amd import train_step
    z = np.load(os.path.join(GOLDEN, "ssim_reference.npz"))
    for name in ("a", "b"):
        x = torch.from_numpy(z[name + "_x"]).requires_grad_(True)
        y = torch.from_numpy(z[name + "_y"])
        s = train_step.ssim(x, y)
        assert abs(float(s.detach()) - float(z[name + "_ssim"])) < 2e-6
        loss = train_step.image_loss(x, y)
        assert abs(float(loss.detach()) - float(z[name + "_loss"])) < 2e-6
        assert abs(float((x - y).abs().mean().detach()) - float(z[name + "_l1"])) < 1e-7
        loss.backward()
        np.testing.assert_allclose(x.grad.numpy(), z[name + "_grad"], rtol=1e-4, atol=2e-7)


def test_tv_loss_matches_the_reference():
    """The env-smoothness term (neilf.py:303-307 -> utils/loss_utils.tv_loss: mean SQUARED differences)."""
    import numpy as np
    from relightable3dgaussian_amd import train_step
    z = np.load(os.path.join(GOLDEN, "ssim_reference.npz"))
    for name in ("a", "b"):
        assert abs(float(train_step.tv_loss(torch.from_numpy(z[name + "_x"]))) - float(z[name + "_tv"])) < 1e-7
        got = train_step.psnr(torch.from_numpy(z[name + "_x"]), torch.from_numpy(z[name + "_y"]))       # image_utils.psnr
        np.testing.assert_allclose(got.numpy(), z[name + "_psnr"], rtol=1e-6, atol=1e-5)
        assert got.shape == (3, 1)


def test_rgb_to_srgb_matches_the_reference_including_the_clip():
    """results["pbr"] = rgb_to_srgb(...) with clip=True (neilf.py:179, utils/graphics_utils.py:207-213): values and gradient,
    below 0, around the knee, inside (0,1) and above 1 (where the clamp stops the gradient)."""
    import numpy as np
    from relightable3dgaussian_amd import relight, train_step
    z = np.load(os.path.join(GOLDEN, "ssim_reference.npz"))
    for fn in (train_step.rgb_to_srgb, relight.rgb_to_srgb):
        v = torch.from_numpy(z["srgb_in"]).requires_grad_(True)
        out = fn(v)
        np.testing.assert_allclose(out.detach().numpy(), z["srgb_out"], rtol=1e-6, atol=1e-7)
        (out * torch.from_numpy(z["srgb_w"])).sum().backward()
        np.testing.assert_allclose(v.grad.numpy(), z["srgb_grad"], rtol=1e-5, atol=1e-7)
    assert float((z["srgb_grad"] == 0).mean()) > 0.3 and float(z["srgb_out"].max()) == 1.0 and float(z["srgb_out"].min()) == 0.0


def test_host_side_visibility_inputs_match_the_reference():
    """The pure-PyTorch host functions that feed the BVH trace and the shading caches, on the reference's own outputs:
    train_step.inverse_covariance vs GaussianModel.get_inverse_covariance (covariance_reference.npz) and
    sampling.fibonacci_sphere_sampling vs utils/graphics_utils.fibonacci_sphere_sampling incl. the n_z = -1 branch of
    rotation_between_z (fibonacci_reference.npz)."""
    import numpy as np
    from relightable3dgaussian_amd import sampling, train_step
    z = np.load(os.path.join(GOLDEN, "covariance_reference.npz"))
    inv = train_step.inverse_covariance(torch.from_numpy(z["scales"]), torch.from_numpy(z["rotations"]))
    ref = z["cov3D_inverse"]
    np.testing.assert_allclose(inv.numpy(), ref, rtol=2e-5, atol=2e-5 * float(np.abs(ref).max()))
    f = np.load(os.path.join(GOLDEN, "fibonacci_reference.npz"))
    dirs, areas = sampling.fibonacci_sphere_sampling(torch.from_numpy(f["normals"]), f["dirs"].shape[1])
    np.testing.assert_allclose(dirs.numpy(), f["dirs"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(areas.numpy(), f["areas"], rtol=0, atol=1e-6)
    assert dirs.shape == (f["normals"].shape[0], f["dirs"].shape[1], 3) and areas.shape[-1] == 1


@pytest.mark.parametrize("variant", ["sh_scale", "color_cov"])
def test_autograd_wrapper_calls_the_backend_like_the_reference(variant, monkeypatch):
    """relightable3dgaussian_amd.rasterizer (GaussianRasterizationSettings / GaussianRasterizer /
    autograd Function) against the call trace of the reference's gaussian_renderer/r3dg_rasterization.py recorded with a
    fake `_C` (tests/wrapper_trace.py -> tests/golden/wrapper_trace_reference.json): same 23 / 26 positional arguments in
    the same order (absent optionals as empty CPU tensors), same 10 outputs, same routing of the nine gradients."""
    import json
    from relightable3dgaussian_amd import rasterizer
    from tests import wrapper_trace
    want = json.load(open(os.path.join(GOLDEN, "wrapper_trace_reference.json")))[variant]

    def install(fwd, bwd):
        monkeypatch.setattr(rasterizer._ops, "rasterize_gaussians", fwd)
        monkeypatch.setattr(rasterizer._ops, "rasterize_gaussians_backward", bwd)
    got = wrapper_trace.run(rasterizer.GaussianRasterizationSettings, rasterizer.GaussianRasterizer, install, variant)
    assert got["n_outputs"] == want["n_outputs"] == 10 and got["num_rendered"] == want["num_rendered"] == 7
    assert len(got["forward_args"]) == len(want["forward_args"]) == 23
    assert len(got["backward_args"]) == len(want["backward_args"]) == 26
    for which in ("forward_args", "backward_args"):
        for i, (a, b) in enumerate(zip(got[which], want[which])):
            assert a == b, "%s[%d]: %r vs reference %r" % (which, i, a, b)
    assert got["grad_routing"] == want["grad_routing"]
    assert tuple(rasterizer.GaussianRasterizationSettings._fields) == (
        "image_height", "image_width", "tanfovx", "tanfovy", "cx", "cy", "bg", "scale_modifier", "viewmatrix", "projmatrix",
        "sh_degree", "campos", "prefiltered", "backward_geometry", "computer_pseudo_normal", "debug")


def test_raytracer_wrapper_calls_the_backend_like_the_reference(monkeypatch):
    """SURVEY.md 8 rows a13/a14 (Python side): relightable3dgaussian_amd.bvh.RayTracer against the call trace of the
    reference's bvh/__init__.py:28-71 -- create_bvh's five arguments, trace_bvh_opacity's eight (ray origins offset by
    0.05 d), the {visibility, contribute} result with a trailing singleton axis."""
    import json
    from relightable3dgaussian_amd import bvh
    from tests import wrapper_trace
    want = json.load(open(os.path.join(GOLDEN, "wrapper_trace_reference.json")))["raytracer"]

    packed = []

    def install(create, trace):
        monkeypatch.setattr(bvh.bvh_ops, "create_bvh", create)
        # (this repo's tracer additionally hands over its packed traversal records: the eight reference arguments unchanged)
        monkeypatch.setattr(bvh._lib, "get_option", lambda name: 4)
        monkeypatch.setattr(bvh.bvh_ops, "trace_records", lambda *a: packed.append(len(a)) or "records")
        monkeypatch.setattr(bvh.bvh_ops, "trace_bvh_opacity", lambda *a, records=None: packed.append(records) or trace(*a))
    got = wrapper_trace.run_raytracer(bvh.RayTracer, install)
    assert packed == [6, "records"]
    assert got["offset_ok"] and want["offset_ok"]
    assert got["create_args"] == want["create_args"]
    assert len(got["trace_args"]) == len(want["trace_args"]) == 8
    for i, (a, b) in enumerate(zip(got["trace_args"], want["trace_args"])):
        assert a["shape"] == b["shape"] and a["dtype"] == b["dtype"] and abs(a["first"] - b["first"]) < 1e-6, (i, a, b)
    assert got["result"] == want["result"]


def test_synthetic_cameras_follow_the_reference_camera_class():
    """synthetic.look_at_camera against the reference's Camera (scene/cameras.py:8-73 + utils/graphics_utils
    getWorld2View2 / getProjectionMatrix), built as relighting.py:150-158 builds it: the matrices every benchmark and parity
    test hands to the ops have the reference's conventions (W2C transposed, full projection, camera centre, FoVy from FoVx)."""
    import numpy as np
    from relightable3dgaussian_amd import synthetic as syn
    z = np.load(os.path.join(GOLDEN, "camera_reference.npz"))
    for j in range(int(z["n"])):
        W, H = [int(v) for v in z["cam%d_size" % j]]
        cam = syn.look_at_camera(tuple(z["cam%d_eye" % j]), target=tuple(z["cam%d_target" % j]), width=W, height=H)
        np.testing.assert_allclose(cam.world_view_transform.numpy(), z["cam%d_world_view_transform" % j], rtol=0, atol=2e-6)
        np.testing.assert_allclose(cam.full_proj_transform.numpy(), z["cam%d_full_proj_transform" % j], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(cam.camera_center.numpy(), z["cam%d_camera_center" % j], rtol=0, atol=5e-6)
        assert abs(cam.FoVy - float(z["cam%d_fovy" % j])) < 1e-9 and cam.image_width == W and cam.image_height == H
        assert abs(cam.tanfovx - np.tan(0.5 * cam.FoVx)) < 1e-12 and cam.cx == W / 2.0 and cam.cy == H / 2.0


def test_blender_dataset_writer_is_read_back_by_the_reference_reader(tmp_path):
    """synthetic.write_blender_dataset: the same split written again here must give the reference's readCamerasFromTransforms
    (scene/dataset_readers.py:215-270) what tests/golden/blender_dataset_reference.npz recorded from it -- R (stored
    transposed, as the reader does), T, FovX / FovY and the 8-bit image content -- i.e. our camera poses survive the
    OpenCV <-> Blender axis flip and the JSON layout."""
    import json
    import numpy as np
    from PIL import Image
    from relightable3dgaussian_amd import synthetic as syn
    z = np.load(os.path.join(GOLDEN, "blender_dataset_reference.npz"))
    cams = syn.orbit_cameras(5, width=24, height=24)[:3]
    imgs = [torch.from_numpy(z["v%d_written" % j]) for j in range(3)]
    path = syn.write_blender_dataset(str(tmp_path), cams, imgs, split="train")
    doc = json.load(open(path))
    assert abs(doc["camera_angle_x"] - cams[0].FoVx) < 1e-15 and len(doc["frames"]) == 3
    for j, (cam, frame) in enumerate(zip(cams, doc["frames"])):
        assert frame["file_path"] == "./train/r_%d" % j and str(z["v%d_name" % j]) == "r_%d" % j
        # what the reference reader derives from this frame (its arithmetic restated: flip, invert, transpose)
        c2w = np.array(frame["transform_matrix"])
        c2w[:3, 1:3] *= -1
        w2c = np.linalg.inv(c2w)
        np.testing.assert_allclose(np.transpose(w2c[:3, :3]), z["v%d_R" % j], rtol=0, atol=1e-12)
        np.testing.assert_allclose(w2c[:3, 3], z["v%d_T" % j], rtol=0, atol=1e-12)
        # ... and that equals the camera we started from
        ours = cam.world_view_transform.double().t().numpy()
        np.testing.assert_allclose(w2c, ours, rtol=0, atol=1e-6)
        assert abs(z["v%d_fov" % j][0] - cam.FoVx) < 1e-12 and abs(z["v%d_fov" % j][1] - cam.FoVy) < 1e-9
        png = np.asarray(Image.open(os.path.join(tmp_path, "train", "r_%d.png" % j))) / 255
        np.testing.assert_array_equal(png, z["v%d_image" % j])
        assert np.abs(png - imgs[j].permute(1, 2, 0).numpy()).max() <= 0.5 / 255 + 1e-7
    with pytest.raises(RuntimeError):
        syn.write_blender_dataset(str(tmp_path), syn.orbit_cameras(2, width=32, height=24), [torch.zeros(3, 24, 32)] * 2)
    # scene.cameras_extent of the same views (getNerfppNorm on what the reader returned)
    radius, translate = syn.cameras_extent(cams)
    assert abs(radius - float(z["extent_radius"])) < 1e-5 * radius
    np.testing.assert_allclose(translate, z["extent_translate"], rtol=0, atol=1e-5)


def test_stage1_loss_restatement_equals_the_reference_calculate_loss():
    """train_step.stage1_loss (the parity target of the fused stage-1 iteration, incl. the restated kornia Sobel stencil) on the
    maps the reference's own render_view produced = the loss its own calculate_loss returned, term by term
    (tests/golden/pipeline_reference_stage1.npz, made by tests/golden/make_pipeline_golden.py from the unmodified reference)."""
    import numpy as np
    from relightable3dgaussian_amd import train_step as ts
    z = np.load(os.path.join(GOLDEN, "pipeline_reference_stage1.npz"))
    t = lambda k: torch.from_numpy(z[k])
    opacity, image, gt, mask = t("map_opacity"), t("map_render"), t("gt"), t("mask")
    normal, depth, var = t("map_normal"), t("map_depth"), t("map_depth_var")
    # rebuild the rasterizer's premultiplied feature row from the divided maps (the division is undone exactly where it happened)
    n_contrib = (opacity[0] > 0).int()
    feature = torch.cat([normal, depth, var + depth.square()], 0) * opacity.clamp_min(1e-5)
    outs = (0, n_contrib, image, opacity, None, feature, t("map_pseudo_normal"), None, None, None)
    it = int(z["iteration"])
    loss = ts.stage1_loss(outs, gt, mask, None, it)
    assert abs(float(loss) - float(z["loss"])) < 2e-6 * max(1.0, abs(float(z["loss"])))
    l1, ssim_v, ent, nrd, nsm, dvar = [float(v) for v in z["tb"]]
    assert abs(float((image - gt).abs().mean()) - l1) < 1e-6 and abs(float(ts.ssim(image, gt)) - ssim_v) < 2e-6
    assert abs(float(ts.first_order_edge_aware_loss(normal, gt)) - nsm) < 2e-6
    assert abs(float(var.clamp_min(1e-6).sqrt().mean()) - dvar) < 2e-6


def test_stage2_smoothness_restatement_equals_the_reference_calculate_loss():
    """train_step.stage2_smoothness (the three edge-aware terms of script/run_syn4.sh / run_dtu.sh, neilf.py:275-292) on the maps
    the reference's own render_view produced = the terms its own calculate_loss logged with those flags
    (tests/golden/pipeline_reference_stage2_syn4.npz; maps and mask from pipeline_reference_stage2.npz, same inputs)."""
    import numpy as np
    from relightable3dgaussian_amd import train_step as ts
    z = np.load(os.path.join(GOLDEN, "pipeline_reference_stage2.npz"))
    y = np.load(os.path.join(GOLDEN, "pipeline_reference_stage2_syn4.npz"))
    t = lambda k: torch.from_numpy(z[k])
    feat = torch.zeros(16, *z["gt"].shape[1:])
    feat[5:8], feat[8:11], feat[11:12], feat[12:15] = t("map_normal"), t("map_base_color"), t("map_roughness"), t("map_diffuse")
    gt, mask = t("gt"), t("mask")
    bcs, rs, ls = [float(v) for v in y["tb"][6:9]]
    # (the result dict holds rgb_to_srgb of the base-colour and diffuse maps: what the loss consumes, neilf.py:153-155)
    one = lambda **kw: float(ts.stage2_smoothness(feat, gt, mask, dict(ts.STAGE2_WEIGHTS, **kw), maps_are_srgb=True))
    assert abs(one(base_color_smooth=1.0) - bcs) < 2e-6 and abs(one(roughness_smooth=1.0) - rs) < 2e-6
    assert abs(one(light_smooth=1.0) - ls) < 2e-6
    assert min(bcs, rs, ls) > 1e-3                                    # (the terms are not trivially zero on this scene)
    w = ts.STAGE2_WEIGHTS_SYN4
    assert (w["base_color_smooth"], w["roughness_smooth"], w["light_smooth"]) == tuple(float(v) for v in y["lambdas"])
    both = float(ts.stage2_smoothness(feat, gt, mask, w, maps_are_srgb=True))
    # and on linear maps the curve is applied here: a map inside (0.0031308, 1) goes through 1.055 x^(1/2.4) - 0.055
    lin = feat.clone()
    lin[8:11] = ((feat[8:11] + 0.055) / 1.055).clamp_min(0) ** 2.4
    lin[12:15] = ((feat[12:15] + 0.055) / 1.055).clamp_min(0) ** 2.4
    ok = (feat[8:11] > 0.05) & (feat[8:11] < 0.999)
    assert float((ts.rgb_to_srgb(lin[8:11]) - feat[8:11])[ok].abs().max()) < 1e-5
    assert abs(both - (bcs + 0.5 * rs + ls)) < 3e-6
    # the whole objective differs from run_nerf.sh's by exactly these terms (same maps, same other lambdas)
    assert abs((float(y["loss"]) - float(z["loss"])) - (bcs + 0.5 * rs + ls)) < 3e-6


def test_checkpoint_defaults_follow_the_reference_flow():
    """(a) capture() without learning_rates writes the reference's training_setup rates (Optimizer.load_state_dict adopts the saved
    groups' hyper-parameters: lr 0 would freeze a reference GaussianModel restored from the file); (b) restore(pbr=True) of a
    15-entry stage-1 file zero-initialises the PBR parameters exactly as GaussianModel.create_from_ckpt does
    (scene/gaussian_model.py:381-403)."""
    from relightable3dgaussian_amd import checkpoint as ck
    r = ck.restore(os.path.join(GOLDEN, "checkpoint_reference_stage1.pth"))
    P = r.xyz.shape[0]
    holder = types.SimpleNamespace(
        xyz=r.xyz, normal=r.normal, scaling=r.scaling, rotation=r.rotation, opacity=r.opacity,
        shs=torch.cat([r.features_dc, r.features_rest], 1), _opt_order=("xyz", "normal", "scaling", "rotation", "opacity", "shs"),
        opt=types.SimpleNamespace(step_count=0, groups=[dict(exp_avg=torch.zeros_like(t), exp_avg_sq=torch.zeros_like(t)) for t in (
            r.xyz, r.normal, r.scaling, r.rotation, r.opacity, torch.cat([r.features_dc, r.features_rest], 1))]))
    captured, it = ck.capture(holder, 123, spatial_lr_scale=2.0)
    lrs = {g["name"]: g["lr"] for g in captured[13]["param_groups"]}
    assert lrs == {"xyz": 0.00016 * 2.0, "normal": 0.01, "rotation": 0.001, "scaling": 0.005, "opacity": 0.05, "f_dc": 0.0025,
                   "f_rest": 0.0025 / 20.0}
    # a step that trains with its own rates (scheduled xyz rate, the 10x smaller stage-2 rates of run_nerf.sh): those are written
    for g, lr in zip(holder.opt.groups, (3e-5, 0.001, 0.0005, 0.0001, 0.005, 0.00025)):
        g["lr"] = lr
    holder.opt.groups[5]["lr_tail"] = 0.00025 / 20.0
    lrs = {g["name"]: g["lr"] for g in ck.capture(holder, 123, spatial_lr_scale=2.0)[0][13]["param_groups"]}
    assert lrs == {"xyz": 3e-5, "normal": 0.001, "rotation": 0.0001, "scaling": 0.0005, "opacity": 0.005, "f_dc": 0.00025,
                   "f_rest": 0.00025 / 20.0}
    lrs = {g["name"]: g["lr"] for g in ck.capture(holder, 1, learning_rates={"xyz": 7.0})[0][13]["param_groups"]}
    assert lrs["xyz"] == 7.0 and lrs["normal"] == 0.001
    s2 = ck.restore(os.path.join(GOLDEN, "checkpoint_reference_stage1.pth"), pbr=True)
    assert s2.base_color.shape == (P, 3) and s2.roughness.shape == (P, 1) and s2.incidents_dc.shape == (P, 1, 3)
    assert s2.incidents_rest.shape == (P, 15, 3) and s2.visibility_rest.shape == (P, 15, 1)
    assert all(float(t.abs().max()) == 0.0 for t in (s2.base_color, s2.roughness, s2.incidents_dc, s2.incidents_rest))


def test_fused_stage2_iteration_host_logic_with_a_recording_library(monkeypatch):
    """The Python side of FusedStage2Step.__call__ with every C-ABI entry point and the rasterizer replaced by recorders
    (nothing runs on a GPU): the order of the calls of one iteration and the bounded forward from the second iteration on."""
    import contextlib
    import types
    from relightable3dgaussian_amd import _lib, fused_step, rasterizer_ops, shading_ops
    calls = []

    class Recorder:
        def __getattr__(self, name):
            def fn(*args):
                calls.append((name, args))
                return 1 if name == "r3dg_bounded_forward_supported" else 0
            return fn

    P, K, H, W = 6, 8, 4, 4
    z = torch.zeros
    monkeypatch.setattr(_lib, "lib", lambda: Recorder())
    monkeypatch.setattr(_lib, "current_stream", lambda: 0)
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: contextlib.nullcontext())

    class FakeStream:
        cuda_stream = 0

        def wait_stream(self, other):
            pass
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a: FakeStream())
    monkeypatch.setattr(fused_step, "update_visibility", lambda *a, **k: (torch.ones(P, K, 1), torch.ones(P, K, 3),
                                                                          torch.full((P, K, 1), 2.0), None))
    monkeypatch.setattr(shading_ops, "build_taps", lambda dirs, He, We, *a, **k: z(P * K * 3))
    monkeypatch.setattr(shading_ops, "_c", lambda t: t.contiguous())             # (its device check is not under test)
    begun = []

    class Pending:
        def finish(self, ordering_stream=None):
            return (17, z(H, W, dtype=torch.int32), z(3, H, W), z(1, H, W), z(1, H, W), z(16, H, W), z(3, H, W), z(3, H, W),
                    z(P, 1), z(P, dtype=torch.int32), z(64, dtype=torch.uint8), z(8, dtype=torch.uint8), z(8, dtype=torch.uint8))

    def begin(*a, **k):
        begun.append(k.get("capacity"))
        return Pending()
    monkeypatch.setattr(rasterizer_ops, "rasterize_gaussians_begin", begin)
    monkeypatch.setattr(rasterizer_ops, "rasterize_gaussians_backward", lambda *a, **k: (
        z(P, 3), z(P, 3), z(P, 1), z(P, 3), z(P, 16), z(P, 6), z(P, 16, 3), z(P, 3), z(P, 4)))
    monkeypatch.setattr(rasterizer_ops, "num_rendered_of", lambda geom, P_: torch.tensor(17))
    params = types.SimpleNamespace(xyz=z(P, 3), normal=z(P, 3), scaling=z(P, 3), rotation=z(P, 4), opacity=z(P, 1),
                                   features_dc=z(P, 1, 3), features_rest=z(P, 15, 3), base_color=z(P, 3), roughness=z(P, 1),
                                   incidents_dc=z(P, 1, 3), incidents_rest=z(P, 15, 3), env=z(1, 16, 32, 3))
    cam = types.SimpleNamespace(image_height=H, image_width=W, world_view_transform=torch.eye(4), full_proj_transform=torch.eye(4),
                                camera_center=z(3), tanfovx=0.5, tanfovy=0.5, cx=2.0, cy=2.0)
    step = fused_step.FusedStage2Step(params, K)
    calls.clear()
    for _ in range(3):
        step(cam, torch.ones(3), z(3, H, W))
    names = [c[0] for c in calls]
    assert begun[0] is None and begun[1] == begun[2] == step._capacity_for(17)      # two-phase first, bounded afterwards
    fwd, bwd = "r3dg_shade_forward_cached", "r3dg_shade_backward_cached"
    assert names.count(fwd) == 3 and names.count(bwd) == 3 and names.count("r3dg_adam_step") == 6
    one = names[names.index("r3dg_stage2_activate_with"):]
    order = [n for n in one if n in (fwd, "r3dg_stage2_pack_features", "r3dg_ssim_forward_pair", "r3dg_stage2_loss",
                                     "r3dg_stage2_unpack_gradients", bwd, "r3dg_stage2_activate_backward_with", "r3dg_adam_step")][:9]
    assert order == [fwd, "r3dg_stage2_pack_features", "r3dg_ssim_forward_pair", "r3dg_stage2_loss", "r3dg_adam_step",
                     "r3dg_stage2_unpack_gradients", bwd, "r3dg_stage2_activate_backward_with", "r3dg_adam_step"], order
    assert "r3dg_stage2_smooth_forward" not in names                          # run_nerf.sh's objective has no smoothness terms
    # above a million Gaussians (R3DG_EARLY_ADAM overrides the size rule) the SH group is NOT updated under the shading backward:
    # one Adam launch per iteration, all ten groups, behind the chain rule
    monkeypatch.setenv("R3DG_EARLY_ADAM", "0")
    calls.clear()
    step(cam, torch.ones(3), z(3, H, W))
    late = [c for c in calls if c[0] == "r3dg_adam_step"]
    assert len(late) == 1 and late[0][1][1] == 10 and calls[-1][0] in ("r3dg_adam_step", "r3dg_context_make_current")
    monkeypatch.delenv("R3DG_EARLY_ADAM")
    # ---- the Synthetic4Relight / DTU schedule (run_syn4.sh:22-42): smoothness terms on, every geometry rate 0 ----------------
    from relightable3dgaussian_amd import train_step
    feats = []
    monkeypatch.setattr(rasterizer_ops, "rasterize_gaussians_backward_features",
                        lambda P_, S_, H_, W_, gF, geom, R, binning, img, debug=False, active_features=None:
                        feats.append(tuple(active_features)) or z(P, 16))
    monkeypatch.setattr(rasterizer_ops, "rasterize_gaussians_backward", lambda *a, **k: pytest.fail("geometry backward ran"))
    frozen_lrs = dict(xyz=0.0, normal=0.0, scaling=0.0, rotation=0.0, opacity=0.0, shs=0.0, shs_rest=0.0, base_color=0.01,
                      roughness=0.01, incidents=0.001, incidents_rest=0.0001, env=0.1)
    step = fused_step.FusedStage2Step(params, K, lrs=frozen_lrs, loss_weights=train_step.STAGE2_WEIGHTS_SYN4)
    assert step.frozen_geometry and step.frozen == {"xyz", "normal", "scaling", "rotation", "opacity", "shs"}
    calls.clear()
    mask = torch.ones(1, H, W)
    for _ in range(2):
        step(cam, torch.ones(3), z(3, H, W), image_mask=mask)
    names = [c[0] for c in calls]
    # the smoothness terms: one streaming kernel behind the loss kernel (the three-pass formulation only under R3DG_SMOOTH_FUSED=0)
    assert names.count("r3dg_stage2_smooth_fused") == 2 and names.count("r3dg_stage2_smooth_forward") == 0
    assert names.index("r3dg_stage2_loss") < names.index("r3dg_stage2_smooth_fused")
    # pbr 2-4, base colour 8-10, roughness 11, diffuse light 12-14; the normal maps' gradient has no consumer
    assert feats == [(2, 3, 4, 8, 9, 10, 11, 12, 13, 14)] * 2
    adam = [c[1] for c in calls if c[0] == "r3dg_adam_step"]
    assert len(adam) == 2 and all(a[1] == 4 for a in adam)                     # ONE launch per iteration, four groups that train
    ab = [c[1] for c in calls if c[0] == "r3dg_stage2_activate_backward_with"][0]
    assert ab[2] is None and ab[19] is None and ab[24] == step.grads["base_color"].data_ptr()   # no geometry in or out
    sm = [c[1] for c in calls if c[0] == "r3dg_stage2_smooth_fused"][0]
    N_ = H * W
    assert sm[7] == mask.data_ptr() and abs(sm[8] - 1.0 / (3 * N_)) < 1e-9 and abs(sm[9] - 0.5 / (3 * N_)) < 1e-9
    loss_args = [c[1] for c in calls if c[0] == "r3dg_stage2_loss"][0]
    assert loss_args[10] == mask.data_ptr()
    # gradient slab: what is reduced under data parallelism is [flag, base colour, roughness, env] and [incidents] only
    assert step._bucket_a is None
    assert step._bucket_c.numel() == 4 + 4 * ((3 * P + 3) // 4) + 4 * ((P + 3) // 4) + 16 * 32 * 3
    assert step._bucket_b.numel() == 48 * P and step._bucket_b.data_ptr() == step.grads["incidents"].data_ptr()
    assert float(step.grads["xyz"].abs().max()) == 0.0 and float(step.grads["shs"].abs().max()) == 0.0
    # a single frozen group (not the whole geometry): the full backward runs, that group just gets no Adam launch
    monkeypatch.setattr(rasterizer_ops, "rasterize_gaussians_backward", lambda *a, **k: (
        z(P, 3), z(P, 3), z(P, 1), z(P, 3), z(P, 16), z(P, 6), z(P, 16, 3), z(P, 3), z(P, 4)))
    step = fused_step.FusedStage2Step(params, K, lrs=dict(xyz=0.0))
    assert not step.frozen_geometry and step.frozen == {"xyz"}
    calls.clear()
    step(cam, torch.ones(3), z(3, H, W))
    adam = [c[1] for c in calls if c[0] == "r3dg_adam_step"]
    assert [a[1] for a in adam] == [1, 8]                                      # shs early, then the other nine minus xyz
    with pytest.raises(RuntimeError):
        fused_step.FusedStage2Step(params, K, loss_weights={"no_such_term": 1.0})


def test_fused_stage2_iteration_spreads_the_fixed_ray_set_path_over_three_streams(monkeypatch):
    """The schedule of the single-GPU stage-2 iteration when the fixed-ray-set kernels are on and some Gaussians are off the
    rotated path, with recorders in place of the library, the rasterizer and the ray-set object -- nothing
    runs on a GPU.  From the second iteration on: the listed Gaussians' forward kernel gets the early-Adam stream; the rasterizer's
    geometry backward gets it too, the SH group's Adam behind it, the rotation back as well -- and (round 5) behind that the
    incident-light group's Adam and the rotation of the NEW coefficients for the next iteration, whose forward is told that the
    rotation is done; the main stream joins that stream once, in front of the shading forward; every join goes through the
    library's pooled events; the side streams are the process-wide ones, shared by a second step object."""
    import contextlib
    import types
    from relightable3dgaussian_amd import _lib, fused_step, rasterizer_ops, shading_ops
    events, rows = [], []

    class Recorder:
        def __getattr__(self, name):
            def fn(*args):
                events.append((name, args))
                return 1 if name == "r3dg_bounded_forward_supported" else 0
            return fn

    P, K, H, W = 6, 8, 4, 4
    z = torch.zeros
    streams = []

    class FakeStream:
        def __init__(self):
            self.cuda_stream = 1000 + len(streams)
            streams.append(self)

        def wait_stream(self, other):
            events.append(("torch.wait_stream", (self.cuda_stream, other.cuda_stream)))

    class FakeEvent:
        def record(self, stream):
            events.append(("event.record", (stream.cuda_stream,)))

    main = FakeStream()

    class FakeStreamContext:
        def __init__(self, s):
            self.s = s

        def __enter__(self):
            events.append(("enter", (self.s.cuda_stream,)))

        def __exit__(self, *a):
            events.append(("exit", (self.s.cuda_stream,)))
            return False

    def wait_event(ev):
        events.append(("main.wait_event", ()))
    main.wait_event = wait_event

    class FakeRaySet:
        n_invalid = 2

        def taps(self, He, We):
            events.append(("frs.taps", (He, We)))

        def rotate(self, incidents):
            events.append(("frs.rotate", ()))

        def forward(self, *a, uniform_area=None, leave_room=False, listed_stream=None, rotated=False, feature_rows=None):
            events.append(("frs.forward", (listed_stream, rotated, leave_room)))
            rows.append(feature_rows)

        def backward(self, *a, uniform_area=None, out_incidents=None, out_env=None, block_absmax=None, rotate_stream=None,
                     rotation_back=True):
            events.append(("frs.backward", (rotate_stream, rotation_back)))
            return z(P, 3), z(P, 1), z(P, 3), out_incidents, out_env

        def incident_chain(self, incidents, grad, m, v, lr, lr_tail, betas, eps, step, grad_scale=1.0, skip_flag=None):
            events.append(("frs.chain", (step, lr, lr_tail)))

    monkeypatch.setattr(_lib, "lib", lambda: Recorder())
    monkeypatch.setattr(_lib, "current_stream", lambda: 0)
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: FakeStreamContext(s))
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: FakeEvent())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a: main)
    monkeypatch.setattr(fused_step, "update_visibility", lambda *a, **k: (torch.ones(P, K, 1), torch.ones(P, K, 3),
                                                                          torch.full((P, K, 1), 2.0), None))
    monkeypatch.setattr(shading_ops, "build_taps", lambda dirs, He, We, *a, **k: z(P * K * 3))
    monkeypatch.setattr(shading_ops.FixedRaySet, "supported", staticmethod(lambda K_, M_, He, We: True))
    monkeypatch.setattr(shading_ops.FixedRaySet, "try_build", classmethod(lambda cls, normals, dirs, **k: FakeRaySet()))
    geometry_streams = []

    class Pending:
        def finish(self, ordering_stream=None):
            events.append(("raster.finish", (ordering_stream,)))
            return (17, z(H, W, dtype=torch.int32), z(3, H, W), z(1, H, W), z(1, H, W), z(16, H, W), z(3, H, W), z(3, H, W),
                    z(P, 1), z(P, dtype=torch.int32), z(64, dtype=torch.uint8), z(8, dtype=torch.uint8), z(8, dtype=torch.uint8))

    def begin(*a, **k):
        events.append(("raster.begin", (k.get("ordering_stream"),)))
        return Pending()

    def backward(*a, **k):
        geometry_streams.append(k.get("geometry_stream"))
        events.append(("raster.backward", ()))
        return (z(P, 3), z(P, 3), z(P, 1), z(P, 3), z(P, 16), z(P, 6), z(P, 16, 3), z(P, 3), z(P, 4))
    monkeypatch.setattr(rasterizer_ops, "rasterize_gaussians_begin", begin)
    monkeypatch.setattr(rasterizer_ops, "rasterize_gaussians_backward", backward)
    monkeypatch.setattr(rasterizer_ops, "num_rendered_of", lambda geom, P_: torch.tensor(17))
    params = types.SimpleNamespace(xyz=z(P, 3), normal=z(P, 3), scaling=z(P, 3), rotation=z(P, 4), opacity=z(P, 1),
                                   features_dc=z(P, 1, 3), features_rest=z(P, 15, 3), base_color=z(P, 3), roughness=z(P, 1),
                                   incidents_dc=z(P, 1, 3), incidents_rest=z(P, 15, 3), env=z(1, 16, 32, 3))
    cam = types.SimpleNamespace(image_height=H, image_width=W, world_view_transform=torch.eye(4), full_proj_transform=torch.eye(4),
                                camera_center=z(3), tanfovx=0.5, tanfovy=0.5, cx=2.0, cy=2.0)
    step = fused_step.FusedStage2Step(params, K)
    order_stream = step._order_stream
    marks = []
    for _ in range(3):
        marks.append(len(events))
        step(cam, torch.ones(3), z(3, H, W))
    assert isinstance(step._frs, FakeRaySet)
    early = step._adam_stream
    assert early is not None and early is not order_stream and early is not main
    # ---- the third iteration (bounded forward, fixed-ray-set path, incident-light chain of round 5) ---------------------------
    it, args = [e[0] for e in events[marks[2]:]], [e[1] for e in events[marks[2]:]]
    pos = {n: it.index(n) for n in ("r3dg_stage2_activate_with", "raster.begin", "frs.forward",
                                    "raster.finish", "raster.backward", "r3dg_stage2_unpack_gradients", "frs.backward",
                                    "r3dg_stage2_activate_backward_with", "frs.chain")}
    assert sorted(pos, key=pos.get) == ["r3dg_stage2_activate_with", "raster.begin", "frs.forward", "raster.finish", "raster.backward",
                                        "r3dg_stage2_unpack_gradients", "frs.backward", "r3dg_stage2_activate_backward_with",
                                        "frs.chain"]
    # the coefficient rotation is NOT at the top of the iteration any more, nor a launch of its own: the previous iteration's
    # incident-light chain (rotation back + the group's Adam + rotation forward: ONE kernel on the early stream, queued by
    # optimizer_step behind the other groups' Adam) left the rotation of the current coefficients in the ray set
    assert it.count("frs.rotate") == 0 and step._rotation_is_current()
    # no pack kernel on this path: the activations and the shading kernels write the feature rows between them, and the
    # light-smoothness sum comes from the unpack kernel (its last argument)
    assert "r3dg_stage2_pack_features" not in it and rows[-1] is step.features
    act = args[pos["r3dg_stage2_activate_with"]]
    assert act[-6] == step.features.data_ptr()
    # the softplus of the environment texture and the loss-sum reset ride in the activation launch: texture in, activated texture
    # out, the sums to zero -- and the texture's chain rule rides in the activation chain rule's launch (consume = 1)
    assert act[-5:] == (step.env.numel(), step.env.data_ptr(), step._env_c.data_ptr(), step.sums.data_ptr(), step.sums.numel())
    bwd_args = args[pos["r3dg_stage2_activate_backward_with"]]
    assert bwd_args[-9:-7] == (16, 32) and bwd_args[-7] == step.env.data_ptr() and bwd_args[-6] == step._env_c.data_ptr()
    assert bwd_args[-3] == step.grads["env"].data_ptr() and bwd_args[-2] == step.sums[4].data_ptr() and bwd_args[-1] == 1
    assert "r3dg_stage2_env_backward" not in it and "r3dg_stage2_pbr_srgb" not in it and "r3dg_stage2_normals_srgb" in it
    assert args[pos["r3dg_stage2_unpack_gradients"]][-1] == step.sums[3].data_ptr()
    joins = [(i, a) for i, (n, a) in enumerate(zip(it, args)) if n == "r3dg_stream_wait_stream"]
    assert joins[0][1] == (early.cuda_stream, main.cuda_stream)              # fork at the top of the iteration
    # the main stream joins the early stream in front of the shading forward (behind it: the previous iteration's incident-light
    # chain) and behind the listed Gaussians' forward kernel; there is no join at the end of the iteration any more
    main_joins = [i for i, a in joins if a == (main.cuda_stream, early.cuda_stream)]
    assert len(main_joins) == 2 and pos["raster.begin"] < main_joins[0] < pos["frs.forward"] < main_joins[1] < pos["raster.finish"]
    # the ordering stream (next projection reads the SH colour coefficients) is ordered behind the SH group's Adam when that is
    # queued -- before the rotation back, the incident-light group's Adam and the next rotation go to the early stream
    order_joins = [i for i, a in joins if a == (order_stream.cuda_stream, early.cuda_stream)]
    adam = [i for i, n in enumerate(it) if n == "r3dg_adam_step"]
    assert len(order_joins) == 1 and len(adam) == 2
    assert adam[0] < order_joins[0] < pos["frs.backward"] < pos["r3dg_stage2_activate_backward_with"] < adam[1] < pos["frs.chain"]
    assert args[pos["raster.begin"]] == (order_stream,) and args[pos["raster.finish"]] == (order_stream,)
    listed, rotated, leave_room = args[pos["frs.forward"]]
    # the chain as one kernel ends before the projection starts: the shading forward keeps its one-workgroup-per-CU cap beside the
    # ordering chain (it loses it behind a chain of three launches: R3DG_INCIDENT_CHAIN_KERNEL=0)
    assert listed is early and rotated is True and leave_room is True
    assert geometry_streams[-1] is early                                       # geometry backward beside the listed backward
    # the main shading backward leaves the coefficient gradient in the rotated frame: the chain kernel rotates it back
    assert args[pos["frs.backward"]] == (early, False)
    assert "main.wait_event" in it[pos["frs.backward"]:pos["r3dg_stage2_activate_backward_with"]]     # geometry joined by its event
    assert "torch.wait_stream" not in it                                       # no per-call event objects on the hot path
    # Adam: the SH group inside the early stream's context, the other groups on the main stream; the chain kernel behind THAT
    # launch (a join early <- main in front of it), inside the early stream's context, with the iteration's own step count and the
    # incident-light group's two learning rates
    assert it[adam[0] - 1] == "enter" and args[adam[0] - 1] == (early.cuda_stream,)
    assert "enter" not in it[pos["frs.backward"]:adam[1]]
    assert it[pos["frs.chain"] - 1] == "enter" and args[pos["frs.chain"] - 1] == (early.cuda_stream,)
    assert it[pos["frs.chain"] - 2] == "r3dg_stream_wait_stream" and args[pos["frs.chain"] - 2] == (early.cuda_stream, main.cuda_stream)
    grp = step.opt.groups[step._opt_order.index("incidents")]
    assert args[pos["frs.chain"]] == (step.opt.step_count, grp["lr"], grp["lr_tail"])
    # the small view-independent jobs (softplus, sum reset) ride in the activation launch and the accumulator slab needs no zero
    # fill: nothing but the fork enters the early stream's context at the top of the iteration
    assert "enter" not in it[:pos["frs.forward"]]
    # anybody outside the iteration is ordered behind the chain before it sees the coefficients; an edited tensor is re-rotated
    assert step._early_pending
    n_joins = sum(1 for e in events if e[0] == "r3dg_stream_wait_stream")
    _ = step.incidents
    assert not step._early_pending and sum(1 for e in events if e[0] == "r3dg_stream_wait_stream") == n_joins + 1
    step.incidents.add_(1.0)
    assert not step._rotation_is_current()
    marks.append(len(events))
    step(cam, torch.ones(3), z(3, H, W))
    it4 = [e[0] for e in events[marks[3]:]]
    assert it4.count("frs.rotate") == 1 and it4.index("frs.rotate") < it4.index("r3dg_stage2_activate_with")
    # R3DG_INCIDENT_CHAIN_KERNEL=0: the chain as three launches -- rotation back by the shading backward's call, the group's Adam and
    # the next rotation behind the other groups' Adam -- and the shading forward uncapped behind it
    monkeypatch.setenv("R3DG_INCIDENT_CHAIN_KERNEL", "0")
    three = fused_step.FusedStage2Step(params, K)
    for _ in range(3):
        mark = len(events)
        three(cam, torch.ones(3), z(3, H, W))
    it5, args5 = [e[0] for e in events[mark:]], [e[1] for e in events[mark:]]
    assert "frs.chain" not in it5 and it5.count("frs.rotate") == 1 and it5.count("r3dg_adam_step") == 3
    assert args5[it5.index("frs.backward")] == (early, True) and args5[it5.index("frs.forward")][2] is False
    assert it5.index("r3dg_stage2_activate_backward_with") < it5.index("frs.rotate")
    monkeypatch.delenv("R3DG_INCIDENT_CHAIN_KERNEL")
    # ---- a second step object gets the same side streams -------------------------------------------------------------------
    other = fused_step.FusedStage2Step(params, K)
    other(cam, torch.ones(3), z(3, H, W))
    other(cam, torch.ones(3), z(3, H, W))
    assert other._order_stream is order_stream and other._adam_stream is early


def test_fused_stage2_data_parallel_iteration_issues_its_buckets_from_the_streams_that_finish_them(monkeypatch):
    """The same recorders around the DATA-PARALLEL iteration (two ranks pretended): bucket A's all-reduce is issued inside the
    early stream's context right behind the geometry backward that runs there, bucket C's from the main stream; bucket B (round 6)
    is the ray set's ROTATED-frame gradient buffer, issued from the main stream behind C (no rotation back in front of it); the deferred incident-light update waits for B at the top of the next iteration's forward and closes the
    group with ONE chain kernel (rotation back of the reduced gradient + Adam + rotation of the new coefficients), so the forward
    is told that its rotated coefficients are in place.  R3DG_DP_CHAIN=0: the three-launch path of rounds 2-5."""
    import contextlib
    import types
    from relightable3dgaussian_amd import _lib, fused_step, rasterizer_ops, shading_ops
    events = []
    depth = []            # stack of stream contexts

    class Recorder:
        def __getattr__(self, name):
            def fn(*args):
                events.append((name, tuple(depth)))
                return 1 if name == "r3dg_bounded_forward_supported" else 0
            return fn

    P, K, H, W = 6, 8, 4, 4
    z = torch.zeros
    n_streams = [0]

    class FakeStream:
        def __init__(self):
            n_streams[0] += 1
            self.cuda_stream = 2000 + n_streams[0]

        def wait_stream(self, other):
            pass

        def wait_event(self, ev):
            pass

    main = FakeStream()

    class Ctx:
        def __init__(self, s):
            self.s = s

        def __enter__(self):
            depth.append(self.s.cuda_stream)

        def __exit__(self, *a):
            depth.pop()
            return False

    class Handle:
        def __init__(self, tag):
            self.tag = tag

        def wait(self):
            events.append(("wait " + self.tag, tuple(depth)))

    class FakeRaySet:
        n_invalid = 1

        def __init__(self):
            self.dcprime = z(P, 48)

        def dcprime_rows(self):
            return self.dcprime.view(P, 16, 3)

        def taps(self, He, We):
            pass

        def rotate(self, incidents):
            events.append(("frs.rotate", tuple(depth)))

        def forward(self, *a, listed_stream=None, rotated=False, **k):
            events.append(("frs.forward rotated=%s" % rotated, tuple(depth)))

        def backward(self, *a, out_incidents=None, out_env=None, rotate_stream=None, rotation_back=True, **k):
            events.append(("frs.backward rotation_back=%s into_dcprime=%s" % (
                rotation_back, out_incidents is not None and out_incidents.data_ptr() == self.dcprime.data_ptr()),
                (rotate_stream.cuda_stream if rotate_stream is not None else None,)))
            return z(P, 3), z(P, 1), z(P, 3), out_incidents, out_env

        def incident_chain(self, *a, listed_in_dcprime=False, **k):
            events.append(("frs.incident_chain listed_in_dcprime=%s grad_scale=%s" % (listed_in_dcprime, a[9]), tuple(depth)))

    the_ray_set = FakeRaySet()
    monkeypatch.setattr(fused_step, "_world_of", lambda group: (2, True))
    monkeypatch.setattr(_lib, "lib", lambda: Recorder())
    monkeypatch.setattr(_lib, "current_stream", lambda: 0)
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "stream", lambda s: Ctx(s))
    monkeypatch.setattr(torch.cuda, "Stream", lambda device=None: FakeStream())
    monkeypatch.setattr(torch.cuda, "Event", lambda *a, **k: types.SimpleNamespace(record=lambda s: None))
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *a: main)
    monkeypatch.setattr(fused_step, "update_visibility", lambda *a, **k: (torch.ones(P, K, 1), torch.ones(P, K, 3),
                                                                          torch.full((P, K, 1), 2.0), None))
    monkeypatch.setattr(shading_ops, "build_taps", lambda dirs, He, We, *a, **k: z(P * K * 3))
    monkeypatch.setattr(shading_ops.FixedRaySet, "supported", staticmethod(lambda K_, M_, He, We: True))
    monkeypatch.setattr(shading_ops.FixedRaySet, "try_build", classmethod(lambda cls, normals, dirs, **k: the_ray_set))

    class Pending:
        def finish(self, ordering_stream=None):
            return (17, z(H, W, dtype=torch.int32), z(3, H, W), z(1, H, W), z(1, H, W), z(16, H, W), z(3, H, W), z(3, H, W),
                    z(P, 1), z(P, dtype=torch.int32), z(64, dtype=torch.uint8), z(8, dtype=torch.uint8), z(8, dtype=torch.uint8))
    monkeypatch.setattr(rasterizer_ops, "rasterize_gaussians_begin", lambda *a, **k: Pending())

    def backward(*a, **k):
        events.append(("raster.backward geometry_stream=%s" % getattr(k.get("geometry_stream"), "cuda_stream", None), tuple(depth)))
        return (z(P, 3), z(P, 3), z(P, 1), z(P, 3), z(P, 16), z(P, 6), z(P, 16, 3), z(P, 3), z(P, 4))
    monkeypatch.setattr(rasterizer_ops, "rasterize_gaussians_backward", backward)
    monkeypatch.setattr(rasterizer_ops, "num_rendered_of", lambda geom, P_: torch.tensor(17))
    buckets = {}

    def all_reduce(flat, group=None, async_op=False):
        tag = buckets.get(flat.data_ptr(), "?")
        events.append(("all_reduce " + tag, tuple(depth)))
        return Handle(tag)
    monkeypatch.setattr(torch.distributed, "all_reduce", all_reduce)
    params = types.SimpleNamespace(xyz=z(P, 3), normal=z(P, 3), scaling=z(P, 3), rotation=z(P, 4), opacity=z(P, 1),
                                   features_dc=z(P, 1, 3), features_rest=z(P, 15, 3), base_color=z(P, 3), roughness=z(P, 1),
                                   incidents_dc=z(P, 1, 3), incidents_rest=z(P, 15, 3), env=z(1, 16, 32, 3))
    cam = types.SimpleNamespace(image_height=H, image_width=W, world_view_transform=torch.eye(4), full_proj_transform=torch.eye(4),
                                camera_center=z(3), tanfovx=0.5, tanfovy=0.5, cx=2.0, cy=2.0)
    step = fused_step.FusedStage2Step(params, K, process_group=object())
    assert step.dp and step.world == 2
    buckets.update({step._bucket_a.data_ptr(): "A", step._bucket_c.data_ptr(): "C", step._bucket_b.data_ptr(): "B (slab)",
                    the_ray_set.dcprime.data_ptr(): "B"})
    for _ in range(3):
        step(cam, torch.ones(3), z(3, H, W))
    early = step._adam_stream.cuda_stream
    names = [e[0] for e in events]
    start = len(names) - 1 - names[::-1].index("r3dg_stage2_activate_with")        # the third iteration
    it = events[start:]
    kinds = ("all_reduce", "wait", "frs.rotate", "frs.forward", "frs.backward", "frs.incident_chain", "raster.backward")
    seq = [(n, d) for n, d in it if n.split()[0] in kinds]
    assert seq == [
        ("wait B", ()),                                     # flush(): the previous iteration's incident-light update ...
        ("frs.incident_chain listed_in_dcprime=True grad_scale=0.5", ()),      # ... ONE kernel, 1 / world inside
        ("frs.forward rotated=True", ()),                   # the chain left the rotated coefficients in place
        ("raster.backward geometry_stream=%d" % early, ()),
        ("all_reduce A", (early,)),                         # behind the geometry backward, on its stream
        ("wait A", (early,)),                               # the SH group's early Adam waits there, not on the main stream
        ("frs.backward rotation_back=False into_dcprime=True", (early,)),
        ("all_reduce C", ()),                               # C first: it is the bucket the main stream waits for
        ("all_reduce B", ()),                               # the rotated-frame buffer, from the main stream (no rotation back)
        ("wait C", ()),
    ], seq
    assert "frs.rotate" not in names
    # R3DG_DP_CHAIN=0: rotation back on the early stream, bucket B = the slab's incident-light region behind it, Adam + the forward's
    # own rotation in the next iteration
    monkeypatch.setenv("R3DG_DP_CHAIN", "0")
    del events[:]
    for _ in range(2):
        step(cam, torch.ones(3), z(3, H, W))
    names = [e[0] for e in events]
    it = events[len(names) - 1 - names[::-1].index("r3dg_stage2_activate_with"):]
    seq = [(n, d) for n, d in it if n.split()[0] in kinds]
    assert seq == [
        ("wait B (slab)", ()),
        ("frs.forward rotated=False", ()),
        ("raster.backward geometry_stream=%d" % early, ()),
        ("all_reduce A", (early,)),
        ("wait A", (early,)),
        ("frs.backward rotation_back=True into_dcprime=False", (early,)),
        ("all_reduce C", ()),
        ("all_reduce B (slab)", (early,)),
        ("wait C", ()),
    ], seq
    monkeypatch.delenv("R3DG_DP_CHAIN")
    # ---- per-bucket attribution (bench.py: measure_comm): every probed bucket gets a `ready` event on the stream that issues it
    # (no stream of its own: that perturbed the schedule); the waits carry the bucket's name; comm_table() reads the collective's own
    # time from the work handle where the backend times it, else the ready -> released interval -----------------------------------
    clock = [0.0]

    class TimedEvent:
        def __init__(self, *a, **k):
            self.t = None

        def record(self, s=None):
            clock[0] += 0.25
            self.t = clock[0]

        def elapsed_time(self, other):
            return other.t - self.t
    monkeypatch.setattr(torch.cuda, "Event", TimedEvent)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    step.measure_comm, step.comm_probe_every = True, 1
    del events[:]
    step(cam, torch.ones(3), z(3, H, W))
    step.flush()
    waits = [(n, d) for n, d in events if n.startswith("wait")]
    assert ("wait A", (early,)) in waits and ("wait C", ()) in waits and ("wait B", ()) in waits     # no stream of its own
    total, by_bucket = step.exposed_comm_ms(split=True)
    assert set(by_bucket) == {"B", "C"} and abs(total - sum(by_bucket.values())) < 1e-9       # (A is waited for on the side stream)
    table = step.comm_table()
    assert set(table) == {"A", "B", "C"}
    assert table["A"]["MB"] == round(step._bucket_a.numel() * 4 / 1e6, 2) and table["A"]["ready_us"] == 0.0
    for row in table.values():
        # (the stand-in handle has no _get_duration: the ready -> released interval stands in, as over gloo)
        assert row["collective_ms"] > 0 and row["bus_GBs"] is not None and row["released_us"] > row["ready_us"]
        assert row["timed_by"].startswith("ready -> released")
    assert step.comm_table() is None                                    # read once
    step.measure_comm = False
    # ---- R3DG_DP_BUCKETS=1 (the A/B of message size against overlap): ONE all-reduce of the whole slab behind the backward, every
    # group's Adam in one launch behind it, nothing deferred into the next iteration ------------------------------------------------
    monkeypatch.setenv("R3DG_DP_BUCKETS", "1")
    one = fused_step.FusedStage2Step(params, K, process_group=object())
    assert one._single_bucket and one._bucket_all.numel() == one.grad_flat.numel()
    buckets[one._bucket_all.data_ptr()] = "ALL"        # (same address as bucket A: the slab starts there)
    calls_before = len(events)
    for _ in range(2):
        one(cam, torch.ones(3), z(3, H, W))
    names = [e[0] for e in events[calls_before:]]
    last = names[len(names) - 1 - names[::-1].index("r3dg_stage2_activate_with"):]
    seq = [n for n in last if n.split()[0] in ("all_reduce", "wait", "r3dg_adam_step", "frs.backward")]
    assert seq == ["frs.backward rotation_back=True into_dcprime=False", "all_reduce ALL", "wait ALL", "r3dg_adam_step"], seq
    assert one._pending_b is None


def test_pytorch_rendering_equation_restatement_equals_the_oracle():
    """train_step.rendering_equation_pytorch (the reference's pure-PyTorch shading integral, restated for bench.py's "reference
    loop shape" rows) against oracle/shading.py, which is pinned to the reference's own function: values and autograd gradients."""
    from oracle import shading
    from relightable3dgaussian_amd import sampling, train_step as ts
    g = torch.Generator().manual_seed(0)
    P, K, He = 60, 24, 8
    n = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1)
    dirs, areas = sampling.fibonacci_sphere_sampling(n, K)
    vis = (torch.rand(P, K, 1, generator=g) > 0.3).float()
    leaves = [torch.rand(P, 3, generator=g), 0.1 + 0.8 * torch.rand(P, 1, generator=g), torch.randn(P, 3, generator=g),
              0.3 * torch.randn(P, 16, 3, generator=g), torch.rand(He, 2 * He, 3, generator=g)]
    gp, gd = torch.randn(P, 3, generator=g), torch.randn(P, 3, generator=g)
    grads = []
    for fn in (lambda b, r, v, i, e: ts.rendering_equation_pytorch(b, r, n, v, i, e, vis, dirs, areas)[:2],
               lambda b, r, v, i, e: (lambda o: (o["pbr"], o["diffuse_light"]))(shading.rendering_equation(b, r, n, v, i, e, vis, dirs, areas))):
        ls = [t.clone().requires_grad_(True) for t in leaves]
        pbr, diff = fn(*ls)
        ((pbr * gp).sum() + (diff * gd).sum()).backward()
        grads.append([pbr.detach(), diff.detach()] + [t.grad for t in ls])
    for a, b in zip(*grads):
        assert float((a - b).abs().max()) <= 1e-6 + 2e-5 * float(b.abs().max())
