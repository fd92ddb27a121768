"""Data-parallel path of the fused stage-2 iteration: two ranks (two processes sharing the one GPU of the test box,
`gloo` backend on device tensors -- RCCL refuses two ranks on one device) render different cameras; after the bucketed
all-reduces (three buckets, the last one deferred into the next iteration) both must hold the SAME summed gradients,
equal to the sum of two single-process backward passes, and stay bit-identical replicas after the Adam steps."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _make(dev, P=2500, res=128, K=8, seed=11):
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams, render_stage1
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    torch.manual_seed(1234)
    scene = syn.make_scene(P=P, seed=seed, stage2=True, scale_log_mean=-3.2)
    cams = [c.to(dev) for c in syn.orbit_cameras(8, width=res, height=res)[:2]]
    bg = torch.tensor([1.0, 1.0, 1.0], device=dev)
    params = GaussianParams(scene, dev, True)
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=P, seed=seed, stage2=False, scale_log_mean=-3.2), dev, False)
        gts = [render_stage1(teacher, c, bg)[2].clone() + 0.1 for c in cams]
    return params, cams, bg, gts, K, FusedStage2Step


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    params, cams, bg, gts, K, FusedStage2Step = _make(dev)
    step = FusedStage2Step(params, K, lr=1e-3, loss_weights={"normal": 0.01})     # every parameter group gets a gradient
    assert step.world == 2
    step.forward_backward(cams[rank], bg, gts[rank])
    step.optimizer_step()                    # waits for buckets A and C; the incident-light bucket B stays in flight
    assert step._pending_b is not None
    step.flush()
    torch.cuda.synchronize()
    grads = {k: v.detach().cpu().clone() for k, v in step.grads.items()}      # all-reduced SUMS over the two ranks
    step(cams[rank], bg, gts[rank])          # a second full iteration on the updated parameters
    step.flush()
    torch.cuda.synchronize()
    pars = {k: getattr(step, k).detach().cpu().clone() for k in ("xyz", "shs", "incidents", "env", "opacity")}
    torch.save(dict(grads=grads, pars=pars, visibility=step.visibility.cpu()), os.path.join(out_dir, "rank%d.pt" % rank))
    dist.destroy_process_group()


def test_fused_step_two_ranks(tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "rank0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "rank1.pt"))
    for k in r0["grads"]:
        assert torch.equal(r0["grads"][k], r1["grads"][k]), "averaged gradient differs between ranks: " + k
    for k in r0["pars"]:
        assert torch.equal(r0["pars"][k], r1["pars"][k]), "replicas diverged: " + k
    # single process: mean of the two cameras' gradients
    dev = torch.device("cuda", 0)
    params, cams, bg, gts, K, FusedStage2Step = _make(dev)
    single = FusedStage2Step(params, K, lr=1e-3, loss_weights={"normal": 0.01})
    # the ranks traced half of the ray bundles each and all-gathered them (train_step.update_visibility)
    assert torch.equal(r0["visibility"], single.visibility.cpu()) and torch.equal(r1["visibility"], r0["visibility"])
    acc = None
    for i in range(2):
        single.forward_backward(cams[i], bg, gts[i])
        g = {k: v.detach().clone() for k, v in single.grads.items()}
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
    for k, v in acc.items():
        want = v.cpu()                       # the ranks hold the SUM; 1/world is applied inside the Adam kernel
        got = r0["grads"][k]
        scale = float(want.abs().max())
        err = float((got - want).abs().max())
        assert err <= 1e-4 * scale + 1e-9, "%s: err %.3e scale %.3e" % (k, err, scale)
        assert scale > 0, k


def _densify_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams, render_stage1
    from relightable3dgaussian_amd.fused_step import FusedStage1Step
    torch.manual_seed(4321)
    P, res = 3000, 128
    scene = syn.make_scene(P=P, seed=21, stage2=False, scale_log_mean=-3.0)
    cams = [c.to(dev) for c in syn.orbit_cameras(8, width=res, height=res)[:4]]
    bg = torch.ones(3, device=dev)
    params = GaussianParams(scene, dev, False)
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=P, seed=21, stage2=False, scale_log_mean=-3.0), dev, False)
        gts = [render_stage1(teacher, c, bg)[2].clone() * 0.8 for c in cams]
    step = FusedStage1Step(params, lr=1e-3)
    assert step.world == 2
    step.enable_densification()
    for i in range(3):                                   # each rank renders its own views
        v = (2 * i + rank) % 4
        step(cams[v], bg, gts[v])
    local_denom = step.stats.denom.clone()
    # the threshold must be the same number on both ranks: take it from a reduced COPY of the statistics
    # (densify_and_prune reduces the real ones itself)
    tmp = step.stats._slab[:4].clone()
    dist.all_reduce(tmp)
    mean_grad = tmp[0] / tmp[2].clamp_min(1)
    thr = float(mean_grad[mean_grad > 0].median())
    reduced_denom = tmp[2].clone()
    info = step.densify_and_prune(thr, 0.005, 2.6, 20, 1e9, percent_dense=0.03,
                                  generator=torch.Generator(device=dev).manual_seed(77))
    v = rank % 4
    step(cams[v], bg, gts[v])
    torch.cuda.synchronize()
    pars = {k: getattr(step, k).detach().cpu().clone() for k in step._opt_order}
    moms = {k: step.opt.groups[i]["exp_avg"].detach().cpu().clone() for i, k in enumerate(step._opt_order)}
    torch.save(dict(pars=pars, moms=moms, rows=info["rows_out"], cloned=info["cloned"], split=info["split"],
                    local_denom=local_denom.cpu(), reduced_denom=reduced_denom.cpu()),
               os.path.join(out_dir, "dens%d.pt" % rank))
    dist.destroy_process_group()


def test_densification_keeps_replicas_identical(tmp_path):
    """Data-parallel stage-1 training with a densify_and_prune in the middle: per-rank statistics (own views, own
    gradients) are reduced before the decisions and the split draws from identically seeded generators, so both ranks
    end with the same number of rows and bit-identical parameters / Adam moments, also after a further iteration."""
    mp.spawn(_densify_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "dens0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "dens1.pt"))
    assert r0["rows"] == r1["rows"] and r0["cloned"] == r1["cloned"] > 0 and r0["split"] == r1["split"]
    assert not torch.equal(r0["local_denom"], r1["local_denom"])             # the ranks really saw different views
    assert torch.equal(r0["reduced_denom"], r1["reduced_denom"])
    assert torch.equal(r0["reduced_denom"], r0["local_denom"] + r1["local_denom"])
    for k in r0["pars"]:
        assert r0["pars"][k].shape[0] == r0["rows"]
        assert torch.equal(r0["pars"][k], r1["pars"][k]), "replicas diverged: " + k
        assert torch.equal(r0["moms"][k], r1["moms"][k]), "Adam moments diverged: " + k


def _bench(extra, env, timeout=420):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + extra, capture_output=True, text=True,
                       timeout=timeout, env=e, stdin=subprocess.DEVNULL, cwd="/tmp")
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


SMALL = ["--steps", "3", "--warmup", "1", "--points", "20000", "--res", "160", "--sample-num", "16", "--no-cpu-baseline",
         "--relight-frames", "0", "--no-other-configs", "--repeats", "0"]


def _with_relight(flags, frames="3", samples="32"):
    out = list(flags)
    out[out.index("--relight-frames") + 1] = frames
    return out + ["--relight-samples", samples]


def test_bench_gpus_2_runs_two_ranks_on_the_gpu():
    """The driver's literal `python bench.py --gpus 2 ...` (no launcher): two ranks, real kernels, one JSON line with
    n_gpus 2 that carries the WHOLE metric -- train iters/s, the view-sharded relight FPS (frames rank::world, visibility traced
    in halves + one all-gather) and the per-bucket attribution of the gradient all-reduces.  The test box has ONE GPU, so the
    ranks share it through the gloo test backend (RCCL refuses two ranks on one device); the RCCL variant below runs wherever
    two devices exist."""
    r, doc = _bench(["--gpus", "2"] + _with_relight(SMALL), {"R3DG_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert doc["n_gpus"] == 2 and doc["value"] > 0 and "dp2" in doc["config"]["parallelism"]
    assert sum(1 for x in r.stdout.splitlines() if x.startswith("{")) == 1
    assert doc["relight_fps"] > 0 and doc["relight"]["frames_per_rank"] == 3
    assert doc["relight"]["per_rank_fps_min"] <= doc["relight"]["per_rank_fps_max"]
    assert doc["dp_buckets"] == 3 and set(doc["comm_buckets"]) == {"A", "B", "C"}
    for row in doc["comm_buckets"].values():
        assert row["MB"] > 0 and row["collective_ms"] > 0 and row["bus_GBs"] > 0
    assert set(doc["exposed_comm_ms_by_bucket"]) <= {"A", "B", "C"} and doc["exposed_comm_ms"] is not None


def test_bench_gpus_2_one_bucket_knob():
    """R3DG_DP_BUCKETS=1: the same launch with the whole gradient slab as ONE all-reduce (the A/B of message size against
    overlap the first real multi-GPU run is meant to make)."""
    r, doc = _bench(["--gpus", "2"] + SMALL, {"R3DG_DIST_BACKEND": "gloo", "R3DG_DP_BUCKETS": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert doc["n_gpus"] == 2 and doc["value"] > 0 and doc["dp_buckets"] == 1 and set(doc["comm_buckets"]) == {"ALL"}


def test_bench_config_presets_name_the_other_baseline_workloads():
    """`--config dtu4` = BASELINE configs[3] (1600x1200, run_dtu.sh objective, sample_num 32), with later flags still
    overriding: run small here, two ranks."""
    flags = [f for f in SMALL]
    for k in ("--res", "--sample-num"):
        i = flags.index(k)
        del flags[i:i + 2]
    r, doc = _bench(["--gpus", "2", "--config", "dtu4", "--width", "160", "--height", "120"] + flags, {"R3DG_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert doc["n_gpus"] == 2 and "160x120" in doc["config"]["workload"] and "run_syn4.sh" in doc["config"]["workload"]
    assert "K=32" in doc["config"]["workload"]
    assert set(doc["comm_buckets"]) == {"B", "C"}               # frozen geometry: no SH colour bucket


def _relight_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    frames = _relight_frames(dev, range(rank, 6, world))
    torch.save(frames, os.path.join(out_dir, "relight%d.pt" % rank))
    dist.destroy_process_group()


def _relight_frames(dev, which):
    from relightable3dgaussian_amd import relight, synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams
    scene = syn.make_scene(P=3001, seed=5, stage2=True, scale_log_mean=-3.2)
    cams = [c.to(dev) for c in syn.orbit_cameras(6, width=112, height=96)]
    params = GaussianParams(scene, dev, True)
    envmap = (3.0 * torch.rand(32, 64, 3, generator=torch.Generator().manual_seed(7)) ** 2).to(dev)
    r = relight.RelightRenderer(params, envmap, 24)           # (initialised group: ray bundles sharded, one all-gather)
    bg = torch.zeros(3, device=dev)
    out = {}
    for f in which:
        res = r.frame(cams[f], bg, outputs=("pbr_env", "render_env"))
        out[f] = {k: res[k].cpu().clone() for k in ("pbr_env", "render_env", "render", "feature", "num_contrib")}
        out[f]["num_rendered"] = int(res["num_rendered"])
    out["visibility"] = r.visibility.cpu().clone()
    return out


def test_sharded_relight_frames_equal_single_rank_frames(tmp_path):
    """BASELINE configs[4] "view-sharded render": rank r of W renders frames r, r + W, ... of the trajectory on its replica
    (relighting.py:114-185 walks them in one process).  Two ranks on the one test GPU: every frame -- composites, the S=28
    feature image, the contributor counts -- is bit-identical to the frame a single process renders, and the all-gathered
    visibility equals the single-process trace."""
    mp.spawn(_relight_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    single = _relight_frames(torch.device("cuda", 0), range(6))
    seen = set()
    for rank in range(2):
        got = torch.load(os.path.join(tmp_path, "relight%d.pt" % rank))
        assert torch.equal(got.pop("visibility"), single["visibility"])
        for f, frame in got.items():
            assert f % 2 == rank
            seen.add(f)
            assert frame["num_rendered"] == single[f]["num_rendered"]
            for k in ("pbr_env", "render_env", "render", "feature", "num_contrib"):
                assert torch.equal(frame[k], single[f][k]), "frame %d of rank %d: %s differs from the single-process frame" % (f, rank, k)
    assert seen == set(range(6))


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank")
def test_bench_gpus_2_rccl():
    r, doc = _bench(["--gpus", "2"] + SMALL, {"R3DG_DIST_BACKEND": "nccl"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert doc["n_gpus"] == 2 and doc["value"] > 0


def _overflow_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    params, cams, bg, gts, K, FusedStage2Step = _make(dev)
    step = FusedStage2Step(params, K, lr=1e-3, bounded=True)
    step(cams[rank], bg, gts[rank])                      # two-phase iteration: learns the count
    step.flush()
    n = step.rendered_counts(1)[0]
    if rank == 1:
        step._capacity = n - 7                           # rank 1's next view will not fit; rank 0's will
    before = {k: getattr(step, k).detach().clone() for k in ("xyz", "shs", "incidents", "env")}
    steps_before, cap_before = step.opt.step_count, step._capacity
    step(cams[rank], bg, gts[rank])
    step.flush()
    torch.cuda.synchronize()
    unchanged = all(torch.equal(getattr(step, k), v) for k, v in before.items())
    dropped = step.poll_overflow()
    grew = step._capacity > cap_before and step._capacity >= 2 * n
    step(cams[rank], bg, gts[rank])                      # trains again
    step.flush()
    torch.cuda.synchronize()
    pars = {k: getattr(step, k).detach().cpu().clone() for k in ("xyz", "shs", "incidents", "env")}
    torch.save(dict(unchanged=unchanged, dropped=dropped, grew=grew, steps=(steps_before, step.opt.step_count),
                    moved=not torch.equal(step.xyz, before["xyz"]), pars=pars, later=step.poll_overflow()),
               os.path.join(out_dir, "ovf%d.pt" % rank))
    dist.destroy_process_group()


def test_a_view_that_does_not_fit_on_one_rank_drops_the_step_on_all_ranks(tmp_path):
    """Bounded forward under data parallelism: the overflow flag rides in the first all-reduce bucket, so when ONE rank's
    view needs more instance slots than it has, EVERY rank's Adam launches skip the step, every rank takes it back from its
    step count, only the rank concerned grows its capacity, and the replicas stay bit-identical."""
    mp.spawn(_overflow_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = torch.load(os.path.join(tmp_path, "ovf0.pt"))
    r1 = torch.load(os.path.join(tmp_path, "ovf1.pt"))
    for r in (r0, r1):
        assert r["unchanged"], "a dropped step updated parameters"
        assert r["dropped"] == 1 and r["later"] == 0 and r["moved"]
        assert r["steps"][1] == r["steps"][0] + 1        # three iterations launched, one dropped
    assert r1["grew"] and not r0["grew"]
    for k in r0["pars"]:
        assert torch.equal(r0["pars"][k], r1["pars"][k]), "replicas diverged: " + k


def _world8_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.bench_core import SYN4_LRS, GaussianParams, render_stage1
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    from relightable3dgaussian_amd.train_step import STAGE2_WEIGHTS_SYN4
    torch.manual_seed(99)
    P, res, K = 1003, 96, 8                                  # ceil(1003 / 8) = 126: the last rank traces 121 ray bundles
    scene = syn.make_scene(P=P, seed=13, stage2=True, scale_log_mean=-3.0)
    cams = [c.to(dev) for c in syn.orbit_cameras(8, width=res, height=res)]
    bg = torch.ones(3, device=dev)
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=P, seed=13, stage2=False, scale_log_mean=-3.0), dev, False)
        gts = [render_stage1(teacher, c, bg)[2].clone() * 0.9 for c in cams]
    out = {}
    for name, kw in (("nerf", dict(loss_weights={"normal": 0.01})), ("syn4", dict(loss_weights=STAGE2_WEIGHTS_SYN4, lrs=SYN4_LRS))):
        step = FusedStage2Step(GaussianParams(scene, dev, True), K, lr=1e-3, bounded=True, **kw)
        assert step.world == 8 and step.dp and step.frozen_geometry == (name == "syn4")
        if name == "nerf":
            out["visibility"] = step.visibility.cpu()
        step(cams[rank], bg, gts[rank])                      # iteration 1: two-phase forward, learns the count
        step.flush()
        if rank == 5:
            step._capacity = step.rendered_counts(1)[0] // 2  # rank 5's next view (another camera of the orbit) will not fit
        before = {k: getattr(step, k).detach().clone() for k in ("xyz", "shs", "incidents", "env", "base_color")}
        n_before = step.opt.step_count
        step(cams[(rank + 3) % 8], bg, gts[(rank + 3) % 8])  # iteration 2: dropped on EVERY rank
        step.flush()
        torch.cuda.synchronize()
        unchanged = all(torch.equal(getattr(step, k), v) for k, v in before.items())
        dropped = step.poll_overflow()
        step(cams[(rank + 5) % 8], bg, gts[(rank + 5) % 8])  # iteration 3 trains again
        step.flush()
        torch.cuda.synchronize()
        out[name] = dict(unchanged=unchanged, dropped=dropped, steps=(n_before, step.opt.step_count),
                         pars={k: getattr(step, k).detach().cpu().clone() for k in before}, later=step.poll_overflow())
    torch.save(out, os.path.join(out_dir, "w8_%d.pt" % rank))
    dist.destroy_process_group()


def test_eight_ranks_rehearsal_on_one_gpu(tmp_path):
    """The 8-rank launch the driver makes at round end, rehearsed on the one GPU of the test box (gloo on device tensors):
    ragged ceil(P / 8) split of the visibility trace + all-gather, the three-bucket all-reduce schedule (and the two-bucket one
    of the frozen-geometry schedule, whose flag rides in its first bucket), a view that does not fit on rank 5 dropping the
    step on all eight ranks, replicas bit-identical afterwards."""
    mp.spawn(_world8_worker, args=(8, _free_port(), str(tmp_path)), nprocs=8, join=True)
    r = [torch.load(os.path.join(tmp_path, "w8_%d.pt" % i)) for i in range(8)]
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    dev = torch.device("cuda", 0)
    single = FusedStage2Step(GaussianParams(syn.make_scene(P=1003, seed=13, stage2=True, scale_log_mean=-3.0), dev, True), 8)
    for i in range(8):
        assert torch.equal(r[i]["visibility"], single.visibility.cpu()), "gathered visibility differs on rank %d" % i
        for name in ("nerf", "syn4"):
            d = r[i][name]
            assert d["unchanged"], "a dropped step updated parameters (%s, rank %d)" % (name, i)
            assert d["dropped"] == 1 and d["later"] == 0 and d["steps"] == (1, 2), (name, i, d["dropped"], d["steps"])
            for k, v in d["pars"].items():
                assert torch.equal(v, r[0][name]["pars"][k]), "replicas diverged: %s %s rank %d" % (name, k, i)
    start = GaussianParams(syn.make_scene(P=1003, seed=13, stage2=True, scale_log_mean=-3.0), dev, True)
    assert not torch.equal(r[0]["nerf"]["pars"]["xyz"], start.xyz.detach().cpu())          # trained ...
    assert torch.equal(r[0]["syn4"]["pars"]["xyz"], start.xyz.detach().cpu())              # ... frozen
    assert not torch.equal(r[0]["syn4"]["pars"]["base_color"], start.base_color.detach().cpu())


def _single_rank_rccl_worker(rank, world, port, out_dir):
    """One rank, backend nccl (= RCCL): with R3DG_DP_SINGLE_RANK=1 the iteration takes the data-parallel path -- three
    async all-reduce buckets on RCCL's stream, the reduced skip flag, the deferred incident-light update, the all-gather
    of update_visibility, the statistics all-reduce of the densification -- with collectives that are identities."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), R3DG_DP_SINGLE_RANK="1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    params, cams, bg, gts, K, FusedStage2Step = _make(dev)
    step = FusedStage2Step(params, K, lr=1e-3, loss_weights={"normal": 0.01})
    assert step.world == 1 and step.dp
    for i in range(4):                       # iteration 1 learns the count, 2..4 run the bounded forward
        step(cams[i % 2], bg, gts[i % 2])
        assert step._pending_b is not None   # the incident-light bucket stays in flight into the next iteration
    step.flush()
    torch.cuda.synchronize()
    out = {k: getattr(step, k).detach().cpu().clone() for k in ("xyz", "shs", "incidents", "env", "opacity", "base_color")}
    out["visibility"] = step.visibility.cpu()
    out["dropped"] = step.poll_overflow()
    out["steps"] = step.opt.step_count
    # stage 1 with a densification: one bucket, statistics reduced before the decisions
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams
    from relightable3dgaussian_amd.fused_step import FusedStage1Step
    p1 = GaussianParams(syn.make_scene(P=2500, seed=11, stage2=False, scale_log_mean=-3.2), dev, False)
    s1 = FusedStage1Step(p1, lr=1e-3)
    assert s1.dp
    s1.enable_densification()
    for i in range(3):
        s1(cams[i % 2], bg, gts[i % 2])
    info = s1.densify_and_prune(1e-7, 0.005, 2.6, 20, 1e9, percent_dense=0.03,
                                generator=torch.Generator(device=dev).manual_seed(77))
    s1(cams[0], bg, gts[0])
    torch.cuda.synchronize()
    out["s1_xyz"], out["s1_rows"] = s1.xyz.detach().cpu().clone(), info["rows_out"]
    torch.save(out, os.path.join(out_dir, "rccl1.pt"))
    dist.destroy_process_group()


def test_single_rank_rccl_path_equals_the_plain_iteration(tmp_path):
    """The RCCL calls of the data-parallel path on the one GPU of the test box (RCCL refuses two ranks on one device, so the
    two-rank tests above use gloo): a one-rank nccl group, R3DG_DP_SINGLE_RANK=1.  Every collective is an identity, so four
    iterations must leave the parameters the plain single-GPU iteration leaves (up to the order of the float atomics)."""
    mp.spawn(_single_rank_rccl_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    got = torch.load(os.path.join(tmp_path, "rccl1.pt"))
    assert got["dropped"] == 0 and got["steps"] == 4
    dev = torch.device("cuda", 0)
    params, cams, bg, gts, K, FusedStage2Step = _make(dev)
    plain = FusedStage2Step(params, K, lr=1e-3, loss_weights={"normal": 0.01})
    assert not plain.dp
    start = {k: getattr(plain, k).detach().cpu().clone() for k in ("xyz", "shs", "incidents", "env", "opacity", "base_color")}
    for i in range(4):
        plain(cams[i % 2], bg, gts[i % 2])
    torch.cuda.synchronize()
    assert torch.equal(got["visibility"], plain.visibility.cpu())
    # (the per-Gaussian gradient sums are float atomics: two runs agree up to their order, and Adam turns a sign flip of a
    # rounding-level gradient into a step of 2 lr -- so: all but a sliver of the elements equal to 1e-5, none off by more
    # than the 4 steps could move it; a bucket that was not reduced / waited for / applied fails both by a wide margin)
    for k in ("xyz", "shs", "incidents", "env", "opacity", "base_color"):
        a, b = got[k], getattr(plain, k).detach().cpu()
        diff = (a - b).abs()
        assert torch.isfinite(a).all() and float(diff.max()) <= 4 * 2e-3, (k, float(diff.max()))
        assert float((diff > 1e-5).float().mean()) < 0.05, "RCCL path differs from the plain iteration: " + k
        assert not torch.equal(a, start[k]), k + " did not train"
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams
    from relightable3dgaussian_amd.fused_step import FusedStage1Step
    p1 = GaussianParams(syn.make_scene(P=2500, seed=11, stage2=False, scale_log_mean=-3.2), dev, False)
    s1 = FusedStage1Step(p1, lr=1e-3)
    s1.enable_densification()
    for i in range(3):
        s1(cams[i % 2], bg, gts[i % 2])
    info = s1.densify_and_prune(1e-7, 0.005, 2.6, 20, 1e9, percent_dense=0.03,
                                generator=torch.Generator(device=dev).manual_seed(77))
    s1(cams[0], bg, gts[0])
    torch.cuda.synchronize()
    assert info["rows_out"] > 2500 and abs(info["rows_out"] - got["s1_rows"]) <= 0.01 * info["rows_out"]
    assert got["s1_xyz"].shape[0] == got["s1_rows"] and torch.isfinite(got["s1_xyz"]).all()


def test_bench_single_rank_over_rccl():
    """`bench.py` with a one-rank RCCL group around the data-parallel iteration (barriers, max-over-ranks reduction of the
    elapsed time, destroy_process_group included)."""
    r, doc = _bench(["--gpus", "1"] + SMALL, {"R3DG_DIST_BACKEND": "nccl", "R3DG_DP_SINGLE_RANK": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert doc["n_gpus"] == 1 and doc["value"] > 0
    # RCCL prints its version banner to stdout through C stdio; captured (a pipe, as under the driver) it used to be flushed at
    # exit, BEHIND the JSON line: the line must be the last thing on stdout
    assert r.stdout.strip().splitlines()[-1].startswith("{"), r.stdout[-400:]
    assert set(doc["comm_buckets"]) == {"A", "B", "C"} and doc["comm_buckets"]["A"]["collective_ms"] > 0
