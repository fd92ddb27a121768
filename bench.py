#!/usr/bin/env python
"""bench.py -- throughput of the relightable-3DGS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one training iteration of the hot path on one synthetic 800x800 view over a ~300k-Gaussian scene
(BASELINE.json metric): [stage 2] per-Gaussian shading integral (K=64) forward -> rasterize forward (S=16 feature
channels) -> pixel loss -> rasterize backward -> shading backward -> fused Adam step on every Gaussian parameter.
Views are sharded over ranks (one camera per rank per step, weak scaling); per-Gaussian gradients are summed with
one RCCL all-reduce per step.  value = iterations/s summed over all ranks.  Inputs are resident in HBM before the
timed region.  rank 0 prints ONE JSON line.
"""
import argparse
import os
import sys

# the GPU boxes give the container a CPU quota well below the host's core count: OpenMP pools sized for the host (256 spinning
# threads after every parallel CPU op of the scene setup) run into it and the kernel then stalls EVERY thread of the process,
# the one enqueueing kernels included, for the rest of the 100 ms period
os.environ.setdefault("OMP_NUM_THREADS", "8")
os.environ.setdefault("MKL_NUM_THREADS", "8")
# HIP spreads a process's streams over this many hardware queues (default 4), round robin; two streams on one queue run their
# kernels in turn.  The iteration uses three streams and RCCL brings its own: with 4 queues the data-parallel path's early-Adam
# stream shared the main stream's queue (one rank: 578 it/s; with 8 queues 621).  Must be set before the runtime starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# BASELINE.json configs as named workloads (the default, no --config, is the headline: 300k Gaussians, 800x800, run_nerf.sh stage 2)
CONFIG_PRESETS = {
    "dtu4": ("configs[3]: DTU stage-2 (run_dtu.sh: 1600x1200, sample_num 32, smoothness terms, geometry frozen), one view per "
             "rank -- meant for --gpus 4", dict(width=1600, height=1200, objective="syn4", sample_num=32)),
    "teaser8": ("configs[4]: composition scale, 2M Gaussians, 1800x700 (configs/teaser), relight at sample_num 384, frames "
                "sharded over the ranks -- meant for --gpus 8", dict(points=2_000_000, width=1800, height=700, relight_samples=384,
                                                                      relight_frames=12, steps=12, warmup=4)),
    "syn4": ("configs[2]: Synthetic4Relight stage-2 (run_syn4.sh objective + schedule), sample_num 384 as BASELINE.json states",
             dict(objective="syn4", sample_num=384, steps=20, warmup=4)),
    "stage1": ("configs[1]: stage-1 3DGS train iteration, 800x800", dict(stage=1)),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--points", type=int, default=300_000)
    ap.add_argument("--res", type=int, default=800, help="square image size (the BASELINE headline: 800)")
    ap.add_argument("--width", type=int, default=0, help="image width (default: --res); e.g. 1600 for the DTU configuration")
    ap.add_argument("--height", type=int, default=0, help="image height (default: --res); e.g. 1200")
    ap.add_argument("--sample-num", type=int, default=64)
    ap.add_argument("--stage", type=int, default=2, choices=[1, 2])
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true",
                    help="stage 2 through PyTorch autograd glue + torch.optim.Adam instead of the fused glue kernels")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short side measurements of the other BASELINE configurations (stage 1, Syn4Relight / DTU "
                         "objective, 2M-Gaussian composition)")
    ap.add_argument("--objective", default="nerf", choices=["nerf", "syn4"],
                    help="stage-2 objective and schedule: nerf = script/run_nerf.sh (default, the headline); syn4 = "
                         "script/run_syn4.sh / run_dtu.sh (edge-aware smoothness terms, geometry frozen)")
    ap.add_argument("--relight-frames", type=int, default=20)
    ap.add_argument("--relight-samples", type=int, default=384)
    ap.add_argument("--repeats", type=int, default=5,
                    help="N=1 only: extra timed blocks of --steps iterations after the headline block (min/median/max)")
    ap.add_argument("--config", default=None, choices=sorted(CONFIG_PRESETS),
                    help="one of the other BASELINE.json configurations as the timed workload (sets the size / objective flags; "
                         "flags given after it still override): " + "; ".join("%s = %s" % (k, v[0]) for k, v in sorted(CONFIG_PRESETS.items())))
    ap.add_argument("--plumbing-only", action="store_true",
                    help="launch/rendezvous/reduction path only, no kernels (CPU test of the --gpus N launcher)")
    args = ap.parse_args()
    if args.config:
        # a preset fills in every flag the command line left at its default
        given = {a.lstrip("-").split("=")[0].replace("-", "_") for a in sys.argv[1:] if a.startswith("--")}
        for k, v in CONFIG_PRESETS[args.config][1].items():
            if k not in given:
                setattr(args, k, v)
    return args


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start N ranks ourselves, one process per GPU, exactly as the
    driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` would, and let
    rank 0's JSON line through.  Fails loudly when the node has fewer than N GPUs (R3DG_DIST_BACKEND=gloo, the test
    backend, lets ranks share devices)."""
    import subprocess
    backend = os.environ.get("R3DG_DIST_BACKEND", "nccl")
    if backend == "nccl" and not args.plumbing_only and torch.cuda.device_count() < args.gpus:
        sys.exit("bench.py: --gpus %d requested but only %d GPU(s) are visible" % (args.gpus, torch.cuda.device_count()))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
               OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "4"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env, stdin=subprocess.DEVNULL).returncode)


def main():
    args = parse()
    world = os.environ.get("WORLD_SIZE")
    if world is None and args.gpus > 1:
        spawn_ranks(args)
    if world is not None and int(world) != args.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, world))
    from relightable3dgaussian_amd import bench_core
    bench_core.run(args)


if __name__ == "__main__":
    main()
