#!/usr/bin/env python
""""effective batch becomes G views/step (reference is 1) -- PSNR parity at equal VIEW count must be checked,
lr unchanged".  Trains the same stage-2 scene (objective of script/run_nerf.sh:20-39) three ways and prints the
view-averaged PSNR of the SH render and of the PBR render over all views:

    1 rank,  2N iterations          the reference's schedule: one view per optimizer step
    G ranks, N iterations each      data parallel: the SAME 2N views (rank r takes views G*i + r), G views per optimizer step
    1 rank,  N iterations           equal number of optimizer steps, half the views

    python tools/dp_psnr_equal_views.py [--ranks 2,4] [--iters 240]  > profiles/rNN_dp_psnr_equal_views.txt

and, for every G, the data-parallel run again with the learning rate of every group scaled by sqrt(G) and by G (the two usual
large-batch rules; Adam normalises the gradient's scale, so summing G views changes the direction's noise, not its length --
what fewer steps lose has to come from the step size).

The test box has one GPU: the ranks share it over the gloo test backend (RCCL refuses two ranks on one device); the
arithmetic of the reduction is the same sum."""
import argparse
import math
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
P, RES, K, VIEWS, LR = 50_000, 320, 64, 8, 2e-3


def _psnr(a, b):
    return -10.0 * math.log10(float(((a - b) ** 2).mean()))


def _setup(dev):
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams, render_stage1
    torch.manual_seed(4321)
    scene = syn.make_scene(P=P, seed=31, stage2=True, scale_log_mean=-3.9)
    cams = [c.to(dev) for c in syn.orbit_cameras(VIEWS, width=RES, height=RES)]
    bg = torch.ones(3, device=dev)
    with torch.no_grad():
        teacher = GaussianParams(syn.make_scene(P=P, seed=31, stage2=False, scale_log_mean=-3.9), dev, False)
        teacher.features_dc.add_(0.3 * torch.randn_like(teacher.features_dc))
        gts = [render_stage1(teacher, c, bg)[2].clone() for c in cams]
    return GaussianParams(scene, dev, True), cams, bg, gts


def _evaluate(step, cams, bg, gts):
    from relightable3dgaussian_amd.train_step import rgb_to_srgb
    render, pbr = [], []
    with torch.no_grad():
        for c, gt in zip(cams, gts):
            o = step.forward_backward(c, bg, gt)              # (gradients are written, parameters are not touched)
            for h in (getattr(step, "_handles", None) or ()):  # data parallel: EVERY rank evaluates, so the collectives
                if h is not None:                              # this launches are matched; complete them
                    h.wait()
            n_contrib, image, opacity, feature = o[1], o[2], o[3], o[5]
            feat = feature / opacity.clamp_min(1e-5) * (n_contrib > 0)
            render.append(_psnr(image, gt))
            pbr.append(_psnr(rgb_to_srgb(feat[2:5] * opacity + (1 - opacity) * bg[:, None, None]), gt))
    return sum(render) / len(render), sum(pbr) / len(pbr)


def _train(rank, world, iters, port, out, lr_scale=1.0):
    from relightable3dgaussian_amd.fused_step import FusedStage2Step
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    params, cams, bg, gts = _setup(dev)
    step = FusedStage2Step(params, K, lr=LR * lr_scale)
    before = _evaluate(step, cams, bg, gts)
    for it in range(iters):
        v = (world * it + rank) % VIEWS
        step(cams[v], bg, gts[v])
    step.flush()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    after = _evaluate(step, cams, bg, gts)
    if rank == 0:
        torch.save(dict(before=before, after=after, steps=step.opt.step_count, dropped=step.dropped_steps), out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", default="2", help="comma-separated rank counts, e.g. 2,4")
    ap.add_argument("--iters", type=int, default=240, help="iterations of the one-view-per-step schedule")
    a = ap.parse_args()
    N2 = a.iters
    print("stage-2 objective of script/run_nerf.sh:20-39; %d Gaussians, %dx%d, sample_num %d, %d views, lr %g on every group"
          % (P, RES, RES, K, VIEWS, LR))
    print("%-100s %10s %10s %8s" % ("run", "PSNR", "PSNR pbr", "steps"))

    def run(title, world, iters, lr_scale=1.0, show_before=False):
        out = "/tmp/dp_psnr_%d_%d.pt" % (world, iters)
        mp.spawn(_train, args=(world, iters, _free_port(), out, lr_scale), nprocs=world, join=True)
        r = torch.load(out)
        if show_before:
            print("%-100s %10.3f %10.3f %8s" % ("before training", r["before"][0], r["before"][1], "-"))
        print("%-100s %10.3f %10.3f %8d" % (title, r["after"][0], r["after"][1], r["steps"]))
        assert r["dropped"] == 0
        return r

    base = run("1 rank, %d iterations (one view per step: the reference's schedule)" % N2, 1, N2, show_before=True)
    lines = []
    for G in [int(g) for g in a.ranks.split(",")]:
        n = N2 // G
        half = run("1 rank, %d iterations (as many optimizer steps as the %d-rank run, 1/%d of the views)" % (n, G, G), 1, n)
        for name, scale in (("lr x 1", 1.0), ("lr x sqrt(G) = %.3f" % math.sqrt(G), math.sqrt(G)), ("lr x G = %d" % G, float(G))):
            r = run("%d ranks, %d iterations each (the same %d views, %d per step), %s" % (G, n, N2, G, name), G, n, scale)
            lines.append("G=%d %-22s equal views: data parallel - single = %+.3f dB (render) %+.3f dB (pbr);   equal steps: %+.3f dB "
                         "(render) %+.3f dB (pbr)" % (G, name, r["after"][0] - base["after"][0], r["after"][1] - base["after"][1],
                                                      r["after"][0] - half["after"][0], r["after"][1] - half["after"][1]))
    print()
    for ln in lines:
        print(ln)


if __name__ == "__main__":
    main()
