#!/usr/bin/env python
"""Evidence run for SURVEY.md 8(f) n4 ON HARDWARE: the reference's unmodified train.py -- stage 1, then stage 2 from its
checkpoint -- and its unmodified relighting.py, executed through tools/run_reference.py against this repo's HIP-backed
extension packages (no --cpu-oracle: every `_C.*` call is a HIP kernel) on a synthetic Blender-format scene.

    python tools/reference_train_py_gpu_run.py --reference reference_scratch > gpurun_out/reference_train_py_gpu.txt

`--reference` is a checkout of NJU-3DV/Relightable3DGaussian; on the GPU box that is the untracked copy of its Python which
tools/stage_reference_scratch.sh stages (the box has no /root/reference).  The dataset is rendered here by the HIP rasterizer
from a teacher scene (synthetic.py), so PSNR against it is meaningful; the run prints the reference's own log lines, the
progress bar's first and last PSNR, wall time and iterations per second of the reference's Python loop."""
import argparse
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def write_dataset(root, n_views, res, P, n_points):
    import reference_shims as shims
    from relightable3dgaussian_amd import synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams, render_stage1
    dev = torch.device("cuda", 0)
    sc = syn.make_scene(P=P, seed=5, stage2=False, scale_log_mean=-3.4)
    teacher = GaussianParams(sc, dev, False)
    cams = syn.orbit_cameras(n_views, width=res, height=res)
    bg = torch.ones(3, device=dev)
    with torch.no_grad():
        images = [render_stage1(teacher, c.to(dev), bg)[2].clamp(0, 1).cpu() for c in cams]
    syn.write_blender_dataset(root, cams, images, split="train")
    syn.write_blender_dataset(root, cams[::8], images[::8], split="test")
    g = np.random.default_rng(3)
    keep = g.permutation(P)[:n_points]
    xyz = sc["xyz"].numpy()[keep] + 0.01 * g.standard_normal((n_points, 3)).astype(np.float32)
    data = np.empty(n_points, dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"),
                                     ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    nrm = sc["normal"].numpy()[keep]
    for i, n in enumerate(("x", "y", "z")):
        data[n], data["n" + n] = xyz[:, i], nrm[:, i]
    rgb = (255 * g.random((n_points, 3))).astype(np.uint8)
    data["red"], data["green"], data["blue"] = rgb[:, 0], rgb[:, 1], rgb[:, 2]
    shims.PlyData([shims.PlyElement.describe(data, "vertex")]).write(os.path.join(root, "points3d.ply"))
    return len(cams)


def headline(ns, ref):
    """the reference's UNMODIFIED train.py, stage 2 (script/run_nerf.sh:20-39 flags), at the headline size --
    300 000 Gaussians, 800x800, sample_num 64 -- for `--stage2-iterations` iterations, once as is (its pure-PyTorch
    rendering_equation between the drop-in ops) and once with INTEGRATION.md's one-line rendering_equation patch applied from
    outside (tools/run_reference.py --patch-rendering-equation).  The stage-1 checkpoint it starts from is written in the
    reference's format by checkpoint.capture from a 300 000-Gaussian synthetic scene (bench.py's), the views are rendered from
    the same scene by the HIP rasterizer."""
    from relightable3dgaussian_amd import checkpoint, fused_step, synthetic as syn
    from relightable3dgaussian_amd.bench_core import GaussianParams, render_stage1
    import reference_shims as shims
    dev = torch.device("cuda", 0)
    P, res, I1 = ns.points, ns.res, 30000
    tmp = tempfile.mkdtemp(prefix="n4head_")
    data, s1 = os.path.join(tmp, "data"), os.path.join(tmp, "stage1")
    os.makedirs(data)
    os.makedirs(s1)
    sc = syn.make_scene(P=P, seed=0, stage2=False)
    teacher = GaussianParams(sc, dev, False)
    cams = syn.orbit_cameras(ns.views, width=res, height=res)
    bg = torch.ones(3, device=dev)
    with torch.no_grad():
        images = [render_stage1(teacher, c.to(dev), bg)[2].clamp(0, 1).cpu() for c in cams]
    syn.write_blender_dataset(data, cams, images, split="train")
    syn.write_blender_dataset(data, cams[::8], images[::8], split="test")
    n_points = 4000
    g = np.random.default_rng(3)
    keep = g.permutation(P)[:n_points]
    pts = np.empty(n_points, dtype=[("x", "f4"), ("y", "f4"), ("z", "f4"), ("nx", "f4"), ("ny", "f4"), ("nz", "f4"),
                                    ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    for i, n in enumerate(("x", "y", "z")):
        pts[n], pts["n" + n] = sc["xyz"].numpy()[keep, i], sc["normal"].numpy()[keep, i]
    pts["red"] = pts["green"] = pts["blue"] = 128
    shims.PlyData([shims.PlyElement.describe(pts, "vertex")]).write(os.path.join(data, "points3d.ply"))
    step = fused_step.FusedStage1Step(GaussianParams(sc, dev, False), lr=1e-4)
    torch.save(checkpoint.capture(step, I1), os.path.join(s1, "chkpnt%d.pth" % I1))
    del step, teacher
    torch.cuda.empty_cache()
    from relightable3dgaussian_amd import _lib
    print("device: %s   library: %s" % (torch.cuda.get_device_name(0), _lib.LIB_PATH))
    print("reference checkout: %s  train.py sha256 %s" % (ns.reference, hashlib.sha256(open(os.path.join(ref, "train.py"), "rb").read()).hexdigest()[:16]))
    print("headline size: %d Gaussians (checkpoint.capture of bench.py's synthetic scene as chkpnt%d.pth), %d views %dx%d, sample_num %d, "
          "%d stage-2 iterations; target of BASELINE.json: >= 40 train iters/s" % (P, I1, ns.views, res, res, ns.sample_num,
                                                                                    ns.stage2_iterations))
    i2 = ns.stage2_iterations
    for title, extra in (("train.py UNPATCHED (pure-PyTorch rendering_equation, neilf.py:339-407)", []),
                         ("train.py with INTEGRATION.md's one-line rendering_equation patch (applied from outside)",
                          ["--patch-rendering-equation"])):
        s2 = os.path.join(tmp, "stage2" + ("_patched" if extra else ""))
        args = ["train.py", "-s", data, "-m", s2, "-c", os.path.join(s1, "chkpnt%d.pth" % I1), "-t", "neilf",
                "--sample_num", str(ns.sample_num), "--position_lr_init", "0.000016", "--position_lr_final", "0.00000016",
                "--normal_lr", "0.001", "--sh_lr", "0.00025", "--opacity_lr", "0.005", "--scaling_lr", "0.0005", "--rotation_lr",
                "0.0001", "--iterations", str(I1 + i2), "--lambda_base_color_smooth", "0", "--lambda_roughness_smooth", "0",
                "--lambda_light_smooth", "0", "--lambda_light", "0.01", "--lambda_env_smooth", "0.01", "--test_interval",
                str(10 * i2), "--checkpoint_interval", str(10 * i2), "--save_interval", str(10 * i2), "--densify_until_iter", "10"]
        cmd = [sys.executable, os.path.join(ROOT, "tools", "run_reference.py"), "--reference", ref] + extra + ["--"] + args
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=ns.timeout, env=dict(os.environ, PYTHONPATH=ROOT), cwd=ROOT,
                           stdin=subprocess.DEVNULL)
        dt = time.time() - t0
        print("\n== %s ==\n$ python tools/run_reference.py --reference %s %s-- %s" % (
            title, ns.reference, " ".join(extra) + (" " if extra else ""), " ".join(a.replace(tmp, "$TMP") for a in args)))
        print("exit code %d, %.1f s wall (process start + data loading + visibility trace + %d iterations + saving)" % (
            r.returncode, dt, i2))
        for line in r.stdout.splitlines():
            if re.search(r"run_reference\]|Training complete|Number of points|Found|\[ITER", line):
                print("  " + line.strip().replace(tmp, "$TMP"))
        bar = re.findall(r"(\d+)/(\d+) \[(\d+):(\d+)<[^,\]]*, *([0-9.]+)(it/s|s/it)[^\r\n]*?num=(\d+)[^\r\n]*?psnr=([0-9.]+)(?:, psnr_pbr=([0-9.]+))?",
                         r.stderr)
        if bar:
            # tqdm's rate is smoothed over the last updates; the mean rate = iterations done / elapsed at the last update
            f, l = bar[0], bar[-1]
            done, elapsed = int(l[0]) - I1, 60 * int(l[2]) + int(l[3])
            rate = float(l[4]) if l[5] == "it/s" else 1.0 / max(float(l[4]), 1e-9)
            print("  progress bar: num=%s, psnr %s -> %s (ema), psnr_pbr %s -> %s (ema); tqdm rate at the end %.1f it/s%s -- target 40: %s"
                  % (l[6], f[7], l[7], f[8], l[8], rate, (", %d iterations in %d s" % (done, elapsed)) if elapsed else "",
                     "met" if rate >= 40 else "NOT met"))
        if r.returncode != 0:
            print(r.stdout[-3000:])
            print(r.stderr[-6000:])
            sys.exit(1)
    if ns.profile:
        host_profile(ns, ref, tmp, args)
    shutil.rmtree(tmp, ignore_errors=True)


def host_profile(ns, ref, tmp, args):
    """where the reference's loop spends its HOST time at the headline size -- cProfile over the patched run,
    self time (tottime) partitioned by the file a function lives in: the reference's own Python, this repo's wrappers (the three
    drop-in packages + relightable3dgaussian_amd), torch / everything else.  cProfile inflates small Python calls, so the shares
    matter more than the absolute numbers; the unprofiled rate is printed above."""
    import pstats
    prof = os.path.join(tmp, "patched.prof")
    args = [a.replace("stage2_patched", "stage2_profiled") for a in args]
    cmd = [sys.executable, "-m", "cProfile", "-o", prof, os.path.join(ROOT, "tools", "run_reference.py"), "--reference", ref,
           "--patch-rendering-equation", "--"] + args
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=ns.timeout, env=dict(os.environ, PYTHONPATH=ROOT), cwd=ROOT,
                       stdin=subprocess.DEVNULL)
    print("\n== host profile of the patched run (python -m cProfile; %d iterations; exit code %d, %.1f s wall) ==" % (
        ns.stage2_iterations, r.returncode, time.time() - t0))
    if r.returncode != 0:
        print(r.stderr[-3000:])
        return
    st = pstats.Stats(prof)
    groups = {"reference (reference_scratch/)": 0.0, "this repo's wrappers (relightable3dgaussian_amd/, r3dg_rasterization/, bvh_tracing/, simple_knn/)": 0.0,
              "torch (python side)": 0.0, "built-ins / C calls (kernel launches, ctypes calls, .item() waits)": 0.0, "other python": 0.0}
    ours = ("relightable3dgaussian_amd", "r3dg_rasterization", "bvh_tracing", "simple_knn")
    rows = []
    for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
        if "reference_scratch" in fn:
            k = "reference (reference_scratch/)"
        elif any(("/" + o + "/") in fn for o in ours):
            k = [g for g in groups if g.startswith("this repo")][0]
        elif "/torch/" in fn:
            k = "torch (python side)"
        elif fn.startswith("~") or fn.startswith("<"):
            k = [g for g in groups if g.startswith("built-ins")][0]
        else:
            k = "other python"
        groups[k] += tt
        rows.append((tt, ct, nc, fn, line, name))
    total = sum(groups.values())
    n = ns.stage2_iterations
    print("self time by where the function lives (whole process: start-up, data loading, the visibility trace and %d iterations):" % n)
    for k, v in groups.items():
        print("  %6.2f s  %5.1f %%   %s" % (v, 100 * v / total, k))
    def cum(pred):
        return sum(ct for tt, ct, nc, fn, line, name in rows if pred(fn, name))
    print("cumulative time of the calls a training iteration makes (ms per iteration, %d iterations):" % n)
    for label, pred in (
            ("train.py: training loop body (whole run)", lambda fn, nm: fn.endswith("train.py") and nm == "training"),
            ("reference render_view (gaussian_renderer/neilf.py)", lambda fn, nm: "reference_scratch" in fn and fn.endswith("neilf.py") and nm == "render_view"),
            ("reference calculate_loss (neilf.py)", lambda fn, nm: "reference_scratch" in fn and fn.endswith("neilf.py") and nm == "calculate_loss"),
            ("this repo: rasterize_gaussians (forward wrapper)", lambda fn, nm: "rasterizer_ops" in fn and nm == "rasterize_gaussians"),
            ("this repo: rasterize_gaussians_backward (wrapper)", lambda fn, nm: "rasterizer_ops" in fn and nm == "rasterize_gaussians_backward"),
            ("this repo: rendering_equation (shading_ops)", lambda fn, nm: "shading_ops" in fn and nm == "rendering_equation"),
            ("torch: Tensor.backward", lambda fn, nm: fn.endswith("torch/_tensor.py") and nm == "backward"),
            ("torch: optimizer step (Adam, 13 groups)", lambda fn, nm: fn.endswith("optim/adam.py") and nm == "step"),
            ("Tensor.item (host waits for the GPU here)", lambda fn, nm: "item" in nm and fn.startswith("~")),
    ):
        print("  %8.2f   %s" % (1e3 * cum(pred) / n, label))
    print("largest self times:")
    for tt, ct, nc, fn, line, name in sorted(rows, reverse=True)[:22]:
        short = fn.replace(ROOT, ".").replace(ref, "reference_scratch")
        print("  %6.2f s self %6.2f s cum %8d calls  %s:%d %s" % (tt, ct, nc, short[-70:], line, name))


def run(reference, args, timeout):
    cmd = [sys.executable, os.path.join(ROOT, "tools", "run_reference.py"), "--reference", reference, "--"] + args
    env = dict(os.environ, PYTHONPATH=ROOT)
    return subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT, stdin=subprocess.DEVNULL)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", default=os.environ.get("R3DG_REFERENCE", os.path.join(ROOT, "reference_scratch")))
    ap.add_argument("--res", type=int, default=256)
    ap.add_argument("--views", type=int, default=48)
    ap.add_argument("--teacher-points", type=int, default=30000)
    ap.add_argument("--stage1-iterations", type=int, default=600)
    ap.add_argument("--stage2-iterations", type=int, default=400)
    ap.add_argument("--sample-num", type=int, default=24)
    ap.add_argument("--timeout", type=int, default=600)
    ap.add_argument("--headline", action="store_true",
                    help="stage 2 only, at --points / --res / --sample-num (VERDICT r3 item 4): unpatched and patched train.py")
    ap.add_argument("--points", type=int, default=300000)
    ap.add_argument("--profile", action="store_true", help="with --headline: cProfile of the patched run, split by code owner")
    ns = ap.parse_args()
    ref = os.path.abspath(ns.reference)
    if ns.headline:
        return headline(ns, ref)
    tmp = tempfile.mkdtemp(prefix="n4gpu_")
    data, s1, s2 = (os.path.join(tmp, d) for d in ("data", "stage1", "stage2"))
    os.makedirs(data)
    n = write_dataset(data, ns.views, ns.res, ns.teacher_points, 4000)
    from relightable3dgaussian_amd import _lib
    print("device: %s   library: %s" % (torch.cuda.get_device_name(0), _lib.LIB_PATH))
    print("reference checkout: %s  train.py sha256 %s (run as is through tools/run_reference.py, HIP extension packages)" % (
        ns.reference, hashlib.sha256(open(os.path.join(ref, "train.py"), "rb").read()).hexdigest()[:16]))
    print("dataset: %d train views %dx%d rendered by the HIP rasterizer from a %d-Gaussian teacher, points3d.ply with 4000 points"
          % (n, ns.res, ns.res, ns.teacher_points))
    i1, i2 = ns.stage1_iterations, ns.stage2_iterations
    runs = [
        ("stage 1 (script/run_nerf.sh:7-14 flags, short schedule)", s1, i1,
         ["train.py", "-s", data, "-m", s1, "--lambda_normal_render_depth", "0.01",
          "--lambda_normal_smooth", "0.01", "--lambda_mask_entropy", "0.1", "--lambda_depth_var", "1e-2", "--iterations", str(i1),
          "--densify_from_iter", "100", "--densification_interval", "100", "--opacity_reset_interval", "300", "--test_interval",
          str(i1 // 2), "--checkpoint_interval", str(i1), "--save_interval", str(i1), "--save_training_vis",
          "--save_training_vis_iteration", str(i1 // 2)]),
        ("stage 2 (script/run_nerf.sh:20-39 flags, from the stage-1 checkpoint, sample_num %d)" % ns.sample_num, s2, i2,
         ["train.py", "-s", data, "-m", s2, "-c", os.path.join(s1, "chkpnt%d.pth" % i1), "-t", "neilf",
          "--sample_num", str(ns.sample_num), "--position_lr_init", "0.000016", "--position_lr_final", "0.00000016", "--normal_lr",
          "0.001", "--sh_lr", "0.00025", "--opacity_lr", "0.005", "--scaling_lr", "0.0005", "--rotation_lr", "0.0001",
          "--iterations", str(i1 + i2), "--lambda_base_color_smooth", "0", "--lambda_roughness_smooth", "0", "--lambda_light_smooth",
          "0", "--lambda_light", "0.01", "--lambda_env_smooth", "0.01", "--test_interval", str(i2 // 2), "--checkpoint_interval",
          str(i1 + i2), "--save_interval", str(i1 + i2), "--save_training_vis", "--save_training_vis_iteration", str(i2 // 2),
          "--densify_until_iter", "10"])]
    for title, out, iters, args in runs:
        t0 = time.time()
        r = run(ref, args, ns.timeout)
        dt = time.time() - t0
        print("\n== %s ==\n$ python tools/run_reference.py --reference %s -- %s" % (
            title, ns.reference, " ".join(a.replace(tmp, "$TMP") for a in args)))
        print("exit code %d, %.1f s wall (process start + data loading + %d iterations + evaluation / saving)" % (r.returncode, dt, iters))
        for line in r.stdout.splitlines():
            if re.search(r"stand-ins|Evaluating|Saving|Training complete|Create Gaussians|Number of points|Found|\[ITER", line):
                print("  " + line.strip().replace(tmp, "$TMP"))
        bar = re.findall(r"(\d+)/(\d+) \[(\d+):(\d+)<[^,\]]*, *([0-9.]+)(it/s|s/it)[^\r\n]*?num=(\d+)[^\r\n]*?psnr=([0-9.]+)(?:, psnr_pbr=([0-9.]+))?",
                         r.stderr)
        if bar:
            f, l = bar[0], bar[-1]
            rate = float(l[4]) if l[5] == "it/s" else 1.0 / max(float(l[4]), 1e-9)
            print("  progress bar, first -> last: num=%s psnr=%s%s  ->  num=%s psnr(ema)=%s%s   (%s/%s, tqdm rate %.1f it/s)" % (
                f[6], f[7], (" psnr_pbr=" + f[8]) if f[8] else "", l[6], l[7], (" psnr_pbr(ema)=" + l[8]) if l[8] else "",
                l[0], l[1], rate))
        if r.returncode != 0:
            print(r.stdout[-3000:])
            print(r.stderr[-6000:])
            sys.exit(1)
        print("  files: " + ", ".join(sorted(os.listdir(out))))
    # relighting.py: two copies of the trained object under different transforms, a turning light, three frames
    from relightable3dgaussian_amd import synthetic as syn
    cfg, cap = os.path.join(tmp, "relight_cfg"), os.path.join(tmp, "capture")
    os.makedirs(cfg)
    ply = os.path.join(s2, "point_cloud", "iteration_%d" % (i1 + i2), "point_cloud.ply")
    eye = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0]
    moved = [0.6, 0, 0, 1.1, 0, 0.6, 0, 0.2, 0, 0, 0.6, 0, 0, 0, 0, 1.0]
    json.dump({"a": {"path": ply, "transform": eye}, "b": {"path": ply, "transform": moved}},
              open(os.path.join(cfg, "transform.json"), "w"))
    traj, lights = {}, {}
    for i, cam in enumerate(syn.orbit_cameras(3, width=400, height=300)):
        traj[str(i)] = cam.world_view_transform.t().reshape(-1).tolist()
        a = 0.4 * i
        lights[str(i)] = [float(np.cos(a)), float(-np.sin(a)), 0.0, float(np.sin(a)), float(np.cos(a)), 0.0, 0.0, 0.0, 1.0]
    json.dump({"camera": {"width": 400, "height": 300, "fov": 40}, "trajectory": traj}, open(os.path.join(cfg, "trajectory.json"), "w"))
    json.dump({"transform": lights}, open(os.path.join(cfg, "light_transform.json"), "w"))
    args = ["relighting.py", "-co", cfg, "-e", os.path.join(ref, "env_map", "envmap3.png"), "--output", cap, "--sample_num", "64",
            "--capture_list", "pbr_env,render_env,base_color,normal,visibility", "-bg", "0"]
    t0 = time.time()
    r = run(ref, args, ns.timeout)
    print("\n== relighting.py (relighting.py:102-170): composition of two objects from the point_cloud.ply train.py wrote, "
          "envmap3.png, a light that turns with the frames ==\n$ python tools/run_reference.py --reference %s -- %s"
          % (ns.reference, " ".join(a.replace(tmp, "$TMP").replace(ref, ns.reference) for a in args)))
    print("exit code %d, %.1f s wall" % (r.returncode, time.time() - t0))
    for line in r.stdout.splitlines():
        if re.search(r"Totally|stand-ins", line):
            print("  " + line.strip())
    if r.returncode != 0:
        print(r.stdout[-3000:])
        print(r.stderr[-6000:])
        sys.exit(1)
    print("  files: " + ", ".join("%s/%s" % (d, f) for d in sorted(os.listdir(cap)) for f in sorted(os.listdir(os.path.join(cap, d)))))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
