#!/usr/bin/env python
"""Evidence run for SURVEY.md 8(f) n4 (this container only: needs /root/reference; no GPU here, so the tests' CPU-oracle launcher tests/run_reference_cpu.py):
the reference's unmodified train.py, stage 1 then stage 2 from its checkpoint, on a small synthetic Blender-format scene, then
its unmodified relighting.py on a composition of the trained result.

    python tools/reference_train_py_cpu_demo.py > profiles/r02_reference_train_py_unchanged_cpu.txt
"""
import hashlib
import os
import re
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"


def main():
    from tests import test_reference_scripts_cpu as t
    tmp = tempfile.mkdtemp(prefix="n4_")
    data, s1, s2 = (os.path.join(tmp, d) for d in ("data", "stage1", "stage2"))
    os.makedirs(data)
    n = t._write_dataset(data, n_views=8, res=48, P=1200)
    print("reference: %s  train.py sha256 %s (run as is through tests/run_reference_cpu.py)" % (
        REF, hashlib.sha256(open(os.path.join(REF, "train.py"), "rb").read()).hexdigest()[:16]))
    print("dataset: %d train views 48x48 rendered from a 1200-Gaussian teacher by the CPU oracle, points3d.ply with 600 points" % n)
    runs = [
        ("stage 1 (script/run_nerf.sh:7-14 flags, short schedule)",
         ["train.py", "-s", data, "-m", s1, "--data_device", "cpu", "--lambda_normal_render_depth", "0.01",
          "--lambda_normal_smooth", "0.01", "--lambda_mask_entropy", "0.1", "--lambda_depth_var", "1e-2", "--iterations", "150",
          "--densify_from_iter", "20", "--densification_interval", "30", "--opacity_reset_interval", "100", "--test_interval",
          "50", "--checkpoint_interval", "150", "--save_interval", "150", "--save_training_vis",
          "--save_training_vis_iteration", "75"]),
        ("stage 2 (script/run_nerf.sh:20-39 flags, from the stage-1 checkpoint, sample_num 24)",
         ["train.py", "-s", data, "-m", s2, "-c", os.path.join(s1, "chkpnt150.pth"), "--data_device", "cpu", "-t", "neilf",
          "--sample_num", "24", "--position_lr_init", "0.000016", "--position_lr_final", "0.00000016", "--normal_lr", "0.001",
          "--sh_lr", "0.00025", "--opacity_lr", "0.005", "--scaling_lr", "0.0005", "--rotation_lr", "0.0001", "--iterations",
          "250", "--lambda_base_color_smooth", "0", "--lambda_roughness_smooth", "0", "--lambda_light_smooth", "0",
          "--lambda_light", "0.01", "--lambda_env_smooth", "0.01", "--test_interval", "50", "--checkpoint_interval", "250",
          "--save_interval", "250", "--save_training_vis", "--save_training_vis_iteration", "125", "--densify_until_iter",
          "10"])]
    for title, args in runs:
        t0 = time.time()
        r = t._run(args, timeout=3000)
        print("\n== %s ==\n$ python tests/run_reference_cpu.py --reference %s -- %s" % (
            title, REF, " ".join(a.replace(tmp, "$TMP") for a in args)))
        print("exit code %d, %.1f s" % (r.returncode, time.time() - t0))
        for line in r.stdout.splitlines():
            if re.search(r"stand-ins|Evaluating|Saving|Training complete|Create Gaussians|Number of points|Found", line):
                print("  " + line.strip().replace(tmp, "$TMP"))
        last = [m for m in re.findall(r"num=(\d+)[^\r\n]*?psnr=([0-9.]+)(?:, psnr_pbr=([0-9.]+))?", r.stderr)]
        if last:
            print("  progress bar, first -> last: num=%s psnr=%s%s  ->  num=%s psnr(ema)=%s%s" % (
                last[0][0], last[0][1], (" psnr_pbr=" + last[0][2]) if last[0][2] else "",
                last[-1][0], last[-1][1], (" psnr_pbr(ema)=" + last[-1][2]) if last[-1][2] else ""))
        if r.returncode != 0:
            print(r.stderr[-3000:])
            sys.exit(1)
        out = s1 if "stage 1" in title else s2
        print("  files: " + ", ".join(sorted(f for f in os.listdir(out))))
    # relighting.py: two copies of the trained object composed under different transforms, a turning light, three frames
    import json
    import numpy as np
    from relightable3dgaussian_amd import synthetic as syn
    cfg, cap = os.path.join(tmp, "relight_cfg"), os.path.join(tmp, "capture")
    os.makedirs(cfg)
    ply = os.path.join(s2, "point_cloud", "iteration_250", "point_cloud.ply")
    eye = [1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0, 0, 0, 0, 0, 1.0]
    moved = [0.6, 0, 0, 1.1, 0, 0.6, 0, 0.2, 0, 0, 0.6, 0, 0, 0, 0, 1.0]
    json.dump({"a": {"path": ply, "transform": eye}, "b": {"path": ply, "transform": moved}},
              open(os.path.join(cfg, "transform.json"), "w"))
    traj, lights = {}, {}
    for i, cam in enumerate(syn.orbit_cameras(3, width=64, height=48)):
        traj[str(i)] = cam.world_view_transform.t().reshape(-1).tolist()
        a = 0.4 * i
        lights[str(i)] = [float(np.cos(a)), float(-np.sin(a)), 0.0, float(np.sin(a)), float(np.cos(a)), 0.0, 0.0, 0.0, 1.0]
    json.dump({"camera": {"width": 64, "height": 48, "fov": 40}, "trajectory": traj}, open(os.path.join(cfg, "trajectory.json"), "w"))
    json.dump({"transform": lights}, open(os.path.join(cfg, "light_transform.json"), "w"))
    args = ["relighting.py", "-co", cfg, "-e", os.path.join(REF, "env_map", "envmap3.png"), "--output", cap, "--sample_num", "24",
            "--capture_list", "pbr_env,render_env,base_color,normal,visibility", "-bg", "0"]
    t0 = time.time()
    r = t._run(args, timeout=3000)
    print("\n== relighting.py (relighting.py:102-170): composition of two objects from the point_cloud.ply train.py wrote, "
          "envmap3.png, a light that turns with the frames ==\n$ python tests/run_reference_cpu.py --reference %s -- %s"
          % (REF, " ".join(a.replace(tmp, "$TMP") for a in args)))
    print("exit code %d, %.1f s" % (r.returncode, time.time() - t0))
    for line in r.stdout.splitlines():
        if re.search(r"Totally|stand-ins", line):
            print("  " + line.strip())
    if r.returncode != 0:
        print(r.stderr[-3000:])
        sys.exit(1)
    print("  files: " + ", ".join("%s/%s" % (d, f) for d in sorted(os.listdir(cap)) for f in sorted(os.listdir(os.path.join(cap, d)))))
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
