#!/usr/bin/env python
"""Ablation builds of the direct tile binning's kernels (tools/build_variant.py; the variants' RESULTS ARE WRONG -- they exist to
attribute the kernels' time at 2M Gaussians, where count + emit take 0.18 + 0.34 ms inside the iteration):
    count_no_flush     count: the per-(workgroup, tile) global atomics removed
  (Ablations of the EMIT kernel were tried in round 6 and removed: entries that are missing or misplaced leave uninitialised
  Gaussian indices in the tile lists, and the tile kernels behind them fault on the gathers -- a wrong-result variant is only safe
  when everything downstream becomes empty, as with count_no_flush.)
    python tools/variants_binning.py build
    python tools/variants_binning.py run      # GPU box: stage times of a forward at 2M per variant (tools/kbench_raster.py)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = "rasterizer_preprocess.hip"
VARIANTS = {
    "count_no_flush": [("        if (c) atomicAdd(&tile_counts[t], c);\n", "        if (c == 0xffffffffu) atomicAdd(&tile_counts[t], c);\n")],
}


def build():
    from tools.build_variant import build_variant
    for name, reps in VARIANTS.items():
        print(name, build_variant(name, SRC, reps))


def run():
    from tools.build_variant import VARIANTS as VDIR
    for name in [None] + list(VARIANTS):
        env = dict(os.environ, P="2000000", RES="1120", ITERS="6")          # (1120^2 ~ the 1800x700 pixels of the composition configuration)
        if name:
            env["R3DG_LIB_PATH"] = os.path.join(VDIR, name, "libr3dg_hip.so")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench_raster.py")], env=env, capture_output=True, text=True)
        print("%-16s %s" % (name or "product", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
