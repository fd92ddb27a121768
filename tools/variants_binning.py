#!/usr/bin/env python
"""Ablation builds of the direct tile binning's count kernel (tools/build_variant.py; the variants' RESULTS ARE WRONG -- they exist to
attribute the kernel's time at 2M Gaussians, where it takes 0.25 ms for 24 MB of input):
    count_no_flush     the per-(workgroup, tile) global atomics removed
    count_no_hist      the LDS histogram pass removed (zero fill + flush stay)
    python tools/variants_binning.py build
    python tools/variants_binning.py run      # GPU box: rocprofv3 kernel stats of a forward at 2M per variant
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = "rasterizer_preprocess.hip"
VARIANTS = {
    "count_no_flush": [("        if (c) atomicAdd(&tile_counts[t], c);\n", "        if (c == 0xffffffffu) atomicAdd(&tile_counts[t], c);\n")],
    "count_no_hist": [("                          [&](uint32_t tile, uint32_t, uint32_t) { atomicAdd(&s_bins[tile], 1u); });\n    __syncthreads();\n    for (int t = threadIdx.x; t < T; t += BIN_THREADS) {\n        const uint32_t c = s_bins[t];\n        if (c) atomicAdd(&tile_counts[t], c);",
                       "                          [&](uint32_t tile, uint32_t, uint32_t) { if (tile == 0xffffffffu) atomicAdd(&s_bins[tile & 1023u], 1u); });\n    __syncthreads();\n    for (int t = threadIdx.x; t < T; t += BIN_THREADS) {\n        const uint32_t c = s_bins[t] + 1u;\n        if (c) atomicAdd(&tile_counts[t], c);")],
}


def build():
    from tools.build_variant import build_variant
    for name, reps in VARIANTS.items():
        print(name, build_variant(name, SRC, reps))


def run():
    from tools.build_variant import VARIANTS as VDIR
    for name in [None] + list(VARIANTS):
        env = dict(os.environ, P="2000000", W="1800", H="700", ITERS="6")
        if name:
            env["R3DG_LIB_PATH"] = os.path.join(VDIR, name, "libr3dg_hip.so")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench_raster.py")], env=env, capture_output=True, text=True)
        print("%-16s %s" % (name or "product", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
