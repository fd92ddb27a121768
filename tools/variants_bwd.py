#!/usr/bin/env python
"""Ablation builds of the tile backward (tools/build_variant.py; the product sources are untouched, the variants' RESULTS ARE WRONG --
they exist to attribute the kernel's time):
    bwd_no_atomics     the per-(wave, Gaussian) atomic instruction removed            -> what the atomics cost
    (round 5, against the round-4 kernel whose lanes went to FIVE arrays: "bwd_one_line", every lane aimed at one 64-byte row per
     Gaussian, took 0.258 ms against 0.417 -- profiles/r05_bwd_ablation.txt; the product now has that layout)
    bwd_no_reduce      the transposing reduction replaced by the lane's own first value (+ the atomic)
    python tools/variants_bwd.py build
    python tools/variants_bwd.py run        # GPU box: tools/kbench_raster.py per variant (ACTIVE=2,3,4, no depth gradient)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = "rasterizer_render_bwd.hip"
ATOMIC = "                if (dst_base != nullptr) atomicAdd(dst_base + (size_t)__float_as_uint(g1.w) * NVP, total);\n"
VARIANTS = {
    "bwd_no_atomics": [(ATOMIC, "                if (dst_base != nullptr && total == 12345.678f) atomicAdd(dst_base, total);\n")],
    "bwd_no_reduce": [("                    total = transpose_reduce12(vr);\n",
                       "                    total = vr[0];\n#pragma unroll\n                    for (int q = 1; q < 12; q++) total += vr[q];\n")],
    # the generic 16-wide reduction in the twelve-channel instance (what the custom one replaced)
    "bwd_reduce16": [("    constexpr bool R12 = NV == 12; ", "    constexpr bool R12 = false; ")],
}


def build():
    from tools.build_variant import build_variant
    for name, reps in VARIANTS.items():
        print(name, build_variant(name, SRC, reps))


def run():
    from tools.build_variant import VARIANTS as VDIR
    for name in [None] + list(VARIANTS):
        env = dict(os.environ, ACTIVE="2,3,4", NODEPTH="1", ITERS="10")
        if name:
            env["R3DG_LIB_PATH"] = os.path.join(VDIR, name, "libr3dg_hip.so")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench_raster.py")], env=env, capture_output=True, text=True)
        print("%-16s %s" % (name or "product", (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1]))


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
