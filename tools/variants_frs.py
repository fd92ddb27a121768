#!/usr/bin/env python
"""A/B candidates of the fixed-ray-set shading kernels (tools/build_variant.py; results are valid in every variant):
    python tools/variants_frs.py build
    python tools/variants_frs.py run out.json          # GPU box: tools/kbench_shade.py per variant
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
H = "shading_frs.hpp"


def _head(path):
    return subprocess.run(["git", "show", "HEAD:relightable3dgaussian_amd/csrc/" + path], capture_output=True, text=True,
                          cwd=ROOT, check=True).stdout


def _cur(path):
    return open(os.path.join(ROOT, "relightable3dgaussian_amd", "csrc", path)).read()


SB_FWD = "            __builtin_amdgcn_sched_barrier(0);       // (the prefetch is issued HERE, not sunk into the block to save registers)\n"
SB_BWD = "            __builtin_amdgcn_sched_barrier(0);\n            f32x4 l[3]"
FWD_SIG = "__global__ void __launch_bounds__(64 * FRS_WAVES, 3)\nshade_forward_frs_kernel("
VARIANTS = {
    # the committed kernels (git HEAD) beside the working tree's: same box, same run
    "frs_head": lambda: [(_cur("shading.hip"), _head("shading.hip"), None, "shading.hip"), (_cur(H), _head(H), None, H)],
}


def build():
    from tools.build_variant import build_variant
    for name, reps in VARIANTS.items():
        print(name, build_variant(name, "shading.hip", reps()))


def run(out):
    from tools.build_variant import VARIANTS as VDIR
    res = {}
    only = [v for v in os.environ.get("VARIANTS_ONLY", "").split(",") if v]
    for name in ["product"] + sorted(only or VARIANTS):
        env = dict(os.environ, ONLY64="1")
        if name != "product":
            env["R3DG_LIB_PATH"] = os.path.join(VDIR, name, "libr3dg_hip.so")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench_shade.py")], capture_output=True, text=True,
                           env=env, timeout=300)
        vals = dict(re.findall(r"(frs forward|frs backward|backward \(cached taps\)|forward \(train outputs, cached taps, uniform area\))"
                               r"(?: \([^)]*\))? +([0-9.]+) ms", r.stdout))
        res[name] = {k: float(v) for k, v in vals.items()} or {"err": (r.stderr or r.stdout)[-300:]}
        print(name, res[name], flush=True)
    json.dump(dict(note="tools/kbench_shade.py ONLY64=1 (P=300000, K=64) per variant of the fixed-ray-set kernels; ms per call "
                        "(the frs figures include the coefficient rotation and the general kernel on the listed Gaussians)",
                   variants=res), open(out, "w"), indent=1)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run(sys.argv[2])
