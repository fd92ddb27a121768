#!/usr/bin/env python
"""Where does shade_backward_kernel's time go?  Builds experiment variants of libr3dg_hip.so in which ONE part of the kernel is
removed (tools/build_variant.py: the product source is patched in a temporary copy; results of the variants are WRONG by
construction -- only their kernel times mean something) and, on the GPU box, times each with tools/kbench_shade.py.

    python tools/ablate_shade_backward.py build            # here (no GPU): relightable3dgaussian_amd/lib/variants/abl_*/
    python tools/ablate_shade_backward.py run out.json     # on the GPU box: {variant: backward ms (K=64, cached taps, P=300k)}
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ATOMIC = """                                atomicAdd(reinterpret_cast<unsigned long long*>(&s_denv[3 * tex + c]), bits);"""
VARIANTS = {
    # the texture-gradient scatter: without the LDS atomics (conversion kept alive) / without the whole scatter
    "abl_no_lds_atomic": [(ATOMIC, '                                asm volatile("" :: "v"(bits));')],
    "abl_no_env_grad": [("                if (s.vis != 0.f) {\n                    const float ev[3]", "                if (false) {\n                    const float ev[3]")],
    # pass 0 (SH basis + 48 coefficient FMAs per sample, only for the local light's value and sign)
    "abl_no_pass0": [("""                float Y[16];
                sh_basis16(d[0], d[1], d[2], M, Y);
                sh_local_sum(s_u, Y, sum);
            }
            s_park[(3 * t) * PARK] = sum[0];""", """                sum[0] = d[0]; sum[1] = d[1]; sum[2] = d[2];
            }
            s_park[(3 * t) * PARK] = sum[0];""")],
    # pass 2 (basis again + the 48 SH-gradient FMAs per sample)
    "abl_no_pass2": [("                for (int f = 0; f < 48; f++) acc[f] += dl[f % 3] * Y[f / 3];",
                      "                for (int f = 0; f < 3; f++) acc[f] += dl[f % 3] * Y[f / 3];")],
    # the GGX backward chain (roughness + view-direction gradients)
    "abl_no_ggx_backward": [("                accb[3] += dkk * 2.f / 8.f + da * 2.f * G.r;               // roughness", "                accb[3] += dkk;"),
                            ("                for (int c = 0; c < 3; c++) accb[4 + c] += (dV[c] - G.V[c] * vd) / G.vlen;   // view direction",
                             "                for (int c = 0; c < 3; c++) accb[4 + c] += s.Hh[c];")],
    # the final per-Gaussian reduction (4 x 16-channel transposing row reductions per 4 samples per lane at K = 64)
    "abl_no_final_reduce": [("                r[pass] = row_transpose_reduce16(v);", "                r[pass] = v[0] + v[5] + v[11];"),
                            ("                r[3] = row_transpose_reduce16(v);", "                r[3] = v[0] + v[3] + v[6];")],
}


def build():
    from tools.build_variant import build_variant
    for name, reps in VARIANTS.items():
        print(name, build_variant(name, "shading.hip", reps))


def run(out):
    from tools.build_variant import VARIANTS as VDIR
    res = {}
    for name in ["product"] + sorted(VARIANTS):
        env = dict(os.environ, ONLY64="1")
        if name != "product":
            env["R3DG_LIB_PATH"] = os.path.join(VDIR, name, "libr3dg_hip.so")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kbench_shade.py")], capture_output=True, text=True,
                           env=env, timeout=300)
        m = re.search(r"backward \(cached taps\) ([0-9.]+) ms", r.stdout)
        f = re.search(r"forward \(train outputs, cached taps, uniform area\) ([0-9.]+) ms", r.stdout)
        res[name] = dict(backward_ms=float(m.group(1)) if m else None, forward_ms=float(f.group(1)) if f else None,
                         err=None if m else (r.stderr or r.stdout)[-300:])
        print(name, res[name], flush=True)
    json.dump(dict(note="shade_backward_kernel with ONE part removed per variant (wrong results by construction; times only); "
                        "tools/kbench_shade.py ONLY64=1: P=300000, K=64, 16x32 texture, cached taps", variants=res),
              open(out, "w"), indent=1)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    else:
        run(sys.argv[2])
