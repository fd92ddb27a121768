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
FWD_SIG = "__global__ void __launch_bounds__(64 * FRS_WAVES)\nshade_forward_frs_kernel("
BWD_SIG = "__global__ void __launch_bounds__(64 * FRS_WAVES)\nshade_backward_frs_kernel("
VARIANTS = {
    # forward at 4 / 5 waves per SIMD instead of 3 (141 VGPRs today)
    "frs_fwd_occ4": [(FWD_SIG, FWD_SIG.replace("(64 * FRS_WAVES)", "(64 * FRS_WAVES, 4)"), None, H)],
    "frs_fwd_occ5": [(FWD_SIG, FWD_SIG.replace("(64 * FRS_WAVES)", "(64 * FRS_WAVES, 5)"), None, H)],
    # backward at 3 waves per SIMD instead of 2 (256 registers today)
    "frs_bwd_occ3": [(BWD_SIG, BWD_SIG.replace("(64 * FRS_WAVES)", "(64 * FRS_WAVES, 3)"), None, H)],
    # the four samples of a block two at a time (less unrolling, fewer live registers)
    "frs_unroll2": [("#pragma unroll\n            for (int v = 0; v < 4; v++) {\n                const int k = 16 * b + 4 * q + v;",
                     "#pragma unroll 2\n            for (int v = 0; v < 4; v++) {\n                const int k = 16 * b + 4 * q + v;", 2, H)],
    # texture-gradient scatter without double precision: per Gaussian a power-of-two scale brings the clamped contribution into
    # int32 (v_rndne + v_cvt_i32), sign-extended and shifted back into the common 64-bit fixed-point unit
    "frs_i32_fixed": [
        ("""        const float nom1 = G.NoV * (1.f - G.kk) + G.kk;
        float bc[4][3];
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int c = 0; c < 3; c++) bc[s][c] = cprime[(size_t)gc * 48 + (4 * s + q) * 3 + c];
        f32x4 dcq[3]""", """        const float nom1 = G.NoV * (1.f - G.kk) + G.kk;
        // exponent gap between the largest upstream gradient of the launch and this Gaussian's: its contributions are below
        // gmax 2^(14 - E), i.e. 2^(49 - E) fixed-point units; with sh = max(0, 19 - E) and a clamp at 2^29 the value round(x S 2^-sh)
        // fits int32 and one contribution stays below 2^48 units, as in the double-precision form
        int sh;
        float fx_scale_p;
        {
            const float gpmax = fmaxf(fmaxf(fmaxf(fabsf(gp[0]), fabsf(gp[1])), fabsf(gp[2])),
                                      fmaxf(fmaxf(fabsf(gd[0]), fabsf(gd[1])), fabsf(gd[2]))) * (float)K;
            const int E = (int)((__float_as_uint(gmax) >> 23) & 0xffu) - (int)((__float_as_uint(fmaxf(gpmax, 1e-37f)) >> 23) & 0xffu);
            sh = E >= 19 ? 0 : (E < 0 ? 19 : 19 - E);
            fx_scale_p = fx_scale * __uint_as_float((unsigned)(127 - sh) << 23);
        }
        float bc[4][3];
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int c = 0; c < 3; c++) bc[s][c] = cprime[(size_t)gc * 48 + (4 * s + q) * 3 + c];
        f32x4 dcq[3]""", None, H),
        ("""                    if (fixed) {
                        const double scale_d = (double)fx_scale;
#pragma unroll
                        for (int tq = 0; tq < 4; tq++) {
#pragma unroll
                            for (int c = 0; c < 3; c++) {
                                const float cl = __builtin_amdgcn_fmed3f(ev[c] * w4[tq], -fx_clamp, fx_clamp);
                                const double dsum = __builtin_fma((double)cl, scale_d, 6755399441055744.0);
                                const unsigned long long bits =
                                    (unsigned long long)__double_as_longlong(dsum) - 0x4338000000000000ull;
                                atomicAdd(reinterpret_cast<unsigned long long*>(&s_denv[3 * tex[tq] + c]), bits);
                            }
                        }
                    } else {""", """                    if (fixed) {
                        const float evs[3] = {ev[0] * fx_scale_p, ev[1] * fx_scale_p, ev[2] * fx_scale_p};
#pragma unroll
                        for (int tq = 0; tq < 4; tq++) {
#pragma unroll
                            for (int c = 0; c < 3; c++) {
                                const float cl = __builtin_amdgcn_fmed3f(evs[c] * w4[tq], -536870912.0f, 536870912.0f);
                                const long long v64 = (long long)(int)rintf(cl) << sh;
                                atomicAdd(reinterpret_cast<unsigned long long*>(&s_denv[3 * tex[tq] + c]), (unsigned long long)v64);
                            }
                        }
                    } else {""", None, H)],
}


def build():
    from tools.build_variant import build_variant
    for name, reps in VARIANTS.items():
        print(name, build_variant(name, "shading.hip", reps))


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
