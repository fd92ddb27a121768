"""In-tree build of libr3dg_hip.so for gfx950 (hipcc cross-compiles without a GPU).

    python -m relightable3dgaussian_amd.build [--force]

Each csrc/*.hip is compiled to an object with `hipcc --offload-arch=gfx950 -O3`, then linked into
relightable3dgaussian_amd/lib/libr3dg_hip.so.  The built .so is git-ignored but travels to the GPU box.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libr3dg_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-I", os.path.join(os.path.dirname(HERE), "include")]
# Per-file extra flags.  rasterizer_preprocess.hip decides integer outputs (radii, tile rects, sort keys) from
# fp32 math and must match the CPU oracle bit for bit -> no fused multiply-add contraction there.
EXTRA = {
    "rasterizer_preprocess.hip": ["-ffp-contract=off"],
    "bvh.hip": ["-ffp-contract=off"],
    "simple_knn.hip": ["-ffp-contract=off"],
    "densify.hip": ["-ffp-contract=off"],         # clone / split / prune decisions are fp32 comparisons
    # shading: VALU-bound transcendental-heavy float math; fp32 tolerance is 1e-4, so reciprocal/sqrt approximations
    # (v_rcp_f32, v_sqrt_f32: 1 ulp) replace the IEEE division/sqrt expansions
    "shading.hip": ["-ffast-math", "-fno-slp-vectorize"],
}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(path, flags):
    h = hashlib.sha1()
    h.update(" ".join(flags).encode())
    for p in [path] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hpp")] + \
             [os.path.join(os.path.dirname(HERE), "include", "r3dg_hip.h")]:
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _compile(src, force):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJDIR, src[:-4] + ".o")
    flags = COMMON + EXTRA.get(src, []) + os.environ.get("R3DG_EXTRA_HIPCC_FLAGS", "").split()   # experiments only
    stamp_file = obj + ".stamp"
    stamp = _stamp(path, flags)
    if not force and os.path.exists(obj) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return obj, False
    cmd = [HIPCC] + flags + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force), srcs))
    objs = [o for o, _ in results]
    if force or any(c for _, c in results) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
