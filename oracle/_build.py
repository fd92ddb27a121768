"""Build recipe for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Compiles oracle/*.c into oracle/liboracle.so with gcc.  Nothing in the product package imports this
module; it is used by tests/, __graft_entry__.build()/smoke() and bench.py's cpu_baseline leg.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liboracle.so")
SOURCES = ["rasterizer_oracle.c", "bvh_oracle.c", "shading_oracle.c", "knn_oracle.c"]
# -ffp-contract=off: fp32 in the reference's operation order with no fused multiply-add, so integer
# decisions (radii, tile rects, keys) are reproducible bit-for-bit by the HIP kernels.
CFLAGS = ["-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-std=c11", "-fopenmp"]


def sources():
    return [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]


def build(force=False):
    srcs = sources()
    if not force and os.path.exists(LIB):
        newest = max(os.path.getmtime(s) for s in srcs)
        if os.path.getmtime(LIB) >= newest:
            return LIB
    cmd = ["gcc"] + CFLAGS + ["-o", LIB] + srcs + ["-lm"]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
