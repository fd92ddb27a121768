"""Build the REAL reference kernels for gfx950 as a GPU-side oracle (TEST INFRASTRUCTURE ONLY).

    python -m oracle.build_ref        # needs /root/reference (this container); output: oracle/_ref/libr3dg_reference.so

The reference's CUDA sources are compiled where they lie under /root/reference -- never copied into the repo -- with
hipcc and the header shims in oracle/ref_shim/ (cuda_runtime.h -> HIP, cub -> hipcub, cooperative_groups -> HIP's, thrust
is rocThrust).  The only source-level difference hipcc cannot digest is nvcc's tolerance for spaces inside the kernel
launch chevrons (`<< <grid, block >> >`); those are closed up on the fly in a temporary directory that is deleted after
the compile.  Sources built: r3dg-rasterization/cuda_rasterizer/{forward,backward,rasterizer_impl}.cu and
bvh/src/{construct,trace}.cu and submodules/simple-knn/simple_knn.cu (the torch-facing glue -- rasterize_points.cu, bvh.cu -- is replaced by
oracle/ref_shim/ref_wrapper.cpp).
r3dg-rasterization/render_equation.cu (not part of the reference's own setup.py, SURVEY.md F1, but a complete translation
unit) is compiled the same way, against the torch headers of this image because it includes <torch/extension.h>, into a
second library oracle/_ref/libr3dg_reference_shading.so (linked to libtorch; oracle/ref_shim/ref_wrapper_shading.cpp
calls its raw-pointer launchers).  It pins the render_equation_* contract model (tests/test_render_equation_gpu.py).
oracle/_ref/ is git-ignored but travels to the GPU box, where tests/test_reference_gpu.py uses it if present.
"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT_DIR = os.path.join(HERE, "_ref")
OUT = os.path.join(OUT_DIR, "libr3dg_reference.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SHIM = os.path.join(HERE, "ref_shim")
RAST = os.path.join(REF, "r3dg-rasterization")
BVH = os.path.join(REF, "bvh")
KNN = os.path.join(REF, "submodules", "simple-knn")
INCLUDES = ["-I", SHIM, "-I", os.path.join(RAST, "third_party", "glm"), "-I", os.path.join(RAST, "cuda_rasterizer"),
            "-I", os.path.join(BVH, "include"), "-I", KNN]
SOURCES = [os.path.join(RAST, "cuda_rasterizer", f) for f in ("forward.cu", "backward.cu", "rasterizer_impl.cu")] + \
          [os.path.join(BVH, "src", f) for f in ("construct.cu", "trace.cu")] + [os.path.join(KNN, "simple_knn.cu")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w"]
SHADING_SRC = os.path.join(RAST, "render_equation.cu")
SHADING_WRAPPER = os.path.join(SHIM, "ref_wrapper_shading.cpp")
SHADING_OUT = os.path.join(OUT_DIR, "libr3dg_reference_shading.so")


def available():
    return all(os.path.exists(s) for s in SOURCES)


def _compile(args):
    src, tmp = args
    name = os.path.basename(os.path.dirname(os.path.dirname(src))) + "_" + os.path.basename(src)[:-3]
    name = name.replace("-", "_")
    hip_src = os.path.join(tmp, name + ".hip")
    text = open(src).read()
    text = re.sub(r"<<\s+<", "<<<", text)
    text = re.sub(r">>\s+>", ">>>", text)
    with open(hip_src, "w") as f:
        f.write(text)
    obj = os.path.join(tmp, name + ".o")
    r = subprocess.run([HIPCC] + FLAGS + INCLUDES + ["-c", hip_src, "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("reference build failed for %s:\n%s" % (src, r.stderr[-4000:]))
    return obj


def build(force=False):
    if not available():
        return None
    stamp = max(os.path.getmtime(p) for p in SOURCES + [os.path.join(SHIM, "ref_wrapper.cpp")])
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= stamp:
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        with ThreadPoolExecutor(max_workers=6) as ex:
            objs = list(ex.map(_compile, [(s, tmp) for s in SOURCES]))
        wobj = os.path.join(tmp, "ref_wrapper.o")
        r = subprocess.run([HIPCC, "-x", "hip"] + FLAGS + INCLUDES + ["-c", os.path.join(SHIM, "ref_wrapper.cpp"), "-o", wobj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("reference wrapper build failed:\n%s" % r.stderr[-4000:])
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + [wobj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("reference link failed:\n%s" % r.stderr[-4000:])
    return OUT


def build_shading(force=False):
    """render_equation.cu -> oracle/_ref/libr3dg_reference_shading.so (needs the torch headers: ~2 min of hipcc)."""
    if not os.path.exists(SHADING_SRC):
        return None
    stamp = max(os.path.getmtime(p) for p in (SHADING_SRC, SHADING_WRAPPER, __file__))
    if not force and os.path.exists(SHADING_OUT) and os.path.getmtime(SHADING_OUT) >= stamp:
        return SHADING_OUT
    import sysconfig
    import torch
    tdir = os.path.dirname(torch.__file__)
    tinc = os.path.join(tdir, "include")
    tlib = os.path.join(tdir, "lib")
    os.makedirs(OUT_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        hip_src = os.path.join(tmp, "render_equation.hip")
        text = open(SHADING_SRC).read()
        text = re.sub(r"<<\s+<", "<<<", text)
        text = re.sub(r">>\s+>", ">>>", text)
        with open(hip_src, "w") as f:
            f.write(text)
        obj, wobj = os.path.join(tmp, "render_equation.o"), os.path.join(tmp, "wrapper.o")
        defs = ["-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DHIPBLAS_V2",
                "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
        inc = ["-I", SHIM, "-I", os.path.join(RAST, "third_party", "glm"), "-I", RAST, "-I", tinc,
               "-I", os.path.join(tinc, "torch", "csrc", "api", "include"), "-I", sysconfig.get_paths()["include"]]
        for cmd in ([HIPCC] + FLAGS + defs + inc + ["-c", hip_src, "-o", obj],
                    [HIPCC, "-x", "hip"] + FLAGS + ["-I", os.path.join(RAST, "third_party", "glm"), "-c", SHADING_WRAPPER,
                                                    "-o", wobj],
                    [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SHADING_OUT, obj, wobj, "-L" + tlib,
                     "-Wl,-rpath," + tlib, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_hip", "-lc10_hip",
                     "-Wl,--no-undefined"]):
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("reference render_equation build failed:\n%s" % r.stderr[-4000:])
    return SHADING_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_shading(force="--force" in sys.argv))
