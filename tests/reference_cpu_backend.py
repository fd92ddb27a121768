"""TEST INFRASTRUCTURE (never imported by the product): lets the reference's unmodified Python run in THIS container, which
has no GPU -- behind `tools/run_reference.py the CPU-oracle launcher tests/run_reference_cpu.py` and tests/test_reference_scripts_cpu.py.

* the three extension modules the reference imports (`r3dg_rasterization._C`, `bvh_tracing._C`, `simple_knn._C`) are backed
  by the CPU oracle (oracle/rasterizer_oracle.c, bvh_oracle.c, knn_oracle.c: restatements pinned to the real reference
  build by tests/test_reference_gpu.py) behind the reference's signatures -- i.e. the boundary of include/r3dg_hip.h;
* the reference's hard-coded device="cuda" (scene/gaussian_model.py, utils/general_utils.py, bvh/__init__.py, ...) lands on
  the CPU: torch factory functions drop the `device` keyword, `Tensor.cuda()` is the identity, `torch.cuda.set_device` /
  `empty_cache` do nothing (utils/general_utils.py:167, gaussian_model.py:914).
The same pieces generate tests/golden/pipeline_reference_stage{1,2}.npz (tests/golden/make_pipeline_golden.py)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
_installed = False


def _knn_dist2(points):
    """simple_knn._C.distCUDA2 (submodules/simple-knn/simple_knn.cu:185-221): mean squared distance to the 3 nearest
    neighbours -- brute force in the oracle (fine for the few thousand points of a test scene)."""
    from oracle import rasterizer as orc
    lib = orc.lib()
    pts = np.ascontiguousarray(points.detach().cpu().numpy(), np.float32)
    out = np.empty(pts.shape[0], np.float32)
    lib.knno_dist2(C.c_int(pts.shape[0]), pts.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return torch.from_numpy(out)


def install():
    global _installed
    if _installed:
        return
    _installed = True
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_golden as mg
    import make_pipeline_golden as mpg
    mpg.install_oracle_extensions()                      # r3dg_rasterization._C, bvh_tracing._C (+ a constant distCUDA2)
    sys.modules["simple_knn._C"].distCUDA2 = _knn_dist2
    for p in mg._cpu_factories():
        p.start()
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.empty_cache = lambda: None
    to = torch.Tensor.to

    def _is_cuda(x):
        return (isinstance(x, str) and x.startswith("cuda")) or (isinstance(x, torch.device) and x.type == "cuda")

    def to_cpu(self, *a, **k):       # `.to("cuda")` (train.py:349), `.to(torch.device("cuda"))` (scene/cameras.py:27-35)
        a = tuple("cpu" if _is_cuda(x) else x for x in a)
        if _is_cuda(k.get("device")):
            k["device"] = "cpu"
        return to(self, *a, **k)
    torch.Tensor.to = to_cpu
