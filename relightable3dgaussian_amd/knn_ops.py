"""Host-side mirror of the reference's `simple_knn._C` module (submodules/simple-knn/ext.cpp:15-17):
`distCUDA2(points[P,3]) -> float32[P]`, the mean squared distance to the 3 nearest neighbours that
GaussianModel.create_from_pcd uses to initialise the scales (scene/gaussian_model.py:427)."""
import torch

from . import _lib


def distCUDA2(points):
    L = _lib.lib()
    if points.dim() != 2 or points.size(1) != 3 or points.dtype != torch.float32 or not points.is_cuda:
        raise RuntimeError("distCUDA2 expects a float32 CUDA(HIP) tensor of shape [P,3]")
    P = points.size(0)
    pts = points.contiguous()
    out = torch.zeros((P,), dtype=torch.float32, device=points.device)
    if P > 0:
        temp = torch.empty(int(L.r3dg_knn_temp_bytes(P)), dtype=torch.uint8, device=points.device)
        with torch.cuda.device(points.device):
            st = L.r3dg_knn_dist2(_lib.current_stream(), P, pts.data_ptr(), out.data_ptr(), temp.data_ptr())
        _lib.check(st, "distCUDA2")
    return out
