"""Host-side mirror of the reference's `bvh_tracing._C` pybind module (bvh/src/bindings.cpp:8-12, bvh/include/bvh.h:5-18)
over the C ABI:
    create_bvh(means3D, scales, rotations, nodes, aabbs) -> (nodes, aabbs, mortons)      [nodes/aabbs mutated in place]
    trace_bvh_opacity(nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities, normals) -> (num_contributes, rendered_opacity)
`trace_bvh` (per-ray hit lists, bvh/src/trace.cu:8-192) has no Python caller in the reference (bvh/__init__.py only
uses the two above) and is not provided; calling it raises NotImplementedError."""
import torch

from . import _lib


def create_bvh(means3D, scales, rotations, nodes, aabbs):
    L = _lib.lib()
    P = means3D.size(0)
    if nodes.dtype != torch.int32 or aabbs.dtype != torch.float32 or not nodes.is_cuda or not aabbs.is_cuda:
        raise RuntimeError("create_bvh: nodes must be int32 and aabbs float32 CUDA(HIP) tensors")
    if nodes.shape != (2 * P - 1, 5) or aabbs.shape != (2 * P - 1, 6):
        raise RuntimeError("create_bvh: nodes must be [2P-1,5] and aabbs [2P-1,6]")
    if not nodes.is_contiguous() or not aabbs.is_contiguous():
        raise RuntimeError("create_bvh: nodes/aabbs are updated in place and must be contiguous")
    mortons = torch.zeros((P,), dtype=torch.int64, device=means3D.device)
    if P > 0:
        temp = torch.empty(int(L.r3dg_bvh_build_temp_bytes(P)), dtype=torch.uint8, device=means3D.device)
        with torch.cuda.device(means3D.device):
            st = L.r3dg_bvh_build(_lib.current_stream(), P, nodes.data_ptr(), aabbs.data_ptr(), mortons.data_ptr(),
                                  temp.data_ptr())
        _lib.check(st, "create_bvh")
    return nodes, aabbs, mortons


def trace_bvh_opacity(nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities, normals):
    L = _lib.lib()
    shape = rays_o.shape[:-1]
    num_rays = rays_o.numel() // rays_o.size(-1)
    dev = rays_o.device
    num_contributes = torch.zeros(shape, dtype=torch.int32, device=dev)
    rendered_opacity = torch.ones(shape, dtype=torch.float32, device=dev)
    if num_rays > 0:
        overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        t = [x.contiguous() for x in (nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities, normals)]
        with torch.cuda.device(dev):
            P = means3D.shape[0] if nodes.shape[0] == 2 * means3D.shape[0] - 1 else 0
            st = L.r3dg_bvh_trace_opacity(_lib.current_stream(), num_rays, P, *[x.data_ptr() for x in t],
                                          num_contributes.data_ptr(), rendered_opacity.data_ptr(),
                                          overflow.data_ptr())
        _lib.check(st, "trace_bvh_opacity")
        trace_bvh_opacity.last_overflow = overflow
    return num_contributes, rendered_opacity


def trace_bvh(*_args, **_kwargs):
    raise NotImplementedError("trace_bvh (hit lists) has no caller in the reference's Python and is not provided")
