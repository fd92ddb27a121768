"""Host-side mirror of the reference's `bvh_tracing._C` pybind module (bvh/src/bindings.cpp:8-12, bvh/include/bvh.h:5-18)
over the C ABI:
    create_bvh(means3D, scales, rotations, nodes, aabbs) -> (nodes, aabbs, mortons)      [nodes/aabbs mutated in place]
    trace_bvh_opacity(nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities, normals) -> (num_contributes, rendered_opacity)
    trace_bvh(nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities) -> (num_contributes, point_list, position_list, ray_id_list)
`trace_bvh` (per-ray hit lists, bvh/src/trace.cu:8-192) has no Python caller in the reference (bvh/__init__.py only uses the
two above); it is provided for completeness of the binding surface."""
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


def trace_records(nodes, aabbs, means3D, covs3D, opacities, normals):
    """The tree and the per-Gaussian arrays packed into 64-byte traversal records (r3dg_bvh_pack_traversal), for
    trace_bvh_opacity(..., records=...): pack once, trace many ray chunks.  Valid while the six tensors are unchanged."""
    L = _lib.lib()
    P = means3D.shape[0]
    if nodes.shape[0] != 2 * P - 1:
        raise RuntimeError("trace_records: nodes must be the [2P-1,5] table of a tree over these Gaussians")
    dev = means3D.device
    rec = torch.empty(int(L.r3dg_bvh_trace_records_bytes(P)), dtype=torch.uint8, device=dev)
    t = [x.contiguous() for x in (nodes, aabbs, means3D, covs3D, opacities, normals)]
    with torch.cuda.device(dev):
        _lib.check(L.r3dg_bvh_pack_traversal(_lib.current_stream(), P, *[x.data_ptr() for x in t], rec.data_ptr()),
                   "trace_records")
    return rec


def trace_bvh_opacity(nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities, normals, records=None):
    """`records` (not in the reference): trace_records(...) of the SAME six tensors -- the per-call repacking of the tree
    is skipped (one trace at a time per records buffer)."""
    L = _lib.lib()
    shape = rays_o.shape[:-1]
    num_rays = rays_o.numel() // rays_o.size(-1)
    dev = rays_o.device
    num_contributes = torch.zeros(shape, dtype=torch.int32, device=dev)
    rendered_opacity = torch.ones(shape, dtype=torch.float32, device=dev)
    if num_rays > 0:
        overflow = torch.zeros(1, dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            if records is not None:
                ro, rd = rays_o.contiguous(), rays_d.contiguous()
                st = L.r3dg_bvh_trace_opacity_packed(_lib.current_stream(), num_rays, means3D.shape[0], records.data_ptr(),
                                                     ro.data_ptr(), rd.data_ptr(), num_contributes.data_ptr(),
                                                     rendered_opacity.data_ptr(), overflow.data_ptr())
                if st == 0 and _lib.get_option("TRACE_COUNT_VISITS"):
                    # measurement runs (bench.py: node visits per second): the counting instantiation of the trace ran; add
                    # its node / leaf step sums to the module's accumulators (synchronises)
                    import ctypes
                    v = (ctypes.c_uint64 * 2)()
                    _lib.check(L.r3dg_bvh_trace_visits(_lib.current_stream(), means3D.shape[0], records.data_ptr(), v),
                               "bvh_trace_visits")
                    VISITS[0] += int(v[0])
                    VISITS[1] += int(v[1])
                    VISITS[2] += int(num_rays)
            else:
                t = [x.contiguous() for x in (nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities, normals)]
                P = means3D.shape[0] if nodes.shape[0] == 2 * means3D.shape[0] - 1 else 0
                st = L.r3dg_bvh_trace_opacity(_lib.current_stream(), num_rays, P, *[x.data_ptr() for x in t],
                                              num_contributes.data_ptr(), rendered_opacity.data_ptr(),
                                              overflow.data_ptr())
        _lib.check(st, "trace_bvh_opacity")
        trace_bvh_opacity.last_overflow = overflow
    return num_contributes, rendered_opacity


VISITS = [0, 0, 0]         # node steps, leaf steps, rays of the traces run with R3DG_OPT_TRACE_COUNT_VISITS = 1 (reset by the reader)


def trace_bvh(nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities):
    """bvh/src/bvh.cu:28-86 + trace.cu:8-192: rays_o / rays_d [N,3] -> (num_contributes int32[N,1], point_list int32[n,1],
    position_list float32[n,3], ray_id_list int32[n,1]); the entries of each ray sorted by t, rejected hits last with
    id -1 (t = 1e6).  With no entries the reference returns zeros of shapes [0,1] int32, [0,3] float32 and -- sic --
    [0,3] float32 for the ray ids (trace.cu:219-224); reproduced.  covs3D / opacities are accepted and unused, as in the
    reference (its covariance-weighted t is commented out, trace.cu:121)."""
    L = _lib.lib()
    N = rays_o.size(0)
    dev = rays_o.device
    t = [x.contiguous() for x in (nodes, aabbs, rays_o, rays_d, means3D)]
    counts = torch.zeros((N, 1), dtype=torch.int32, device=dev)
    overflow = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L.r3dg_bvh_trace_count(_lib.current_stream(), N, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                          t[3].data_ptr(), counts.data_ptr(), overflow.data_ptr()), "trace_bvh (count)")
        offsets = torch.cumsum(counts.view(-1), 0, dtype=torch.int64)
        n = int(offsets[-1].item()) if N > 0 else 0                      # the reference reads it back too (trace.cu:69)
        trace_bvh.last_overflow = overflow
        if n == 0:
            return (counts, torch.zeros((0, 1), dtype=torch.int32, device=dev),
                    torch.zeros((0, 3), dtype=torch.float32, device=dev), torch.zeros((0, 3), dtype=torch.float32, device=dev))
        keys = torch.empty(n, dtype=torch.int64, device=dev)
        points = torch.empty(n, dtype=torch.int32, device=dev)
        positions = torch.empty((n, 3), dtype=torch.float32, device=dev)
        ray_ids = torch.empty(n, dtype=torch.int32, device=dev)
        _lib.check(L.r3dg_bvh_trace_fill(_lib.current_stream(), N, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                         t[3].data_ptr(), t[4].data_ptr(), counts.data_ptr(), offsets.data_ptr(),
                                         keys.data_ptr(), points.data_ptr(), positions.data_ptr(), ray_ids.data_ptr()),
                   "trace_bvh (fill)")
        # stable sort by (ray, bits of t): t >= 0.01 or 1e6, so the float bits order like the values
        perm_in = torch.arange(n, dtype=torch.int32, device=dev)
        keys_out, perm = torch.empty_like(keys), torch.empty_like(perm_in)
        temp = torch.empty(int(L.r3dg_sort_temp_bytes(n)), dtype=torch.uint8, device=dev)
        end_bit = 32 + max(1, int(N - 1).bit_length())
        _lib.check(L.r3dg_sort_pairs(_lib.current_stream(), n, keys.data_ptr(), perm_in.data_ptr(), keys_out.data_ptr(),
                                     perm.data_ptr(), min(64, end_bit), temp.data_ptr()), "trace_bvh (sort)")
        idx = perm.long()
        # thrust::stable_sort_by_key permutes point_list and position_list with the keys; ray_id_list is not part of the
        # zip and stays in emission order -- which is already grouped by ray (trace.cu:171-175)
        return counts, points[idx].unsqueeze(-1), positions[idx], ray_ids.unsqueeze(-1)
