"""ctypes face of oracle/_ref/libr3dg_reference.so -- the REAL reference kernels compiled for gfx950 (oracle/build_ref.py).
TEST INFRASTRUCTURE ONLY: a GPU-side oracle ("kind: reference") used by tests/test_reference_gpu.py when the library
is present.  Mirrors the call conventions of the reference's own glue (rasterize_points.cu:36-235, bvh.cu:8-116)."""
import ctypes as C
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libr3dg_reference.so")
ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(LIB_PATH)
        _lib.ref_rasterize_forward.restype = C.c_int
    return _lib


def _p(t):
    return None if (t is None or t.numel() == 0) else C.c_void_p(t.data_ptr())


def rasterize_forward(bg, means3D, features, colors, opacity, scales, rotations, scale_modifier, cov3D, viewmatrix,
                      projmatrix, tan_fovx, tan_fovy, cx, cy, H, W, sh, degree, campos, pseudo_normal=True):
    dev = means3D.device
    P, S = means3D.shape[0], features.shape[1]
    M = sh.shape[1] if (sh is not None and sh.numel()) else 0
    f = dict(dtype=torch.float32, device=dev)
    out = dict(color=torch.zeros(3, H, W, **f), opacity=torch.zeros(1, H, W, **f), depth=torch.zeros(1, H, W, **f),
               feature=torch.zeros(S, H, W, **f), normal=torch.zeros(3, H, W, **f), xyz=torch.zeros(3, H, W, **f),
               weights=torch.zeros(P, 1, **f), radii=torch.zeros(P, dtype=torch.int32, device=dev))
    bufs = [None, None, None]

    def mk(i):
        def cb(_u, n):
            bufs[i] = torch.empty(int(n) + 256, dtype=torch.uint8, device=dev)
            return bufs[i].data_ptr()
        return ALLOC_FN(cb)
    cbs = [mk(i) for i in range(3)]
    torch.cuda.synchronize()
    R = lib().ref_rasterize_forward(cbs[0], cbs[1], cbs[2], None, P, S, int(degree), M, _p(bg), W, H, _p(means3D), _p(sh),
                                    _p(colors), _p(features), _p(opacity), _p(scales), C.c_float(scale_modifier),
                                    _p(rotations), _p(cov3D), _p(viewmatrix), _p(projmatrix), _p(campos),
                                    C.c_float(tan_fovx), C.c_float(tan_fovy), C.c_float(cx), C.c_float(cy),
                                    int(bool(pseudo_normal)), _p(out["color"]), _p(out["opacity"]), _p(out["depth"]),
                                    _p(out["feature"]), _p(out["normal"]), _p(out["xyz"]), _p(out["weights"]),
                                    _p(out["radii"]))
    out["num_rendered"] = R
    out["buffers"] = bufs
    T = ((W + 15) // 16) * ((H + 15) // 16)
    out["n_contrib"] = torch.zeros(H, W, dtype=torch.int32, device=dev)
    out["final_T"] = torch.zeros(H, W, **f)
    out["ranges"] = torch.zeros(T, 2, dtype=torch.int32, device=dev)
    out["point_list"] = torch.zeros(max(R, 1), dtype=torch.int32, device=dev)
    out["keys"] = torch.zeros(max(R, 1), dtype=torch.int64, device=dev)
    lib().ref_decode_state(_p(bufs[2]), _p(bufs[1]) if R > 0 else None, W, H, R, _p(out["n_contrib"]), _p(out["final_T"]),
                           _p(out["ranges"]), _p(out["point_list"]), _p(out["keys"]))
    return out


def rasterize_backward(fwd, bg, means3D, features, colors, scales, rotations, scale_modifier, cov3D, viewmatrix,
                       projmatrix, tan_fovx, tan_fovy, gC, gO, gD, gF, sh, degree, campos, backward_geometry=True):
    dev = means3D.device
    P, S = means3D.shape[0], features.shape[1]
    M = sh.shape[1] if (sh is not None and sh.numel()) else 0
    H, W = gC.shape[1], gC.shape[2]
    f = dict(dtype=torch.float32, device=dev)
    g = dict(mean2D=torch.zeros(P, 3, **f), conic=torch.zeros(P, 2, 2, **f), opacity=torch.zeros(P, 1, **f),
             color=torch.zeros(P, 3, **f), feature=torch.zeros(P, S, **f), mean3D=torch.zeros(P, 3, **f),
             cov3D=torch.zeros(P, 6, **f), sh=torch.zeros(P, M, 3, **f), scale=torch.zeros(P, 3, **f),
             rot=torch.zeros(P, 4, **f))
    b = fwd["buffers"]
    torch.cuda.synchronize()
    lib().ref_rasterize_backward(P, S, int(degree), M, int(fwd["num_rendered"]), _p(bg), W, H, _p(means3D), _p(sh),
                                 _p(features), _p(colors), _p(scales), C.c_float(scale_modifier), _p(rotations),
                                 _p(cov3D), _p(viewmatrix), _p(projmatrix), _p(campos), C.c_float(tan_fovx),
                                 C.c_float(tan_fovy), _p(fwd["radii"]), _p(b[0]), _p(b[1]), _p(b[2]), _p(gC), _p(gO),
                                 _p(gD), _p(gF), _p(g["mean2D"]), _p(g["conic"]), _p(g["opacity"]), _p(g["color"]),
                                 _p(g["feature"]), _p(g["mean3D"]), _p(g["cov3D"]), _p(g["sh"]), _p(g["scale"]),
                                 _p(g["rot"]), int(bool(backward_geometry)))
    return g


def bvh_build(means3D, scales, rotations, nodes, aabbs):
    P = means3D.shape[0]
    morton = torch.zeros(P, dtype=torch.int64, device=means3D.device)
    torch.cuda.synchronize()
    lib().ref_bvh_build(P, _p(means3D), _p(scales), _p(rotations), _p(nodes), _p(aabbs), _p(morton))
    return nodes, aabbs, morton


def bvh_trace_opacity(nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities, normals):
    shape = rays_o.shape[:-1]
    n = rays_o.numel() // 3
    cnt = torch.zeros(shape, dtype=torch.int32, device=rays_o.device)
    opa = torch.ones(shape, dtype=torch.float32, device=rays_o.device)
    t = [x.contiguous() for x in (nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities, normals)]
    torch.cuda.synchronize()
    lib().ref_bvh_trace_opacity(n, *[_p(x) for x in t], _p(cnt), _p(opa))
    return cnt, opa


def bvh_trace(nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities, capacity):
    """trace_bvh_cuda of the real reference build (trace.cu:8-192) -> (num_rendered, num_contributes[N], point_list[n],
    position_list[n,3], ray_id_list[n]); `capacity` bounds the copied list length."""
    n_rays = rays_o.shape[0]
    dev = rays_o.device
    cnt = torch.zeros(n_rays, dtype=torch.int32, device=dev)
    pts = torch.zeros(max(capacity, 1), dtype=torch.int32, device=dev)
    pos = torch.zeros(max(capacity, 1), 3, dtype=torch.float32, device=dev)
    rid = torch.zeros(max(capacity, 1), dtype=torch.int32, device=dev)
    t = [x.contiguous() for x in (nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities)]
    torch.cuda.synchronize()
    lib().ref_bvh_trace.restype = C.c_int
    n = lib().ref_bvh_trace(n_rays, *[_p(x) for x in t], _p(cnt), int(capacity), _p(pts), _p(pos), _p(rid))
    return n, cnt, pts[:n], pos[:n], rid[:n]


def knn_dist2(points):
    """SimpleKNN::knn of the real reference build; points [P,3] float32 on the GPU -> float32[P]."""
    import torch
    L = lib()
    pts = points.contiguous()
    out = torch.zeros(pts.size(0), dtype=torch.float32, device=pts.device)
    torch.cuda.synchronize()
    L.ref_knn_dist2(C.c_int(pts.size(0)), _p(pts), _p(out))
    return out


# ---- render_equation.cu (the contract model), oracle/_ref/libr3dg_reference_shading.so ----------------------------------
SHADING_LIB_PATH = os.path.join(HERE, "_ref", "libr3dg_reference_shading.so")
_shading_lib = None


def shading_available():
    return os.path.exists(SHADING_LIB_PATH)


def shading_lib():
    global _shading_lib
    if _shading_lib is None:
        _shading_lib = C.CDLL(SHADING_LIB_PATH)       # linked against libtorch (rpath); torch is imported above
    return _shading_lib


def _re_sizes(a):
    return a[0].shape[0], a[5].shape[1], a[6].shape[1], a[7].shape[1]


def render_equation_forward(base_color, roughness, metallic, normals, viewdirs, incidents_shs, direct_shs, visibility_shs,
                            sample_num, is_training=False, rand_float=None):
    """render_equation_forward_cuda of the real reference (render_equation.cu:668-688) -> (pbr, incident_dirs, diffuse_light).
    `rand_float` [P,K,1] replaces the torch::rand table RenderEquationForwardCUDA draws (:711)."""
    a = [x.contiguous() for x in (base_color, roughness, metallic, normals, viewdirs, incidents_shs, direct_shs,
                                  visibility_shs)]
    P, Si, Sd, Sv = _re_sizes(a)
    K = int(sample_num)
    f = dict(dtype=torch.float32, device=base_color.device)
    pbr, dirs, dl = torch.zeros(P, 3, **f), torch.zeros(P, K, 3, **f), torch.zeros(P, 3, **f)
    rnd = rand_float.contiguous() if rand_float is not None else torch.zeros(P, K, 1, **f)
    torch.cuda.synchronize()
    shading_lib().ref_render_equation_forward(P, Si, Sd, Sv, *[_p(x) for x in a], K, int(bool(is_training)), _p(rnd),
                                              _p(dirs), _p(pbr), _p(dl))
    return pbr, dirs, dl


def render_equation_forward_complex(base_color, roughness, metallic, normals, viewdirs, incidents_shs, direct_shs,
                                    visibility_shs, sample_num):
    """render_equation_forward_complex_cuda (render_equation.cu:192-220) -> the 11-tuple of RenderEquationForwardCUDA_complex."""
    a = [x.contiguous() for x in (base_color, roughness, metallic, normals, viewdirs, incidents_shs, direct_shs,
                                  visibility_shs)]
    P, Si, Sd, Sv = _re_sizes(a)
    K = int(sample_num)
    dev = base_color.device

    def z(*s):
        return torch.zeros(*s, dtype=torch.float32, device=dev)
    pbr, dirs, lights, local, glob = z(P, 3), z(P, K, 3), z(P, K, 3), z(P, K, 3), z(P, K, 3)
    vis, diffuse, local_diffuse, accum, rgb_d, rgb_s = z(P, K, 1), z(P, 3), z(P, 3), z(P, 1), z(P, 3), z(P, 3)
    torch.cuda.synchronize()
    shading_lib().ref_render_equation_forward_complex(P, Si, Sd, Sv, *[_p(x) for x in a], K, _p(dirs), _p(pbr), _p(lights),
                                                      _p(local), _p(glob), _p(vis), _p(diffuse), _p(local_diffuse),
                                                      _p(accum), _p(rgb_d), _p(rgb_s))
    return pbr, dirs, lights, local, glob, vis, diffuse, local_diffuse, accum, rgb_d, rgb_s


def render_equation_backward(base_color, roughness, metallic, normals, viewdirs, incidents_shs, direct_shs, visibility_shs,
                             sample_num, incident_dirs, dL_dpbr, dL_ddiffuse_light):
    """render_equation_backward_cuda (render_equation.cu:465-495) -> the 8 gradients of RenderEquationBackwardCUDA.
    dL_ddirect_shs is a plain (non-atomic) read-modify-write from every thread in the reference (quirk Q5): racy."""
    a = [x.contiguous() for x in (base_color, roughness, metallic, normals, viewdirs, incidents_shs, direct_shs,
                                  visibility_shs)]
    P, Si, Sd, Sv = _re_sizes(a)
    outs = [torch.zeros_like(x) for x in a]
    torch.cuda.synchronize()
    shading_lib().ref_render_equation_backward(P, Si, Sd, Sv, *[_p(x) for x in a], int(sample_num),
                                               _p(incident_dirs.contiguous()), _p(dL_dpbr.contiguous()),
                                               _p(dL_ddiffuse_light.contiguous()), *[_p(o) for o in outs])
    return tuple(outs)
