"""CPU oracle for the rasterizer op -- TEST INFRASTRUCTURE ONLY.

Python face of oracle/rasterizer_oracle.c.  `rasterize_gaussians` / `rasterize_gaussians_backward`
take the same arguments as the reference's `_C.rasterize_gaussians` / `_C.rasterize_gaussians_backward`
(r3dg-rasterization/rasterize_points.h:18-71) on CPU tensors / numpy arrays, and return the same tuples,
except that the three opaque byte buffers are replaced by one dict of named intermediates (the reference
keeps their layout private).

The product package never imports this module.
"""
import ctypes as C

import numpy as np

from . import _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
        _lib.r3dgo_scan.restype = C.c_int64
        _lib.r3dgo_getHigherMsb.restype = C.c_uint32
    return _lib


def _np(t, dtype=np.float32):
    if t is None:
        return None
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    a = np.ascontiguousarray(t, dtype=dtype)
    return a


def _p(a):
    if a is None or a.size == 0:
        return None
    return a.ctypes.data_as(C.c_void_p)


def _opt(t):
    a = _np(t)
    if a is None or a.size == 0:
        return None
    return a


def rasterize_gaussians(bg, means3D, features, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, cx, cy, image_height, image_width, sh, degree,
                        campos, prefiltered=False, computer_pseudo_normal=True, debug=False, want_margin=False):
    L = lib()
    means3D = _np(means3D)
    P = means3D.shape[0]
    H, W = int(image_height), int(image_width)
    features = _np(features).reshape(P, -1)
    S = features.shape[1]
    colors, scales, rotations, cov3D_precomp, sh = map(_opt, (colors, scales, rotations, cov3D_precomp, sh))
    opacity = _np(opacity).reshape(-1)
    bg, viewmatrix, projmatrix, campos = map(_np, (bg, viewmatrix, projmatrix, campos))
    M = sh.shape[1] if sh is not None else 0
    N = H * W
    gx, gy = (W + 15) // 16, (H + 15) // 16
    T = gx * gy

    st = dict(P=P, S=S, M=M, H=H, W=W)
    radii = np.zeros(P, np.int32)
    means2D = np.zeros((P, 2), np.float32)
    depths = np.zeros(P, np.float32)
    cov3D = np.zeros((P, 6), np.float32)
    rgb = np.zeros((P, 3), np.float32)
    conic_opacity = np.zeros((P, 4), np.float32)
    tiles_touched = np.zeros(P, np.uint32)
    clamped = np.zeros((P, 3), np.uint8)
    out_color = np.zeros((3, H, W), np.float32)
    out_opacity = np.zeros((1, H, W), np.float32)
    out_depth = np.zeros((1, H, W), np.float32)
    out_feature = np.zeros((S, H, W), np.float32)
    out_normal = np.zeros((3, H, W), np.float32)
    out_xyz = np.zeros((3, H, W), np.float32)
    weights = np.zeros(P, np.float64)
    n_contrib = np.zeros((H, W), np.uint32)
    final_T = np.zeros((H, W), np.float32)
    ranges = np.zeros((T, 2), np.uint32)
    num_rendered = 0
    if P != 0:
        L.r3dgo_preprocess(P, int(degree), M, _p(means3D), _p(scales), C.c_float(scale_modifier), _p(rotations),
                           _p(opacity), _p(sh), _p(cov3D_precomp), _p(colors), _p(viewmatrix), _p(projmatrix),
                           _p(campos), W, H, C.c_float(tan_fovx), C.c_float(tan_fovy), _p(radii), _p(means2D),
                           _p(depths), _p(cov3D), _p(rgb), _p(conic_opacity), _p(tiles_touched), _p(clamped))
        offsets = np.zeros(P, np.uint32)
        num_rendered = int(L.r3dgo_scan(P, _p(tiles_touched), _p(offsets)))
        R = num_rendered
        keys_u = np.zeros(R, np.uint64)
        vals_u = np.zeros(R, np.uint32)
        keys = np.zeros(R, np.uint64)
        vals = np.zeros(R, np.uint32)
        L.r3dgo_duplicate_with_keys(P, _p(means2D), _p(depths), _p(offsets), _p(radii), W, H, _p(keys_u), _p(vals_u))
        bit = int(L.r3dgo_getHigherMsb(C.c_uint32(T)))
        L.r3dgo_sort_pairs(C.c_int64(R), _p(keys_u), _p(vals_u), _p(keys), _p(vals), 32 + bit)
        L.r3dgo_identify_tile_ranges(C.c_int64(R), _p(keys), T, _p(ranges))
        colors_ptr = colors if colors is not None else rgb
        margin = np.zeros((H, W), np.float32) if want_margin else None
        L.r3dgo_render_forward(W, H, S, _p(ranges), _p(vals), _p(means2D), _p(depths), _p(features),
                               _p(colors_ptr), _p(conic_opacity), _p(bg), _p(final_T), _p(n_contrib), _p(out_color),
                               _p(out_opacity), _p(out_depth), _p(out_feature), _p(weights), _p(margin))
        if computer_pseudo_normal:
            fx = np.float32(W) / (np.float32(2.0) * np.float32(tan_fovx))
            fy = np.float32(H) / (np.float32(2.0) * np.float32(tan_fovy))
            L.r3dgo_pseudo_normal(W, H, _p(viewmatrix), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                  _p(out_opacity), _p(out_depth), _p(out_normal), _p(out_xyz))
        st.update(offsets=offsets, keys_unsorted=keys_u, vals_unsorted=vals_u, keys=keys, point_list=vals,
                  sort_bits=32 + bit, margin=margin)
    st.update(radii=radii, means2D=means2D, depths=depths, cov3D=cov3D, rgb=rgb, conic_opacity=conic_opacity,
              tiles_touched=tiles_touched, clamped=clamped, final_T=final_T, n_contrib=n_contrib, ranges=ranges,
              num_rendered=num_rendered)
    return (num_rendered, n_contrib.astype(np.int32), out_color, out_opacity, out_depth, out_feature, out_normal,
            out_xyz, weights.reshape(P, 1), radii, st)


def rasterize_gaussians_backward(bg, means3D, features, radii, colors, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_opacity, dL_dout_depth, dL_dout_feature, sh, degree, campos, state,
                                 backward_geometry=True, debug=False):
    """Returns the reference's 9-tuple (rasterize_points.cu:234) as float64 (accumulated in double) arrays."""
    L = lib()
    st = state
    means3D = _np(means3D)
    P, S, H, W = st["P"], st["S"], st["H"], st["W"]
    features = _np(features).reshape(P, -1)
    colors, scales, rotations, cov3D_precomp, sh = map(_opt, (colors, scales, rotations, cov3D_precomp, sh))
    bg, viewmatrix, projmatrix, campos = map(_np, (bg, viewmatrix, projmatrix, campos))
    M = sh.shape[1] if sh is not None else 0
    dC, dO, dD, dF = map(_np, (dL_dout_color, dL_dout_opacity, dL_dout_depth, dL_dout_feature))
    radii = _np(radii, np.int32)

    d_mean2D = np.zeros((P, 3), np.float64)
    d_conic = np.zeros((P, 4), np.float64)
    d_opacity = np.zeros((P, 1), np.float64)
    d_colors = np.zeros((P, 3), np.float64)
    d_feature = np.zeros((P, S), np.float64)
    d_means3D = np.zeros((P, 3), np.float32)
    d_cov3D = np.zeros((P, 6), np.float32)
    d_sh = np.zeros((P, M, 3), np.float32)
    d_scales = np.zeros((P, 3), np.float32)
    d_rot = np.zeros((P, 4), np.float32)
    if P != 0:
        color_ptr = colors if colors is not None else st["rgb"]
        L.r3dgo_render_backward(W, H, S, _p(st["ranges"]), _p(st["point_list"]), _p(bg), _p(st["means2D"]),
                                _p(st["depths"]), _p(st["conic_opacity"]), _p(color_ptr), _p(features),
                                _p(st["final_T"]), _p(st["n_contrib"]), _p(dC), _p(dO), _p(dD), _p(dF),
                                int(bool(backward_geometry)), _p(d_mean2D), _p(d_conic), _p(d_opacity),
                                _p(d_colors), _p(d_feature))
        fx = np.float32(W) / (np.float32(2.0) * np.float32(tan_fovx))
        fy = np.float32(H) / (np.float32(2.0) * np.float32(tan_fovy))
        cov_ptr = cov3D_precomp if cov3D_precomp is not None else st["cov3D"]
        d_conic32 = d_conic.astype(np.float32)
        d_mean2D32 = d_mean2D.astype(np.float32)
        d_colors32 = d_colors.astype(np.float32)
        L.r3dgo_cov2d_backward(P, _p(means3D), _p(radii), _p(cov_ptr), C.c_float(fx), C.c_float(fy),
                               C.c_float(tan_fovx), C.c_float(tan_fovy), _p(viewmatrix), _p(d_conic32),
                               _p(d_mean2D32), _p(d_means3D), _p(d_cov3D))
        L.r3dgo_preprocess_backward(P, int(degree), M, _p(means3D), _p(radii), _p(sh), _p(st["clamped"]),
                                    _p(scales), _p(rotations), C.c_float(scale_modifier), _p(projmatrix),
                                    _p(campos), _p(d_mean2D32), _p(d_means3D), _p(d_colors32), _p(d_cov3D),
                                    _p(d_sh), _p(d_scales), _p(d_rot))
    return d_mean2D, d_colors, d_opacity, d_means3D, d_feature, d_cov3D, d_sh, d_scales, d_rot, d_conic


def mark_visible(means3D, viewmatrix, projmatrix=None):
    means3D = _np(means3D)
    P = means3D.shape[0]
    present = np.zeros(P, np.uint8)
    if P:
        lib().r3dgo_mark_visible(P, _p(means3D), _p(_np(viewmatrix)), _p(present))
    return present.astype(bool)
