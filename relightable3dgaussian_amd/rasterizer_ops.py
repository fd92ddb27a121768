"""Host-side mirror of the reference's `r3dg_rasterization._C` pybind module (r3dg-rasterization/ext.cpp:15-19):
same function names, argument order, return tuples and error behaviour (RuntimeError), implemented over the
C ABI in libr3dg_hip.so.  PyTorch is used only for device memory and the current HIP stream.

    rasterize_gaussians(...)          -> 13-tuple   (rasterize_points.h:18-42, rasterize_points.cu:36-141)
    rasterize_gaussians_backward(...) -> 9-tuple    (rasterize_points.h:44-71, rasterize_points.cu:143-235)
    mark_visible(means3D, viewmatrix, projmatrix) -> bool[P]   (rasterize_points.h:73-76)
"""
import ctypes as C

import torch

from . import _lib

NUM_CHANNELS = 3


def _f32c(t):
    """contiguous fp32 view on the device (the reference calls .contiguous().data_ptr<float>() on everything)."""
    if t is None or t.numel() == 0:
        return None
    if t.dtype != torch.float32:
        raise RuntimeError("expected a float32 tensor, got %s" % t.dtype)
    if not t.is_cuda:
        raise RuntimeError("expected a CUDA(HIP) tensor")
    return t.contiguous()


class _Resizer:
    """The reference's resizeFunctional (rasterize_points.cu:28-34): three byte buffers grown on request."""

    def __init__(self, device):
        self.device = device
        self.buffers = [torch.empty(0, dtype=torch.uint8, device=device) for _ in range(3)]
        self.callbacks = [_lib.ALLOC_FN(self._make(i)) for i in range(3)]

    # Large requests are rounded up to 16 MiB steps: the binning buffer's size follows num_rendered, which changes with
    # every view, and a caching allocator serves a handful of recurring sizes from its free lists where an ever-new size
    # sooner or later costs a hipMalloc (milliseconds, in the middle of an iteration).  The state buffers are opaque.
    _STEP = 16 << 20

    def _make(self, i):
        def cb(_user, nbytes):
            try:
                n = int(nbytes)
                if n > (4 << 20):
                    n = (n + self._STEP - 1) // self._STEP * self._STEP
                self.buffers[i] = torch.empty(n, dtype=torch.uint8, device=self.device)
                return self.buffers[i].data_ptr()
            except Exception:          # surfaces as R3DG_EALLOC -> RuntimeError
                return 0
        return cb


def _offsets(fn, n, *args):
    arr = (C.c_size_t * n)()
    fn(*args, arr)
    return list(arr)


class _PendingForward:
    """Second half of a split forward (see rasterize_gaussians_begin): call .finish() exactly once."""

    def __init__(self, ticket, rs, outs, dev, H, W, capacity=None):
        self.ticket, self.rs, self.outs, self.dev, self.H, self.W = ticket, rs, outs, dev, H, W
        self.capacity = capacity

    def finish(self, ordering_stream=None):
        """`ordering_stream`: optional torch stream for the instance ordering (sort) part, which then overlaps whatever
        was queued on the forward's stream since begin; the binning buffer is allocated from that stream's pool.
        A bounded forward (rasterize_gaussians_begin(capacity=...)) has its ordering queued already: finish() renders on
        the current stream and returns `capacity` in the num_rendered slot (the value the backward's state layout needs)."""
        L = _lib.lib()
        rendered = C.c_int(0)
        with torch.cuda.device(self.dev):
            if self.capacity is not None:
                st = L.r3dg_rasterize_forward_finish_bounded(self.ticket, _lib.current_stream())
                rendered = C.c_int(int(self.capacity))
            elif ordering_stream is None:
                st = L.r3dg_rasterize_forward_finish(self.ticket, C.byref(rendered))
            else:
                with torch.cuda.stream(ordering_stream):        # resize callback allocates on that stream
                    st = L.r3dg_rasterize_forward_finish_on(self.ticket, C.c_void_p(ordering_stream.cuda_stream),
                                                            C.byref(rendered))
        self.ticket = None
        geomBuffer, binningBuffer, imgBuffer = self.rs.buffers
        # the resize callbacks are closures over the resizer, which holds them: a reference cycle around the frame's scratch
        # buffers that only the garbage collector would break (hundreds of MB of device memory per frame in the meantime).
        # The library does not call them after this point
        self.rs.callbacks = None
        self.rs = None
        _lib.check(st, "rasterize_gaussians")
        out_color, out_opacity, out_depth, out_feature, out_normal, out_surface_xyz, out_weights, radii = self.outs
        H, W = self.H, self.W
        if imgBuffer.numel() == 0:
            imgBuffer = torch.zeros(int(L.r3dg_image_state_bytes(W, H)), dtype=torch.uint8, device=self.dev)
        # n_contrib: int32 view into the image state (the reference returns a from_blob view, rasterize_points.cu:136-139)
        off = _offsets(L.r3dg_image_state_offsets, 3, W, H)
        n_contrib = imgBuffer[off[1]:off[1] + 4 * H * W].view(torch.int32).view(H, W)
        return (rendered.value, n_contrib, out_color, out_opacity, out_depth, out_feature, out_normal, out_surface_xyz,
                out_weights, radii, geomBuffer, binningBuffer, imgBuffer)


def rasterize_gaussians_begin(background, means3D, features, colors, opacity, scales, rotations, scale_modifier,
                              cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, cx, cy, image_height,
                              image_width, sh, degree, campos, prefiltered, computer_pseudo_normal, debug,
                              capacity=None, overflow_flag=None, overflow_count=None, ordering_stream=None,
                              want_weights=True, defer_pseudo_normal=False):
    """First half of rasterize_gaussians (same arguments): projection + asynchronous read-back of num_rendered.  Returns
    an object whose .finish() completes the call and returns the 13-tuple.  Kernels launched on the current stream in
    between (e.g. the ones that fill `features`, whose CONTENTS are first read by .finish()'s kernels) overlap the wait.

    `capacity` (not in the reference): the BOUNDED forward (r3dg_rasterize_forward_begin_bounded) -- no host read-back;
    the projection is queued on the current stream and the instance ordering behind it on `ordering_stream` (default:
    the current stream) at once, with the binning state sized for `capacity` instances.
    `overflow_flag` (float32 tensor, >= 1 element) is set to 1 when the frame needed more and was dropped, else 0;
    `overflow_count` (int32 tensor) counts dropped frames.  All tensors are allocated on the current stream.
    `want_weights=False` (not in the reference): the per-Gaussian blend weights -- which only the densification statistics
    read -- are not computed; the `weights` slot of the result is None.
    `defer_pseudo_normal=True` (not in the reference): the pseudo-normal / surface-point maps are allocated and returned but NOT
    computed (and not zero-filled): the caller runs r3dg_stage2_normals_srgb on them, fused with its own per-pixel pass."""
    L = _lib.lib()
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise RuntimeError("means3D must have dimensions (num_points, 3)")
    if not means3D.is_cuda:
        raise RuntimeError("means3D must be a CUDA(HIP) tensor")
    P = means3D.size(0)
    S = features.size(1)
    H, W = int(image_height), int(image_width)
    dev = means3D.device
    fopt = dict(dtype=torch.float32, device=dev)

    # Every pixel of every image buffer is written by the kernels, so one uninitialised slab suffices (the reference
    # zero-fills eight tensors, rasterize_points.cu:72-79); only `weights` is accumulated with atomics and needs zeros.
    # P == 0 launches nothing, so that case keeps the reference's all-zero outputs.
    slab = (torch.zeros if P == 0 else torch.empty)((11 + S, H, W), **fopt)
    out_color, out_opacity, out_depth = slab[0:3], slab[3:4], slab[4:5]
    out_feature, out_normal, out_surface_xyz = slab[5:5 + S], slab[5 + S:8 + S], slab[8 + S:11 + S]
    if defer_pseudo_normal:
        computer_pseudo_normal = False
    elif not computer_pseudo_normal:
        slab[5 + S:].zero_()
    out_weights = torch.zeros((P, 1), **fopt) if want_weights else None
    radii = torch.empty((P,), dtype=torch.int32, device=dev)

    rs = _Resizer(dev)
    ticket = C.c_void_p(None)
    if P != 0:
        M = sh.size(1) if sh.size(0) != 0 else 0
        t = [_f32c(x) for x in (background, means3D, sh, colors, features, opacity, scales, rotations, cov3D_precomp,
                                viewmatrix, projmatrix, campos)]
        bg_, means_, sh_, col_, feat_, op_, sc_, rot_, cov_, vm_, pm_, cam_ = t
        if feat_ is not None and feat_.data_ptr() != features.data_ptr():
            raise RuntimeError("rasterize_gaussians_begin needs contiguous features (their contents are read later)")
        rs.keep = t                                   # inputs stay alive until finish()
        common = (rs.callbacks[0], rs.callbacks[1], rs.callbacks[2], None, P, S, int(degree), M,
                  _lib.ptr(bg_), W, H, _lib.ptr(means_), _lib.ptr(sh_), _lib.ptr(col_), _lib.ptr(feat_), _lib.ptr(op_),
                  _lib.ptr(sc_), float(scale_modifier), _lib.ptr(rot_), _lib.ptr(cov_), _lib.ptr(vm_), _lib.ptr(pm_),
                  _lib.ptr(cam_), float(tan_fovx), float(tan_fovy), float(cx), float(cy), int(bool(prefiltered)),
                  int(bool(computer_pseudo_normal)), out_color.data_ptr(), out_opacity.data_ptr(), out_depth.data_ptr(),
                  _lib.ptr(out_feature), out_normal.data_ptr(), out_surface_xyz.data_ptr(),
                  out_weights.data_ptr() if out_weights is not None else None,
                  radii.data_ptr(), int(bool(debug)))
        with torch.cuda.device(dev):
            if capacity is None:
                st = L.r3dg_rasterize_forward_begin(_lib.current_stream(), *common, C.byref(ticket))
            else:
                if overflow_flag is not None and (overflow_flag.dtype != torch.float32 or not overflow_flag.is_cuda):
                    raise RuntimeError("overflow_flag must be a float32 device tensor")
                if overflow_count is not None and (overflow_count.dtype != torch.int32 or not overflow_count.is_cuda):
                    raise RuntimeError("overflow_count must be an int32 device tensor")
                rs.keep.append((overflow_flag, overflow_count))
                st = L.r3dg_rasterize_forward_begin_bounded(
                    _lib.current_stream(), *common,
                    C.c_void_p(ordering_stream.cuda_stream) if ordering_stream is not None else None, int(capacity),
                    overflow_flag.data_ptr() if overflow_flag is not None else None,
                    overflow_count.data_ptr() if overflow_count is not None else None, C.byref(ticket))
        _lib.check(st, "rasterize_gaussians")
    elif capacity is not None and overflow_flag is not None:
        overflow_flag[:1].zero_()
    return _PendingForward(ticket, rs, (out_color, out_opacity, out_depth, out_feature, out_normal, out_surface_xyz,
                                        out_weights, radii), dev, H, W, capacity=capacity)


def rasterize_gaussians(background, means3D, features, colors, opacity, scales, rotations, scale_modifier,
                        cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, cx, cy, image_height, image_width,
                        sh, degree, campos, prefiltered, computer_pseudo_normal, debug):
    return rasterize_gaussians_begin(background, means3D, features, colors, opacity, scales, rotations, scale_modifier,
                                     cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, cx, cy, image_height,
                                     image_width, sh, degree, campos, prefiltered, computer_pseudo_normal,
                                     debug).finish()


def rasterize_gaussians_backward(background, means3D, features, radii, colors, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                 dL_dout_opacity, dL_dout_depth, dL_dout_feature, sh, degree, campos, geomBuffer, R,
                                 binningBuffer, imageBuffer, backward_geometry, debug, dL_dsh_out=None,
                                 geometry_stream=None, active_features=None, zeroed_accumulators=None):
    """`dL_dsh_out` (not in the reference signature): optional preallocated [P,M,3] buffer the SH gradient is written
    into (every element is written), e.g. a view of a flat gradient bucket.  `geometry_stream`: optional torch stream
    for the per-Gaussian geometry backward (dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations are then only
    valid after the caller joins that stream; see r3dg_rasterize_backward_split).  `active_features`: optional list of
    the feature channels whose upstream gradient can be non-zero -- the caller's promise that all other channels of
    dL_dout_feature are zero; they are then skipped by the tile kernel (same results).  `zeroed_accumulators`: optional float32
    tensor of (11 + S) * P floats (any contents since round 5 -- the name is from when it had to be zeros) that takes the place
    of the slab allocated here; dL_dmeans2D, dL_dconic, dL_dopacity, dL_dcolors and dL_dfeatures are views of it."""
    L = _lib.lib()
    P = means3D.size(0)
    S = features.size(1)
    H = dL_dout_color.size(1)
    W = dL_dout_color.size(2)
    M = sh.size(1) if sh.size(0) != 0 else 0
    dev = means3D.device
    fopt = dict(dtype=torch.float32, device=dev)

    # gradients accumulated with atomics share ONE zero-filled slab; the per-Gaussian outputs of the fused
    # preprocess-backward kernel are fully written (zeros for invisible Gaussians) and start uninitialised
    # (dL_dfeatures first: its rows stay 16-byte aligned for S % 4 == 0 whatever P is)
    if zeroed_accumulators is not None:
        acc = zeroed_accumulators
        if acc.numel() != (11 + S) * P or acc.dtype != torch.float32 or not acc.is_contiguous() or acc.device != dev:
            raise RuntimeError("zeroed_accumulators must be a contiguous float32 tensor of (11 + S) * P floats on the device")
    else:
        # (round 5: the tile backward accumulates into per-Gaussian records of the library and its scatter pass WRITES every
        # element of these five arrays -- zeros for Gaussians nothing touched, and for P == 0 / num_rendered == 0 the library
        # fills them -- so the slab starts uninitialised; rounds 1-4 zero-filled it here, the reference fills five tensors,
        # rasterize_points.cu:183-192)
        acc = torch.empty((11 + S) * P, **fopt)
    dL_dfeatures = acc[0:S * P].view(P, S)
    o = S * P
    dL_dconic = acc[o:o + 4 * P].view(P, 2, 2); o += 4 * P
    dL_dmeans2D = acc[o:o + 3 * P].view(P, 3); o += 3 * P
    dL_dopacity = acc[o:o + P].view(P, 1); o += P
    dL_dcolors = acc[o:o + 3 * P].view(P, NUM_CHANNELS)
    have_sh = sh.numel() != 0 and colors.numel() == 0
    have_scale = scales.numel() != 0 and cov3D_precomp.numel() == 0
    dL_dmeans3D = torch.empty((P, 3), **fopt)
    dL_dcov3D = torch.empty((P, 6), **fopt)
    if dL_dsh_out is not None and have_sh:
        if tuple(dL_dsh_out.shape) != (P, M, 3) or not dL_dsh_out.is_contiguous() or dL_dsh_out.dtype != torch.float32:
            raise RuntimeError("dL_dsh_out must be a contiguous float32 [P,M,3] tensor")
        dL_dsh = dL_dsh_out
    else:
        dL_dsh = (torch.empty if have_sh else torch.zeros)((P, M, 3), **fopt)
    dL_dscales = (torch.empty if have_scale else torch.zeros)((P, 3), **fopt)
    dL_drotations = (torch.empty if have_scale else torch.zeros)((P, 4), **fopt)

    if P != 0:
        t = [_f32c(x) for x in (background, means3D, sh, features, colors, scales, rotations, cov3D_precomp, viewmatrix,
                                projmatrix, campos, dL_dout_color, dL_dout_opacity, dL_dout_depth, dL_dout_feature)]
        bg_, means_, sh_, feat_, col_, sc_, rot_, cov_, vm_, pm_, cam_, gC, gO, gD, gF = t
        radii_ = radii.contiguous()
        with torch.cuda.device(dev):
            cur = _lib.current_stream()
            gs = cur if geometry_stream is None else C.c_void_p(geometry_stream.cuda_stream)
            if active_features is None:
                n_act, act = -1, None
            else:
                n_act = len(active_features)
                act = (C.c_int * max(1, n_act))(*[int(a) for a in active_features])
            st = L.r3dg_rasterize_backward_split(
                cur, gs, P, S, int(degree), M, int(R), _lib.ptr(bg_), W, H, _lib.ptr(means_),
                _lib.ptr(sh_), _lib.ptr(feat_), _lib.ptr(col_), _lib.ptr(sc_), float(scale_modifier), _lib.ptr(rot_),
                _lib.ptr(cov_), _lib.ptr(vm_), _lib.ptr(pm_), _lib.ptr(cam_), float(tan_fovx), float(tan_fovy),
                radii_.data_ptr(), _lib.ptr(geomBuffer), _lib.ptr(binningBuffer), _lib.ptr(imageBuffer), _lib.ptr(gC),
                _lib.ptr(gO), _lib.ptr(gD), _lib.ptr(gF), dL_dmeans2D.data_ptr(), dL_dconic.data_ptr(),
                dL_dopacity.data_ptr(), dL_dcolors.data_ptr(), _lib.ptr(dL_dfeatures), dL_dmeans3D.data_ptr(),
                dL_dcov3D.data_ptr(), _lib.ptr(dL_dsh), dL_dscales.data_ptr(), dL_drotations.data_ptr(),
                int(bool(backward_geometry)), int(bool(debug)), n_act, act)
        _lib.check(st, "rasterize_gaussians_backward")
    return (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dfeatures, dL_dcov3D, dL_dsh, dL_dscales,
            dL_drotations)


def rasterize_gaussians_backward_features(P, S, image_height, image_width, dL_dout_feature, geomBuffer, R, binningBuffer,
                                          imageBuffer, debug=False, active_features=None):
    """The backward for FROZEN geometry (not in the reference; r3dg_rasterize_backward_features): only dL_dfeatures [P,S]
    from dL_dout_feature [S,H,W] -- what is left of rasterize_gaussians_backward when positions, covariances, opacities and
    SH colour have learning rate 0 (script/run_syn4.sh:27-33, run_dtu.sh).  Same state buffers and `active_features`
    contract; equals that call's dL_dfeatures up to the order of the float atomics."""
    L = _lib.lib()
    dev = dL_dout_feature.device
    dL_dfeatures = torch.zeros((P, S), dtype=torch.float32, device=dev)
    if P != 0 and int(R) != 0:
        gF = _f32c(dL_dout_feature)
        if active_features is None:
            n_act, act = -1, None
        else:
            n_act = len(active_features)
            act = (C.c_int * max(1, n_act))(*[int(a) for a in active_features])
        with torch.cuda.device(dev):
            st = L.r3dg_rasterize_backward_features(
                _lib.current_stream(), int(P), int(S), int(R), int(image_width), int(image_height), _lib.ptr(geomBuffer),
                _lib.ptr(binningBuffer), _lib.ptr(imageBuffer), gF.data_ptr(), dL_dfeatures.data_ptr(), n_act, act,
                int(bool(debug)))
        _lib.check(st, "rasterize_gaussians_backward_features")
    return dL_dfeatures


def mark_visible(means3D, viewmatrix, projmatrix):
    L = _lib.lib()
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=means3D.device)
    if P != 0:
        m, v, p = _f32c(means3D), _f32c(viewmatrix), _f32c(projmatrix)
        with torch.cuda.device(means3D.device):
            st = L.r3dg_mark_visible(_lib.current_stream(), P, m.data_ptr(), v.data_ptr(), p.data_ptr(),
                                     present.data_ptr())
        _lib.check(st, "mark_visible")
    return present


def num_rendered_of(geomBuffer, P):
    """0-d int64 DEVICE tensor viewing the instance count of the forward that wrote `geomBuffer` (what a bounded forward
    does not hand to the host)."""
    off = int(_lib.lib().r3dg_geometry_state_total_offset(int(P)))
    return geomBuffer[off:off + 8].view(torch.int64)[0]


# ---- helpers for tests / debugging: decode the opaque state buffers ---------------------------------------
def decode_state(geomBuffer, binningBuffer, imgBuffer, P, R, H, W):
    L = _lib.lib()
    T = ((W + 15) // 16) * ((H + 15) // 16)
    g = _offsets(L.r3dg_geometry_state_offsets, 9, P)
    i = _offsets(L.r3dg_image_state_offsets, 3, W, H)
    b = _offsets(L.r3dg_binning_state_offsets, 4, R)

    def view(buf, off, count, dtype, shape):
        nb = count * torch.empty((), dtype=dtype).element_size()
        return buf[off:off + nb].view(dtype).view(*shape)
    out = dict(
        depths=view(geomBuffer, g[0], P, torch.float32, (P,)),
        clamped=view(geomBuffer, g[1], 3 * P, torch.uint8, (P, 3)),
        means2D=view(geomBuffer, g[3], 2 * P, torch.float32, (P, 2)),
        cov3D=view(geomBuffer, g[4], 6 * P, torch.float32, (P, 6)),
        conic_opacity=view(geomBuffer, g[5], 4 * P, torch.float32, (P, 4)),
        rgb=view(geomBuffer, g[6], 3 * P, torch.float32, (P, 3)),
        tiles_touched=view(geomBuffer, g[7], P, torch.int32, (P,)),
        point_offsets=view(geomBuffer, g[8], P, torch.int32, (P,)),
        final_T=view(imgBuffer, i[0], H * W, torch.float32, (H, W)),
        n_contrib=view(imgBuffer, i[1], H * W, torch.int32, (H, W)),
        ranges=view(imgBuffer, i[2], 2 * T, torch.int32, (T, 2)),
    )
    if R > 0:
        out.update(
            keys_unsorted=view(binningBuffer, b[0], R, torch.int64, (R,)),
            keys=view(binningBuffer, b[1], R, torch.int64, (R,)),
            vals_unsorted=view(binningBuffer, b[2], R, torch.int32, (R,)),
            point_list=view(binningBuffer, b[3], R, torch.int32, (R,)),
        )
    return out
