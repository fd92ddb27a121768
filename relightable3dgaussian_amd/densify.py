"""Host mirror of the reference's densification bookkeeping over the C ABI of libr3dg_hip.so.

Same names, argument meaning and order of effects as `GaussianModel` (scene/gaussian_model.py):
    DensificationStats.add(...)          add_densification_stats (:931-937) + train.py:164-165 (max radii)
    densify_and_prune(...)               densify_and_prune (:893-915) = clone (:846-891) -> split (:790-844) -> prune
    prune(...)                           prune (:917-929)
    reset_opacity(...)                   reset_opacity (:563-566)
but each call is one plan (every decision as a destination-row -> source-row map) + ONE gather launch that writes every
parameter group and both Adam moments once, where the reference runs torch.cat twice and boolean-mask indexing twice over
each of them.  There is no fallback path: a missing library or a CPU tensor raises.

`groups` is an ordered dict  name -> {"param": [P, ...] fp32, "exp_avg": tensor | None, "exp_avg_sq": tensor | None}
holding the RAW (pre-activation) parameters; it must contain "xyz", "scaling", "rotation" and "opacity".  Calls return
new dicts of the same shape plus fresh statistics; the inputs are left untouched (the caller drops them).

Randomness of the split (torch.normal, :806) stays with PyTorch so that data-parallel replicas sharing a seed stay
identical: the standard-normal table is drawn with `torch.randn(n_split * n_selected, 3)` -- one row per sample the
reference would draw, in the same order -- and scaled by the Gaussian's own scale inside the gather kernel.
"""
import ctypes as C
import os

import torch

from . import _lib

_f, _i, _p = C.c_float, C.c_int, C.c_void_p


class DensifyConfig(C.Structure):
    _fields_ = [("mode", C.c_int32), ("n_split", C.c_int32), ("grad_threshold", _f), ("grad_normal_threshold", _f),
                ("min_opacity", _f), ("weights_threshold", _f), ("dense_size", _f), ("world_size_limit", _f),
                ("split_divisor", _f), ("max_screen_size", _f)]


class DensifyGroup(C.Structure):
    _fields_ = [("src_param", _p), ("src_exp_avg", _p), ("src_exp_avg_sq", _p), ("dst_param", _p),
                ("dst_exp_avg", _p), ("dst_exp_avg_sq", _p), ("row_floats", C.c_uint32), ("role", C.c_uint32)]


MAX_GROUPS = 20          # R3DG_DENSIFY_MAX_GROUPS (24) minus the four statistics rows prune() adds
ROLE = {"xyz": 1, "scaling": 2}


def _need_device(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a device tensor (there is no CPU path)" % what)
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("%s must be contiguous float32" % what)


class DensificationStats:
    """The five per-Gaussian statistics of GaussianModel.training_setup (:459-463) / max_radii2D (:441), as flat [P]
    tensors (the reference keeps [P,1] columns; `column()` returns that view)."""
    NAMES = ("xyz_gradient_accum", "normal_gradient_accum", "denom", "weights_accum", "max_radii2D")

    def __init__(self, P, device):
        self._slab = torch.zeros(5, P, dtype=torch.float32, device=device)
        self.P = P

    xyz_gradient_accum = property(lambda self: self._slab[0])
    normal_gradient_accum = property(lambda self: self._slab[1])
    denom = property(lambda self: self._slab[2])
    weights_accum = property(lambda self: self._slab[3])
    max_radii2D = property(lambda self: self._slab[4])

    def column(self, name):
        return getattr(self, name).view(-1, 1)

    def add(self, viewspace_grad, normal_grad, radii, weights, skip_flag=None):
        """`skip_flag`: float32 device tensor; non-zero (read on the device) = this view was dropped by a bounded forward and
        is no observation.
        One view's contribution.  viewspace_grad [P,3] = gradient slot of the screen-space dummy, normal_grad [P,3] =
        gradient of the raw normals (or None), radii int32 [P], weights [P,1] | [P] from the rasterizer."""
        P = self.P
        _need_device(viewspace_grad, "viewspace_grad")
        _need_device(weights, "weights")
        if normal_grad is not None:
            _need_device(normal_grad, "normal_grad")
        if not radii.is_cuda or radii.dtype != torch.int32 or not radii.is_contiguous():
            raise RuntimeError("radii must be a contiguous int32 device tensor")
        if viewspace_grad.shape != (P, 3) or radii.numel() != P or weights.numel() != P or (
                normal_grad is not None and normal_grad.shape != (P, 3)):
            raise RuntimeError("densification statistics: shape mismatch with P=%d" % P)
        L = _lib.lib()
        with torch.cuda.device(self._slab.device):
            st = L.r3dg_densify_accumulate(
                _lib.current_stream(), P, viewspace_grad.data_ptr(), _lib.ptr(normal_grad), radii.data_ptr(),
                weights.data_ptr(), self.xyz_gradient_accum.data_ptr(), self.normal_gradient_accum.data_ptr(),
                self.denom.data_ptr(), self.weights_accum.data_ptr(), self.max_radii2D.data_ptr(),
                skip_flag.data_ptr() if skip_flag is not None else None)
        _lib.check(st, "densify_accumulate")

    def all_reduce(self, group=None):
        """Data parallel: make the statistics (and hence every densify decision) identical on all ranks -- one sum over
        the four accumulators and one max over the radii."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return
        if dist.get_world_size(group) == 1 and os.environ.get("R3DG_DP_SINGLE_RANK") != "1":     # (fused_step._world_of)
            return
        h = dist.all_reduce(self._slab[:4], group=group, async_op=True)
        dist.all_reduce(self._slab[4], op=dist.ReduceOp.MAX, group=group)
        h.wait()


def _check_groups(groups):
    for k in ("xyz", "scaling", "rotation", "opacity"):
        if k not in groups:
            raise RuntimeError("densify: parameter group %r is required" % k)
    if len(groups) > MAX_GROUPS:
        raise RuntimeError("densify: at most %d parameter groups" % MAX_GROUPS)
    P = groups["xyz"]["param"].shape[0]
    for name, g in groups.items():
        p = g["param"]
        _need_device(p, "parameter %r" % name)
        if p.shape[0] != P:
            raise RuntimeError("densify: group %r has %d rows, expected %d" % (name, p.shape[0], P))
        m, v = g.get("exp_avg"), g.get("exp_avg_sq")
        if (m is None) != (v is None):
            raise RuntimeError("densify: group %r needs both Adam moments or neither" % name)
        if m is not None:
            _need_device(m, "exp_avg of %r" % name)
            _need_device(v, "exp_avg_sq of %r" % name)
            if m.shape != p.shape or v.shape != p.shape:
                raise RuntimeError("densify: moments of %r do not match the parameter" % name)
    if groups["xyz"]["param"].shape != (P, 3) or groups["scaling"]["param"].shape != (P, 3) or \
            groups["rotation"]["param"].shape != (P, 4) or groups["opacity"]["param"].numel() != P:
        raise RuntimeError("densify: xyz/scaling are [P,3], rotation [P,4], opacity [P,1]")
    return P


def _run(groups, stats, cfg, generator, normal_table=None):
    P = _check_groups(groups)
    if stats.P != P:
        raise RuntimeError("densify: statistics hold %d rows, parameters %d" % (stats.P, P))
    L = _lib.lib()
    dev = groups["xyz"]["param"].device
    cap = max(2, cfg.n_split) * max(P, 1)
    rowmap = torch.empty(2, cap, dtype=torch.int32, device=dev)
    counts = torch.zeros(8, dtype=torch.int32, device=dev)
    temp = torch.empty(max(1, L.r3dg_densify_temp_bytes(P)), dtype=torch.uint8, device=dev)
    xyz, scaling, rotation = groups["xyz"]["param"], groups["scaling"]["param"], groups["rotation"]["param"]
    with torch.cuda.device(dev):
        st = L.r3dg_densify_plan(
            _lib.current_stream(), P, C.addressof(cfg), scaling.data_ptr() if P else None,
            groups["opacity"]["param"].data_ptr() if P else None, stats.xyz_gradient_accum.data_ptr() if P else None,
            stats.normal_gradient_accum.data_ptr() if P else None, stats.denom.data_ptr() if P else None,
            stats.weights_accum.data_ptr() if P else None, stats.max_radii2D.data_ptr() if P else None,
            rowmap[0].data_ptr(), rowmap[1].data_ptr(), counts.data_ptr(), temp.data_ptr())
        _lib.check(st, "densify_plan")
        # the one read-back of the call (the reference syncs on every boolean-mask index): sizes of the new tensors
        n_out, n_keep, n_clone, n_split_all, n_child = counts[:5].tolist()
        table = None
        if cfg.mode == 0 and normal_table is not None:
            _need_device(normal_table, "normal_table")
            if normal_table.dim() != 2 or normal_table.shape[1] != 3 or normal_table.shape[0] < cfg.n_split * n_split_all:
                raise RuntimeError("normal_table must hold at least n_split * %d rows of 3" % n_split_all)
            table = normal_table
        elif cfg.mode == 0:
            # one standard-normal row per sample torch.normal would draw (:806), same order: block b, selected rank r
            table = torch.randn(cfg.n_split * n_split_all, 3, dtype=torch.float32, device=dev, generator=generator)
        out = {}
        # prune() keeps the gradient statistics of the survivors (prune_points :713-717): four more one-float groups
        new_stats = DensificationStats(n_out, dev)
        carried = ("xyz_gradient_accum", "normal_gradient_accum", "denom", "max_radii2D") if cfg.mode == 1 else ()
        arr = (DensifyGroup * (len(groups) + len(carried)))()
        for j, name in enumerate(carried):
            arr[len(groups) + j] = DensifyGroup(getattr(stats, name).data_ptr() if P else None, None, None,
                                                getattr(new_stats, name).data_ptr(), None, None, 1, 0)
        for j, (name, g) in enumerate(groups.items()):
            p = g["param"]
            row = int(p[0].numel()) if P else int(torch.Size(p.shape[1:]).numel())
            dst = {"param": torch.empty((n_out,) + tuple(p.shape[1:]), dtype=torch.float32, device=dev)}
            has_m = g.get("exp_avg") is not None
            dst["exp_avg"] = torch.empty_like(dst["param"]) if has_m else None
            dst["exp_avg_sq"] = torch.empty_like(dst["param"]) if has_m else None
            out[name] = dst
            arr[j] = DensifyGroup(p.data_ptr(), g["exp_avg"].data_ptr() if has_m else None,
                                  g["exp_avg_sq"].data_ptr() if has_m else None, dst["param"].data_ptr(),
                                  dst["exp_avg"].data_ptr() if has_m else None,
                                  dst["exp_avg_sq"].data_ptr() if has_m else None, max(row, 1), ROLE.get(name, 0))
        st = L.r3dg_densify_gather(
            _lib.current_stream(), n_out, rowmap[0].data_ptr(), rowmap[1].data_ptr(), len(arr),
            C.cast(arr, C.c_void_p), xyz.data_ptr() if P else None, scaling.data_ptr() if P else None,
            rotation.data_ptr() if P else None, _lib.ptr(table), cfg.split_divisor if cfg.mode == 0 else 1.0)
        _lib.check(st, "densify_gather")
    # densify: every statistic restarts at zero -- densification_postfix (:774-779) zeroes the accumulators and the radii;
    # both entry points end with `weights_accum[:] = 0` (:912, :927)
    info = dict(rows_out=n_out, kept=n_keep, cloned=n_clone, split=n_split_all, split_surviving=n_child,
                src_row=rowmap[0, :n_out], kind=rowmap[1, :n_out], normal_table=table)
    return out, new_stats, info


def _f32(x):
    return float(torch.tensor(float(x), dtype=torch.float32))


def densify_and_prune(groups, stats, max_grad, min_opacity, extent, max_screen_size, max_grad_normal, percent_dense,
                      weights_threshold=1e-4, n_split=2, generator=None, normal_table=None):
    """GaussianModel.densify_and_prune(max_grad, min_opacity, extent, max_screen_size, max_grad_normal) with
    `percent_dense` from training_setup (:458).  Returns (new_groups, new_stats, info).  `normal_table` (tests) replaces
    the torch.randn draw by a given standard-normal table."""
    cfg = DensifyConfig(0, int(n_split), max_grad, max_grad_normal, min_opacity, weights_threshold,
                        _f32(percent_dense * extent), _f32(0.1 * extent), _f32(0.8 * n_split),
                        float(max_screen_size) if max_screen_size else 0.0)
    return _run(groups, stats, cfg, generator, normal_table)


def prune(groups, stats, min_opacity, extent, max_screen_size, weights_threshold=1e-4):
    """GaussianModel.prune(min_opacity, extent, max_screen_size)."""
    cfg = DensifyConfig(1, 2, 0.0, 0.0, min_opacity, weights_threshold, 0.0, _f32(0.1 * extent), 1.0,
                        float(max_screen_size) if max_screen_size else 0.0)
    return _run(groups, stats, cfg, None)


def reset_opacity(opacity, exp_avg=None, exp_avg_sq=None):
    """GaussianModel.reset_opacity(): in place on the raw opacity and its Adam moments."""
    _need_device(opacity, "opacity")
    for t in (exp_avg, exp_avg_sq):
        if t is not None:
            _need_device(t, "opacity moment")
    with torch.cuda.device(opacity.device):
        st = _lib.lib().r3dg_reset_opacity(_lib.current_stream(), opacity.numel(), opacity.data_ptr(),
                                           _lib.ptr(exp_avg), _lib.ptr(exp_avg_sq))
    _lib.check(st, "reset_opacity")
