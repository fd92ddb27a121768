"""ctypes binding of libr3dg_hip.so (include/r3dg_hip.h).  There is NO fallback: if the HIP library is missing
or a call fails, the op raises -- a silent CPU/eager path would void every parity claim."""
import ctypes as C
import threading
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# R3DG_LIB_PATH: an experiment build of the SAME library (tools/build_variant.py) for A/B measurements; never a fallback
LIB_PATH = os.environ.get("R3DG_LIB_PATH") or os.path.join(_HERE, "lib", "libr3dg_hip.so")

ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)

_lib = None

_f = C.c_float
_i = C.c_int
_p = C.c_void_p

_SIGNATURES = {
    "r3dg_last_error": (C.c_char_p, []),
    "r3dg_version": (_i, []),
    "r3dg_release_scratch": (_i, []),
    "r3dg_max_features_forward": (_i, []),
    "r3dg_max_features_backward": (_i, []),
    "r3dg_bounded_forward_supported": (_i, [_i, _i]),
    "r3dg_geometry_state_bytes": (C.c_size_t, [_i]),
    "r3dg_image_state_bytes": (C.c_size_t, [_i, _i]),
    "r3dg_binning_state_bytes": (C.c_size_t, [C.c_int64]),
    "r3dg_geometry_state_offsets": (_i, [_i, C.POINTER(C.c_size_t)]),
    "r3dg_geometry_state_total_offset": (C.c_size_t, [_i]),
    "r3dg_image_state_offsets": (_i, [_i, _i, C.POINTER(C.c_size_t)]),
    "r3dg_binning_state_offsets": (_i, [C.c_int64, C.POINTER(C.c_size_t)]),
    "r3dg_rasterize_forward": (_i, [_p, ALLOC_FN, ALLOC_FN, ALLOC_FN, _p, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p,
                                    _p, _p, _f, _p, _p, _p, _p, _p, _f, _f, _f, _f, _i, _i, _p, _p, _p, _p, _p, _p,
                                    _p, _p, _i, C.POINTER(_i)]),
    "r3dg_rasterize_forward_begin": (_i, [_p, ALLOC_FN, ALLOC_FN, ALLOC_FN, _p, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p,
                                    _p, _p, _f, _p, _p, _p, _p, _p, _f, _f, _f, _f, _i, _i, _p, _p, _p, _p, _p, _p,
                                    _p, _p, _i, C.POINTER(_p)]),
    "r3dg_rasterize_forward_begin_bounded": (_i, [_p, ALLOC_FN, ALLOC_FN, ALLOC_FN, _p, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p,
                                            _p, _p, _p, _f, _p, _p, _p, _p, _p, _f, _f, _f, _f, _i, _i, _p, _p, _p, _p,
                                            _p, _p, _p, _p, _i, _p, C.c_longlong, _p, _p, C.POINTER(_p)]),
    "r3dg_rasterize_forward_finish_bounded": (_i, [_p, _p]),
    "r3dg_rasterize_forward_finish": (_i, [_p, C.POINTER(_i)]),
    "r3dg_rasterize_forward_finish_on": (_i, [_p, _p, C.POINTER(_i)]),
    "r3dg_rasterize_backward": (_i, [_p, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p,
                                     _f, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                     _i, _i]),
    "r3dg_rasterize_backward_split": (_i, [_p, _p, _i, _i, _i, _i, _i, _p, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p,
                                           _p, _f, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                           _p, _p, _i, _i, _i, C.POINTER(_i)]),
    "r3dg_rasterize_backward_features": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _i, C.POINTER(_i), _i]),
    "r3dg_mark_visible": (_i, [_p, _i, _p, _p, _p, _p]),
    "r3dg_bvh_trace_count": (_i, [_p, C.c_int64, _p, _p, _p, _p, _p, _p]),
    "r3dg_bvh_trace_fill": (_i, [_p, C.c_int64] + [_p] * 11),
    "r3dg_sort_temp_bytes": (C.c_size_t, [C.c_int64]),
    "r3dg_sort_pairs": (_i, [_p, C.c_int64, _p, _p, _p, _p, _i, _p]),
    "r3dg_set_option": (_i, [_i, _i]),
    "r3dg_context_create": (_p, []),
    "r3dg_context_destroy": (None, [_p]),
    "r3dg_context_set_option": (_i, [_p, _i, _i]),
    "r3dg_context_make_current": (_i, [_p, C.POINTER(_p)]),
    "r3dg_get_option": (_i, [_i, C.POINTER(_i)]),
    "r3dg_selftest_transpose_reduce": (_i, [_p, _i, _i, _p, _p, _p, _p]),
    "r3dg_shade_forward": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p]),
    "r3dg_shade_forward_cached": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _f, _p, _i, _p]),
    "r3dg_shade_build_taps": (_i, [_p, C.c_int64, _p, _p, _i, _i, _p, _p]),
    "r3dg_shade_frs_supported": (_i, [_i, _i, _i, _i]),
    "r3dg_shade_frs_tables_bytes": (C.c_size_t, [_i]),
    "r3dg_shade_frs_build_tables": (_i, [_p, _i, _p, _p]),
    "r3dg_shade_frs_classify": (_i, [_p, _i, _p, _p]),
    "r3dg_shade_frs_rotate": (_i, [_p, _i, _p, _p, _p]),
    "r3dg_stream_wait_stream": (_i, [_p, _p]),
    "r3dg_spin": (_i, [_p, _f]),
    "r3dg_store_u64_to_host": (_i, [_p, _p, _p]),
    "r3dg_shade_frs_build_taps": (_i, [_p, _i, _i, _p, _p, _i, _i, _p]),
    "r3dg_shade_frs_forward": (_i, [_p, _i, _i] + [_p] * 6 + [_i, _i, _p, _f] + [_p] * 6 + [_i, _p, _i, _p, _p, _p]),
    "r3dg_shade_frs_backward": (_i, [_p, _i, _i] + [_p] * 6 + [_i, _i, _p, _f] + [_p] * 6 + [_i] + [_p] * 10 + [_i, _p]),
    "r3dg_shade_frs_incident_chain": (_i, [_p, _i] + [_p] * 8 + [_f] * 5 + [_i, _f, _p, _i]),
    "r3dg_shade_build_transport": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p, _p]),
    "r3dg_shade_forward_transport": (_i, [_p, _i, _i] + [_p] * 9),
    "r3dg_shade_build_split": (_i, [_p, _i, _i] + [_p] * 6 + [_f, _p, _p, _p]),
    "r3dg_shade_env_footprints_bytes": (C.c_size_t, [_i, _i]),
    "r3dg_shade_env_footprints": (_i, [_p, _i, _i, _p, _p]),
    "r3dg_shade_forward_split": (_i, [_p, _i, _i] + [_p] * 11 + [_i, _i, _p]),
    "r3dg_shade_backward": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                 _p, _p]),
    "r3dg_shade_backward_cached": (_i, [_p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                        _p, _p, _p, _p, _i]),
    "r3dg_render_equation_forward": (_i, [_p, _i, _i, _i, _i] + [_p] * 8 + [_i] + [_p] * 4),
    "r3dg_render_equation_forward_complex": (_i, [_p, _i, _i, _i, _i] + [_p] * 8 + [_i] + [_p] * 11),
    "r3dg_render_equation_backward": (_i, [_p, _i, _i, _i, _i] + [_p] * 8 + [_i] + [_p] * 11),
    "r3dg_clock_probe": (_i, [_p, _i, _p, _p, C.POINTER(C.c_int)]),
    "r3dg_stage2_activate": (_i, [_p, _i] + [_p] * 17),
    "r3dg_stage2_activate_with": (_i, [_p, _i] + [_p] * 17 + [_i, _p, _p, _p, _i]),
    "r3dg_stage2_pack_features": (_i, [_p, _i] + [_p] * 8),
    "r3dg_stage2_unpack_gradients": (_i, [_p, _i, _p, _p, _f, _p, _p, _p, _p]),
    "r3dg_stage2_activate_backward": (_i, [_p, _i] + [_p] * 24),
    "r3dg_stage2_activate_backward_with": (_i, [_p, _i] + [_p] * 24 + [_i, _i, _p, _p, _p, _f, _p, _p, _i]),
    "r3dg_stage2_loss": (_i, [_p, _i, _i] + [_p] * 8 + [_f, _f, _f] + [_p] * 6 + [_i]),
    "r3dg_stage2_smooth_forward": (_i, [_p, _i, _i] + [_p] * 5 + [_f, _f, _f, _p, _p]),
    "r3dg_stage2_smooth_backward": (_i, [_p, _i, _i] + [_p] * 5 + [_f, _f, _f, _i, _p, _p]),
    "r3dg_stage2_smooth_fused": (_i, [_p, _i, _i] + [_p] * 5 + [_f, _f, _f, _i, _p, _p, _p]),
    "r3dg_stage2_pbr_srgb": (_i, [_p, _i, _i] + [_p] * 5),
    "r3dg_stage2_normals_srgb": (_i, [_p, _i, _i, _p, _f, _f, _f, _f] + [_p] * 8),
    "r3dg_ssim_forward": (_i, [_p, _i, _i, _i, _p, _p, _p, _p]),
    "r3dg_ssim_backward": (_i, [_p, _i, _i, _i, _p, _p, _p, _f, _p]),
    "r3dg_ssim_forward_pair": (_i, [_p, _i, _i, _i] + [_p] * 7),
    "r3dg_ssim_backward_pair": (_i, [_p, _i, _i, _i] + [_p] * 5 + [_f, _f, _p, _p]),
    "r3dg_stage1_pack_features": (_i, [_p, _i, _p, _p, _p, _p]),
    "r3dg_stage1_loss": (_i, [_p, _i, _i] + [_p] * 7 + [_f] * 5 + [_p] * 6),
    "r3dg_stage1_activate_backward": (_i, [_p, _i] + [_p] * 16),
    "r3dg_stage2_env_backward": (_i, [_p, _i, _i, _p, _p, _p, _f, _p, _p, _i]),
    "r3dg_adam_step": (_i, [_p, _i, _p, _f, _f, _f, _i, _f, _p]),
    "r3dg_relight_pack_features": (_i, [_p, _i] + [_p] * 7),
    "r3dg_relight_compose": (_i, [_p, _i, _i, _f, _f, _f, _f, _p, _p, _p, _i, _i] + [_p] * 7),
    "r3dg_densify_accumulate": (_i, [_p, _i] + [_p] * 10),
    "r3dg_densify_temp_bytes": (C.c_size_t, [_i]),
    "r3dg_densify_plan": (_i, [_p, _i] + [_p] * 12),
    "r3dg_densify_gather": (_i, [_p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _f]),
    "r3dg_reset_opacity": (_i, [_p, _i, _p, _p, _p]),
    "r3dg_knn_temp_bytes": (C.c_size_t, [_i]),
    "r3dg_knn_dist2": (_i, [_p, _i, _p, _p, _p]),
    "r3dg_bvh_build_temp_bytes": (C.c_size_t, [_i]),
    "r3dg_bvh_build": (_i, [_p, _i, _p, _p, _p, _p]),
    "r3dg_bvh_trace_opacity": (_i, [_p, C.c_int64, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "r3dg_bvh_trace_records_bytes": (C.c_size_t, [_i]),
    "r3dg_bvh_pack_traversal": (_i, [_p, _i] + [_p] * 7),
    "r3dg_bvh_trace_opacity_packed": (_i, [_p, C.c_int64, _i] + [_p] * 6),
    "r3dg_bvh_trace_visits": (_i, [_p, _i, _p, C.POINTER(C.c_uint64)]),
    "r3dg_profile_enable": (_i, [_i]),
    "r3dg_profile_pause": (_i, [_i]),
    "r3dg_profile_num_stages": (_i, []),
    "r3dg_profile_stage_name": (C.c_char_p, [_i]),
    "r3dg_profile_read": (_i, [C.POINTER(C.c_double), C.POINTER(_i)]),
}

# entry points added by later translation units register themselves here (shading, bvh, knn)
EXTRA_SIGNATURES = {}


def lib():
    """Load the HIP library (once).  Raises RuntimeError with build instructions if it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libr3dg_hip.so not found at %s -- build it with `python -m relightable3dgaussian_amd.build` "
                "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        # PyTorch-ROCm ships its own HIP runtime; it must be in the process BEFORE this library is loaded so that both
        # resolve to ONE runtime instance (loaded the other way round, the library's kernels see "no ROCm-capable
        # device" once torch has initialised its copy)
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        sigs = dict(_SIGNATURES)
        sigs.update(EXTRA_SIGNATURES)
        for name, (res, args) in sigs.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


# enum r3dg_option (include/r3dg_hip.h); tests/test_oracle_cpu.py checks the numbering against the header
OPTIONS = ("TILE_ORDER", "CULL", "TILE_BINNING", "BINNING_BLOCK_K", "STAGE_SH_ROWS", "SHADE_FWD_BLOCKS_PER_CU", "TRACE_FORMULATION",
           "TRACE_REFILL", "TRACE_NODE_WEIGHT", "TRACE_LEAF_WEIGHT", "RESERVE_CUS", "TRACE_COUNT_VISITS", "BWD_LEAN")


def set_option(name, value):
    """r3dg_set_option by name (experiments / tests): set_option("CULL", 0)."""
    check(lib().r3dg_set_option(OPTIONS.index(name), int(value)), "set_option(%s)" % name)


_live_lock = threading.Lock()
_live_entries = {}          # context handle -> number of `with` blocks (any thread) currently inside it


class OptionContext:
    """Tuning options that belong to ONE object (include/r3dg_hip.h "option contexts"): `ctx.set("RESERVE_CUS", 8)`, then
    `with ctx:` around the object's library calls -- inside, launches of the calling thread see the context's values where it
    sets them and the process defaults elsewhere; the previous context is restored on exit (contexts nest)."""

    def __init__(self, **options):
        self._h = lib().r3dg_context_create()
        if self._h is None:                      # (NULL)
            raise RuntimeError("r3dg_context_create failed")
        # the restore stack is PER THREAD, like the library's "current context": two threads inside the same object's context
        # (a worker polling while the main thread is in an iteration) must not pop each other's saved handles
        self._tls = threading.local()
        for k, v in options.items():
            self.set(k, v)

    def set(self, name, value):
        check(lib().r3dg_context_set_option(self._h, OPTIONS.index(name), int(value)), "context_set_option(%s)" % name)

    def __enter__(self):
        prev = _p()
        check(lib().r3dg_context_make_current(self._h, C.byref(prev)), "context_make_current")
        stack = getattr(self._tls, "prev", None)
        if stack is None:
            stack = self._tls.prev = []
        stack.append(prev.value)
        with _live_lock:
            _live_entries[self._h] = _live_entries.get(self._h, 0) + 1
        return self

    def __exit__(self, *exc):
        check(lib().r3dg_context_make_current(self._tls.prev.pop(), None), "context_make_current")
        with _live_lock:
            n = _live_entries.get(self._h, 0) - 1
            if n > 0:
                _live_entries[self._h] = n
            else:
                _live_entries.pop(self._h, None)
        return False

    def __del__(self):
        # (a context that some thread is still inside -- current there, or saved as another context's "previous" -- is leaked
        # rather than destroyed: the library would dereference the freed handle at that thread's next launch)
        try:
            if self._h:
                with _live_lock:
                    busy = _live_entries.get(self._h, 0) > 0
                if not busy:
                    lib().r3dg_context_destroy(self._h)
                self._h = None
        except Exception:
            pass


def get_option(name):
    v = C.c_int(0)
    check(lib().r3dg_get_option(OPTIONS.index(name), C.byref(v)), "get_option(%s)" % name)
    return v.value


def check(status, what):
    if status != 0:
        msg = lib().r3dg_last_error()
        raise RuntimeError("%s failed (%d): %s" % (what, status, msg.decode() if msg else "?"))


def ptr(t):
    """Device pointer of a torch tensor, or None for an absent optional (numel()==0: the reference passes empty
    CPU tensors for absent optionals, gaussian_renderer/r3dg_rasterization.py:235-245)."""
    if t is None or t.numel() == 0:
        return None
    return t.data_ptr()


def current_stream():
    """The raw HIP stream torch launches on right now (of the current device).  torch.cuda.current_stream() builds a Stream
    object through half a dozen Python layers (10 us; the fused iteration asks 16 times): the C accessor it ends in is used
    directly when this torch has it."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return raw(torch._C._cuda_getDevice())
    return torch.cuda.current_stream().cuda_stream


def stream_wait(waiter, signaller):
    """`waiter` (a torch.cuda.Stream) waits for everything queued on `signaller` so far: torch's waiter.wait_stream(signaller)
    through the library's pooled events (r3dg_stream_wait_stream)."""
    check(lib().r3dg_stream_wait_stream(waiter.cuda_stream, signaller.cuda_stream), "stream_wait_stream")


def shader_clock_ghz(device="cuda", iters=4000):
    """The shader clock under VALU load, measured on the device itself (r3dg_clock_probe): cycles of the shader-clock counter per
    tick of the constant-rate wall clock, summed over a device-filling grid of FMA-only waves.  -> (GHz, waves that reported)."""
    import torch
    out = torch.zeros(3, dtype=torch.int64, device=device)
    sink = torch.zeros(1, dtype=torch.float32, device=device)
    khz = C.c_int(0)
    with torch.cuda.device(out.device):
        check(lib().r3dg_clock_probe(current_stream(), int(iters), out.data_ptr(), sink.data_ptr(), C.byref(khz)), "clock_probe")
        torch.cuda.synchronize()
    cyc, ticks, waves = (int(v) for v in out.tolist())
    return (cyc / max(ticks, 1)) * khz.value * 1e-6, waves


def profile_read():
    """{stage name: (total ms, launches)} since the last r3dg_profile_enable(1)."""
    L = lib()
    n = L.r3dg_profile_num_stages()
    ms = (C.c_double * n)()
    cnt = (C.c_int * n)()
    check(L.r3dg_profile_read(ms, cnt), "profile_read")
    return {L.r3dg_profile_stage_name(i).decode(): (ms[i], cnt[i]) for i in range(n)}
