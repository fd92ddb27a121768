"""Host side of the fused shading integral (C ABI r3dg_shade_forward / r3dg_shade_backward).

`rendering_equation(...)` has the signature and return convention of the reference's live
`gaussian_renderer.neilf.rendering_equation` (neilf.py:339-371), so `neilf.rendering_equation = shading_ops.rendering_equation`
swaps the op in without editing the reference file.  The per-sample [P,K,*] tensors of `extra_results` are only ever
consumed through `.mean(-2)` (neilf.py:119-130); they are returned already reduced with a singleton sample axis
([P,1,C]) so that `.mean(-2)` and the eval-time `torch.cat(..., dim=0)` keep working on unchanged caller code.
"""
import ctypes

import torch

from . import _lib

NOUT = 19


def _c(t):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise RuntimeError("shading ops expect float32 CUDA(HIP) tensors")
    return t.contiguous()


def build_taps(incident_dirs, He, We, env_transform=None, radiance_of=None):
    """Lat-long lookup of every cached direction (texel corner + two bilinear weights; int32 [..., 3] holding the 12-byte
    records of r3dg_shade_build_taps) for an environment texture of He x We texels and this `env_transform`.  Valid as
    long as `incident_dirs` is (the directions are frozen between visibility updates, gaussian_model.py:312-342).
    `radiance_of` = a FIXED env[He,We,3] (not trained: relighting): the records hold the sampled radiance instead; pass them
    to shade_forward with taps_are_radiance=True."""
    L = _lib.lib()
    d = _c(incident_dirs)
    tr = _c(env_transform) if env_transform is not None else None
    taps = torch.empty(d.shape, dtype=torch.int32, device=d.device)
    with torch.cuda.device(d.device):
        env = _c(radiance_of) if radiance_of is not None else None
        st = L.r3dg_shade_build_taps(_lib.current_stream(), d.numel() // 3, d.data_ptr(),
                                     tr.data_ptr() if tr is not None else None, int(He), int(We),
                                     env.data_ptr() if env is not None else None, taps.data_ptr())
    _lib.check(st, "shade_build_taps")
    return taps


def shade_forward(base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs, incident_areas,
                  env_transform=None, taps=None, train_outputs=False, out=None, uniform_area=None, taps_are_radiance=False):
    """-> out[P,19] = pbr3 diffuse3 specular3 lights3 local3 global3 vis1 (see include/r3dg_hip.h).
    `taps`: build_taps(incident_dirs, He, We, env_transform) of THESE directions (skips the per-sample acos/atan2);
    `train_outputs`: only columns 0..5 and 18 are written (what the training feature row reads);
    `uniform_area`: every sample's area (then `incident_areas` is not read: fibonacci_sphere_sampling gives 2*pi to all)."""
    L = _lib.lib()
    P, K = incident_dirs.shape[0], incident_dirs.shape[1]
    M = incidents.shape[1]
    He, We = env.shape[-3], env.shape[-2]
    t = [_c(x) for x in (base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs,
                         incident_areas)]
    tr = _c(env_transform) if env_transform is not None else None
    if taps is not None and (taps.dtype != torch.int32 or taps.numel() != 3 * P * K or not taps.is_contiguous()):
        raise RuntimeError("taps must be the contiguous int32 [P,K,3] tensor of build_taps for these directions")
    if out is None:
        out = torch.empty((P, NOUT), dtype=torch.float32, device=base_color.device)
    with torch.cuda.device(base_color.device):
        st = L.r3dg_shade_forward_cached(_lib.current_stream(), P, K, M, t[0].data_ptr(), t[1].data_ptr(),
                                         t[2].data_ptr(), t[3].data_ptr(), t[4].data_ptr(), t[5].data_ptr(), He, We,
                                         tr.data_ptr() if tr is not None else None, t[6].data_ptr(), t[7].data_ptr(),
                                         None if uniform_area is not None else t[8].data_ptr(),
                                         float(uniform_area or 0.0), taps.data_ptr() if taps is not None else None,
                                         (1 if train_outputs else 0) | (2 if taps_are_radiance else 0), out.data_ptr())
    _lib.check(st, "shade_forward")
    return out


def shade_backward(base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs, incident_areas,
                   dL_dpbr, dL_ddiffuse_light, env_transform=None, out_incidents=None, taps=None, out_env=None,
                   block_absmax=None):
    """`out_incidents`: optional preallocated contiguous [P,M,3] buffer for dL_dincidents (fully overwritten);
    `taps`: build_taps(incident_dirs, He, We, env_transform) (lookup records, not radiance);
    `out_env`: optional ZEROED float32 buffer shaped like `env` that receives dL_denv (the kernel accumulates into it);
    `block_absmax`: optional float32 vector whose maximum is max(|dL_dpbr|, |dL_ddiffuse_light|) (+inf if not finite), as
    r3dg_stage2_unpack_gradients writes it -- saves the reduction pass in front of the kernel."""
    L = _lib.lib()
    P, K = incident_dirs.shape[0], incident_dirs.shape[1]
    M = incidents.shape[1]
    He, We = env.shape[-3], env.shape[-2]
    t = [_c(x) for x in (base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs,
                         incident_areas, dL_dpbr, dL_ddiffuse_light)]
    tr = _c(env_transform) if env_transform is not None else None
    dev = base_color.device
    d_base = torch.empty((P, 3), dtype=torch.float32, device=dev)
    d_rough = torch.empty((P, 1), dtype=torch.float32, device=dev)
    d_view = torch.empty((P, 3), dtype=torch.float32, device=dev)
    if out_incidents is not None:
        if tuple(out_incidents.shape) != (P, M, 3) or not out_incidents.is_contiguous():
            raise RuntimeError("out_incidents must be a contiguous [P,M,3] tensor")
        d_inc = out_incidents
    else:
        d_inc = torch.empty((P, M, 3), dtype=torch.float32, device=dev)
    if out_env is not None:
        if out_env.shape != t[5].shape or not out_env.is_contiguous() or out_env.dtype != torch.float32:
            raise RuntimeError("out_env must be a contiguous float32 tensor shaped like env")
        d_env = out_env
    else:
        d_env = torch.zeros_like(t[5])
    if block_absmax is not None and (block_absmax.dtype != torch.float32 or not block_absmax.is_contiguous()):
        raise RuntimeError("block_absmax must be a contiguous float32 tensor")
    with torch.cuda.device(dev):
        st = L.r3dg_shade_backward_cached(_lib.current_stream(), P, K, M, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                   t[3].data_ptr(), t[4].data_ptr(), t[5].data_ptr(), He, We,
                                   tr.data_ptr() if tr is not None else None, t[6].data_ptr(), t[7].data_ptr(),
                                   t[8].data_ptr(), taps.data_ptr() if taps is not None else None,
                                   t[9].data_ptr(), t[10].data_ptr(), d_base.data_ptr(),
                                   d_rough.data_ptr(), d_view.data_ptr(), d_inc.data_ptr(), d_env.data_ptr(),
                                   block_absmax.data_ptr() if block_absmax is not None else None,
                                   block_absmax.numel() if block_absmax is not None else 0)
    _lib.check(st, "shade_backward")
    return d_base, d_rough, d_view, d_inc, d_env


# ---- relight caches (fixed / turning light over static Gaussians): thin wrappers of the C ABI, used by relight.RelightRenderer
# and driven directly by tests/test_shading_gpu.py::test_relight_kernels_match_oracle -------------------------------------------
def build_transport(normals, incidents, visibility, incident_dirs, incident_areas, uniform_area, radiance_inout, consts=None):
    """r3dg_shade_build_transport: `radiance_inout` (build_taps(..., radiance_of=env) of THESE directions) is rewritten in place
    with the per-sample transport; -> consts [P,16] (per Gaussian diffuse_light, mean lights, mean visibility)."""
    P, K, M = incident_dirs.shape[0], incident_dirs.shape[1], incidents.shape[1]
    if radiance_inout.numel() != 3 * P * K or not radiance_inout.is_contiguous():
        raise RuntimeError("radiance_inout must be the contiguous [P,K,3] record tensor of build_taps(..., radiance_of=env)")
    if consts is None:
        consts = torch.empty(P, 16, dtype=torch.float32, device=normals.device)
    with torch.cuda.device(normals.device):
        _lib.check(_lib.lib().r3dg_shade_build_transport(
            _lib.current_stream(), P, K, M, _c(normals).data_ptr(), _c(incidents).data_ptr(), _c(visibility).data_ptr(),
            _c(incident_dirs).data_ptr(), None if uniform_area is not None else _c(incident_areas).data_ptr(),
            float(uniform_area or 0.0), radiance_inout.data_ptr(), consts.data_ptr()), "shade_build_transport")
    return consts


def shade_forward_transport(base_color, roughness, normals, viewdirs, transport, consts, zsamples, incident_dirs=None, out=None):
    """r3dg_shade_forward_transport: the GGX lobe of every sample against the cached transport -> out[P,19].
    `incident_dirs` None: the directions are regenerated from the normals and `zsamples` [K,3]."""
    P, K = consts.shape[0], zsamples.shape[0]
    if out is None:
        out = torch.empty((P, NOUT), dtype=torch.float32, device=consts.device)
    with torch.cuda.device(consts.device):
        _lib.check(_lib.lib().r3dg_shade_forward_transport(
            _lib.current_stream(), P, K, _c(base_color).data_ptr(), _c(roughness).data_ptr(), _c(normals).data_ptr(),
            _c(viewdirs).data_ptr(), transport.data_ptr(), consts.data_ptr(), zsamples.data_ptr(),
            None if incident_dirs is None else _c(incident_dirs).data_ptr(), out.data_ptr()), "shade_forward_transport")
    return out


SPLIT_MAX_ENV = 4095        # r3dg_shade_env_footprints / r3dg_shade_forward_split pack a texel coordinate into 12 bits


def split_supported(K, M, He, We, uniform_area):
    """What the split-transport kernels implement: the reference's relighting configuration (16 incident-light coefficients, the
    Fibonacci ray set with its uniform area), K a multiple of 4 and a map of at most 4095 x 4095 texels."""
    return M == 16 and uniform_area is not None and K % 4 == 0 and 0 < He <= SPLIT_MAX_ENV and 0 < We <= SPLIT_MAX_ENV


def env_footprints(env):
    """r3dg_shade_env_footprints: the map [He,We,3] as (He+1)(We+1) 48-byte bilinear footprints."""
    He, We = env.shape[0], env.shape[1]
    L = _lib.lib()
    env4 = torch.empty(int(L.r3dg_shade_env_footprints_bytes(He, We)) // 4, dtype=torch.float32, device=env.device)
    with torch.cuda.device(env.device):
        _lib.check(L.r3dg_shade_env_footprints(_lib.current_stream(), He, We, _c(env).data_ptr(), env4.data_ptr()),
                   "shade_env_footprints")
    return env4


def build_split(perm, normals, incidents, visibility, incident_dirs, zsamples, uniform_area):
    """r3dg_shade_build_split: the light-independent half of the transport, sample-major, Gaussians in the order `perm` lists
    them -> dict(lt [K,P,4], vis_t [K/4,P,4], consts [P,4]).  `incident_dirs` None: regenerated from normals and zsamples."""
    P, K = normals.shape[0], zsamples.shape[0]
    f = dict(dtype=torch.float32, device=normals.device)
    lt, vis_t, consts = torch.empty(K, P, 4, **f), torch.empty(K // 4, P, 4, **f), torch.empty(P, 4, **f)
    with torch.cuda.device(normals.device):
        _lib.check(_lib.lib().r3dg_shade_build_split(
            _lib.current_stream(), P, K, perm.data_ptr(), _c(normals).data_ptr(), _c(incidents).data_ptr(),
            _c(visibility).data_ptr(), None if incident_dirs is None else _c(incident_dirs).data_ptr(), zsamples.data_ptr(),
            float(uniform_area), lt.data_ptr(), vis_t.data_ptr(), consts.data_ptr()), "shade_build_split")
    return dict(perm=perm, lt=lt, vis_t=vis_t, consts=consts, zsamples=zsamples)


def shade_forward_split(split, base_color, roughness, normals, viewdirs, env_transform, env4, He, We, out=None):
    """r3dg_shade_forward_split: per frame the lat-long lookup of the ROTATED direction (`env_transform` [3,3] on the device or
    None; must be a rotation -- the lobe is evaluated in the light's frame), the transport and the GGX lobe -> out[P,19]."""
    P, K = split["consts"].shape[0], split["zsamples"].shape[0]
    if out is None:
        out = torch.empty((P, NOUT), dtype=torch.float32, device=env4.device)
    with torch.cuda.device(env4.device):
        _lib.check(_lib.lib().r3dg_shade_forward_split(
            _lib.current_stream(), P, K, split["perm"].data_ptr(), _c(base_color).data_ptr(), _c(roughness).data_ptr(),
            _c(normals).data_ptr(), _c(viewdirs).data_ptr(), split["lt"].data_ptr(), split["vis_t"].data_ptr(),
            split["consts"].data_ptr(), split["zsamples"].data_ptr(), _lib.ptr(env_transform), env4.data_ptr(), int(He), int(We),
            out.data_ptr()), "shade_forward_split")
    return out


class FixedRaySet:
    """State of the fixed-ray-set shading kernels (include/r3dg_hip.h "fixed ray set", csrc/shading_frs.hpp) for ONE
    visibility update: the normals the cached directions were generated from (12 bytes per Gaussian -- the kernels read NO
    per-sample direction), the z set and its Y_i(z_k) tables, which Gaussians take the rotated path, the 8-byte lookup
    records of the current texture size, scratch for the rotated coefficients.  `forward` / `backward` compute what
    shade_forward(..., train_outputs=True) / shade_backward(...) compute for caches that ARE
    sampling.fibonacci_sphere_sampling(ray_normals, K); `try_build` checks that and returns None otherwise."""

    def __init__(self, ray_normals, K):
        from . import sampling
        L = _lib.lib()
        self.ray_normals = _c(ray_normals).clone()
        self.P, self.K = self.ray_normals.shape[0], int(K)
        dev = self.ray_normals.device
        self.zsamples = sampling.fibonacci_z_samples(self.K, dev)[0].t().contiguous()            # [K,3]
        self.tables = torch.empty(int(L.r3dg_shade_frs_tables_bytes(self.K)) // 4, dtype=torch.float32, device=dev)
        self.valid = torch.zeros(max(self.P, 1), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.r3dg_shade_frs_build_tables(_lib.current_stream(), self.K, self.zsamples.data_ptr(),
                                                     self.tables.data_ptr()), "shade_frs_build_tables")
            _lib.check(L.r3dg_shade_frs_classify(_lib.current_stream(), self.P, self.ray_normals.data_ptr(),
                                                 self.valid.data_ptr()), "shade_frs_classify")
        self.invalid_list = torch.nonzero(self.valid[:self.P] == 0).to(torch.int32).reshape(-1).contiguous()
        self.n_invalid = int(self.invalid_list.numel())                   # (one read-back per visibility update)
        self.cprime = torch.empty(self.P, 48, dtype=torch.float32, device=dev)
        self.dcprime = torch.empty(self.P, 48, dtype=torch.float32, device=dev)
        self._taps, self._taps_size = None, None

    @staticmethod
    def supported(K, M, He, We):
        return bool(_lib.lib().r3dg_shade_frs_supported(int(K), int(M), int(He), int(We)))

    last_mismatch = None         # max |cached - regenerated| direction component of the latest try_build (diagnostics)

    @classmethod
    def try_build(cls, ray_normals, incident_dirs, tol=5e-5, chunk=1 << 16):
        """-> FixedRaySet, or None when `incident_dirs` [P,K,3] is not the Fibonacci set of `ray_normals` (checked once per
        visibility update; chunked so that the transient stays small).  `tol` only has to tell THIS ray set from any other one:
        a cache generated by the same formulas on another device (the CPU-generated reference fixtures of
        tests/test_reference_pipeline_gpu.py: sin / cos of angles up to 60 rad through another libm) sits ~2e-5 away, a
        different normal or sample count sits at O(1)."""
        from . import sampling
        P, K = incident_dirs.shape[0], incident_dirs.shape[1]
        if ray_normals is None or ray_normals.shape[0] != P or K < 4 or K % 4 != 0:
            return None
        worst = torch.zeros((), dtype=torch.float32, device=incident_dirs.device)
        for o in range(0, P, chunk):
            want, _ = sampling.fibonacci_sphere_sampling(ray_normals[o:o + chunk], K)
            worst = torch.maximum(worst, (want - incident_dirs[o:o + chunk]).abs().max())
        cls.last_mismatch = float(worst)
        if not cls.last_mismatch <= tol:
            return None
        return cls(ray_normals, K)

    def taps(self, He, We):
        """The 8-byte lat-long lookup records [P,K,2] (int32) of this ray set for a He x We texture
        (r3dg_shade_frs_build_taps: regenerated from the ray normals, no direction array is read); kept until the size changes."""
        if self._taps is None or self._taps_size != (int(He), int(We)):
            taps = torch.empty((self.P, self.K, 2), dtype=torch.int32, device=self.ray_normals.device)
            with torch.cuda.device(taps.device):
                _lib.check(_lib.lib().r3dg_shade_frs_build_taps(
                    _lib.current_stream(), self.P, self.K, self.ray_normals.data_ptr(), self.zsamples.data_ptr(), int(He), int(We),
                    taps.data_ptr()), "shade_frs_build_taps")
            self._taps, self._taps_size = taps, (int(He), int(We))
        return self._taps

    def _common(self, base_color, roughness, normals, viewdirs, incidents, env, visibility, uniform_area):
        t = [_c(x) for x in (base_color, roughness, normals, viewdirs, incidents, env, visibility)]
        if tuple(incidents.shape) != (self.P, 16, 3) or visibility.numel() != self.P * self.K:
            raise RuntimeError("FixedRaySet: needs [P,16,3] incident-light coefficients and the [P,K] visibility it was built for")
        He, We = env.shape[-3], env.shape[-2]
        taps = self.taps(He, We)
        head = [self.P, self.K] + [x.data_ptr() for x in t[:6]] + [He, We, t[6].data_ptr(), float(uniform_area or 0.0),
                                                                   taps.data_ptr(), self.ray_normals.data_ptr(),
                                                                   self.zsamples.data_ptr(), self.tables.data_ptr(),
                                                                   self.valid.data_ptr(),
                                                                   self.invalid_list.data_ptr() if self.n_invalid else None,
                                                                   self.n_invalid]
        return head, t

    def rotate(self, incidents):
        """The first step of `forward` on its own (on the current stream): the rotated coefficients of `incidents` [P,16,3].
        A caller that queues it early passes rotated=True to `forward`."""
        inc = _c(incidents)
        if inc.shape != (self.P, 16, 3):
            raise RuntimeError("FixedRaySet: needs [P,16,3] incident-light coefficients")
        with torch.cuda.device(inc.device):
            st = _lib.lib().r3dg_shade_frs_rotate(_lib.current_stream(), self.P, inc.data_ptr(), self.ray_normals.data_ptr(),
                                                  self.cprime.data_ptr())
        _lib.check(st, "shade_frs_rotate")

    def forward(self, base_color, roughness, normals, viewdirs, incidents, env, visibility, out, uniform_area=None,
                leave_room=False, listed_stream=None, rotated=False, feature_rows=None):
        """Writes columns 0..5 and 18 of out [P,19] (pbr, diffuse_light, mean visibility); keeps the rotated coefficients
        for `backward`.  `uniform_area`: the area of every sample (None = 2 pi, what fibonacci_sphere_sampling assigns).
        `listed_stream` (a torch.cuda.Stream): the kernel on the Gaussians off the rotated path runs there, beside the main
        kernel; the caller waits for that stream before reading `out`.  `feature_rows` ([P,16], optional): the same three
        results also go straight into columns 2..4, 12..14, 15 of the rasterizer's feature rows (neilf.py:115-122)."""
        if feature_rows is not None and (tuple(feature_rows.shape) != (self.P, 16) or not feature_rows.is_contiguous() or
                                         feature_rows.dtype != torch.float32):
            raise RuntimeError("feature_rows must be a contiguous float32 [P,16] tensor")
        head, _keep = self._common(base_color, roughness, normals, viewdirs, incidents, env, visibility, uniform_area)
        with torch.cuda.device(base_color.device):
            st = _lib.lib().r3dg_shade_frs_forward(_lib.current_stream(), *head, self.cprime.data_ptr(),
                                                   1 | (4 if leave_room else 0) | (8 if rotated else 0), out.data_ptr(),
                                                   listed_stream.cuda_stream if listed_stream is not None else None,
                                                   _lib.ptr(feature_rows))
        _lib.check(st, "shade_frs_forward")
        return out

    def backward(self, base_color, roughness, normals, viewdirs, incidents, env, visibility, dL_dpbr, dL_ddiffuse_light,
                 uniform_area=None, out_incidents=None, out_env=None, block_absmax=None, rotate_stream=None,
                 rotation_back=True):
        """-> (dL_dbase_color, dL_droughness, dL_dviewdirs, dL_dincidents, dL_denv) as shade_backward; `forward` must have run
        on the same parameters (it left the rotated coefficients).  `rotate_stream` (a torch.cuda.Stream): the rotation of the
        coefficient gradient back to dL_dincidents runs there, beside whatever the caller queues next on the current stream; the
        caller waits for that stream before reading dL_dincidents.  `rotation_back=False`: the coefficient gradient stays in the
        rotated frame (`self.dcprime`) and dL_dincidents is NOT written for the Gaussians on the rotated path -- the caller follows
        with `incident_chain`."""
        head, _keep = self._common(base_color, roughness, normals, viewdirs, incidents, env, visibility, uniform_area)
        dev = base_color.device
        P = self.P
        d_base = torch.empty((P, 3), dtype=torch.float32, device=dev)
        d_rough = torch.empty((P, 1), dtype=torch.float32, device=dev)
        d_view = torch.empty((P, 3), dtype=torch.float32, device=dev)
        d_inc = out_incidents if out_incidents is not None else torch.empty((P, 16, 3), dtype=torch.float32, device=dev)
        d_env = out_env if out_env is not None else torch.zeros_like(env)
        gp, gd = _c(dL_dpbr), _c(dL_ddiffuse_light)
        with torch.cuda.device(dev):
            st = _lib.lib().r3dg_shade_frs_backward(
                _lib.current_stream(), *head, self.cprime.data_ptr(), self.dcprime.data_ptr(), gp.data_ptr(), gd.data_ptr(),
                d_base.data_ptr(), d_rough.data_ptr(), d_view.data_ptr(), d_inc.data_ptr(), d_env.data_ptr(),
                block_absmax.data_ptr() if block_absmax is not None else None,
                block_absmax.numel() if block_absmax is not None else 0,
                (ctypes.c_void_p(-1) if not rotation_back else
                 (rotate_stream.cuda_stream if rotate_stream is not None else None)))
        _lib.check(st, "shade_frs_backward")
        return d_base, d_rough, d_view, d_inc, d_env

    def incident_chain(self, incidents, dL_dincidents, exp_avg, exp_avg_sq, lr, lr_tail, betas, eps, step, grad_scale=1.0,
                       skip_flag=None, listed_in_dcprime=False):
        """r3dg_shade_frs_incident_chain on the current stream, behind backward(rotation_back=False): the gradient rotated back
        into `dL_dincidents`, the Adam step of the incident-light group (`incidents`, its two moment tensors; FusedAdam's
        arithmetic), the NEW coefficients rotated into `self.cprime` for the next forward.  `listed_in_dcprime`: backward() was
        given `out_incidents=self.dcprime_rows()` -- the world-frame rows of the Gaussians off the rotated path sit in `dcprime`
        too (ONE buffer to all-reduce under data parallelism)."""
        for t in (incidents, dL_dincidents, exp_avg, exp_avg_sq):
            if tuple(t.shape) != (self.P, 16, 3) or t.dtype != torch.float32 or not t.is_contiguous():
                raise RuntimeError("FixedRaySet.incident_chain: needs contiguous float32 [P,16,3] tensors")
        with torch.cuda.device(incidents.device):
            st = _lib.lib().r3dg_shade_frs_incident_chain(
                _lib.current_stream(), self.P, self.ray_normals.data_ptr(), self.valid.data_ptr(), self.dcprime.data_ptr(),
                dL_dincidents.data_ptr(), incidents.data_ptr(), exp_avg.data_ptr(), exp_avg_sq.data_ptr(), self.cprime.data_ptr(),
                float(lr), float(lr_tail), float(betas[0]), float(betas[1]), float(eps), int(step), float(grad_scale),
                skip_flag.data_ptr() if skip_flag is not None else None, 1 if listed_in_dcprime else 0)
        _lib.check(st, "shade_frs_incident_chain")

    def dcprime_rows(self):
        """`dcprime` as the [P,16,3] tensor backward(out_incidents=...) takes."""
        return self.dcprime.view(self.P, 16, 3)


class _Shade(torch.autograd.Function):
    @staticmethod
    def forward(ctx, base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs,
                incident_areas, env_transform):
        out = shade_forward(base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs,
                            incident_areas, env_transform)
        ctx.save_for_backward(base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs,
                              incident_areas)
        ctx.env_transform = env_transform
        pbr, diffuse, rest = out[:, 0:3], out[:, 3:6], out[:, 6:]
        ctx.mark_non_differentiable(rest)
        return pbr, diffuse, rest

    @staticmethod
    def backward(ctx, g_pbr, g_diffuse, _g_rest):
        base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs, incident_areas = \
            ctx.saved_tensors
        z = torch.zeros_like(base_color)
        d_base, d_rough, d_view, d_inc, d_env = shade_backward(
            base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs, incident_areas,
            g_pbr if g_pbr is not None else z, g_diffuse if g_diffuse is not None else z, ctx.env_transform)
        return d_base, d_rough.view_as(roughness), None, d_view, d_inc, d_env.view_as(env), None, None, None, None


def shade(base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs, incident_areas,
          env_transform=None):
    """Differentiable fused integral -> (pbr[P,3], diffuse_light[P,3], rest[P,13] = specular3 lights3 local3 global3 vis1).
    Gradients flow to base_color, roughness, viewdirs, incidents and env -- what the reference's callers differentiate
    (normal.detach(), cached samples: neilf.py:115-118); asking for any other one is an error, not a silent zero."""
    for name, t in (("normals", normals), ("visibility", visibility), ("incident_dirs", incident_dirs),
                    ("incident_areas", incident_areas), ("env_transform", env_transform)):
        if t is not None and t.requires_grad:
            raise RuntimeError("shading_ops.shade: no gradient is implemented for `%s` (the reference passes it detached); "
                               "detach it" % name)
    return _Shade.apply(base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs,
                        incident_areas, env_transform)


def _light_to_env(light):
    """DirectLightMap (learnable, softplus) or EnvLight (fixed HDR map + optional rotation) -> (env[He,We,3], transform)."""
    if hasattr(light, "envmap"):
        return light.envmap, getattr(light, "transform", None)
    env = light.get_env
    return (env[0] if env.dim() == 4 else env), None


def rendering_equation(base_color, roughness, normals, viewdirs, incidents, direct_light_env_light=None,
                       visibility_precompute=None, incident_dirs_precompute=None, incident_areas_precompute=None):
    """Drop-in for gaussian_renderer.neilf.rendering_equation (neilf.py:339-371)."""
    env, tr = _light_to_env(direct_light_env_light)
    pbr, diffuse_light, rest = shade(base_color, roughness, normals, viewdirs, incidents, env, visibility_precompute,
                                     incident_dirs_precompute, incident_areas_precompute, tr)
    extra_results = {
        "incident_dirs": incident_dirs_precompute,
        "incident_lights": rest[:, None, 3:6],
        "local_incident_lights": rest[:, None, 6:9],
        "global_incident_lights": rest[:, None, 9:12],
        "incident_visibility": rest[:, None, 12:13],
        "diffuse_light": diffuse_light,
        "specular": rest[:, 0:3],
    }
    return pbr, extra_results


# ---------------------------------------------------------------------------------------------------------------
# The reference's render_equation.h contract model (never bound in the reference snapshot; names/signatures follow
# RenderEquationForwardCUDA / RenderEquationForwardCUDA_complex / RenderEquationBackwardCUDA, render_equation.h:7-46)
# ---------------------------------------------------------------------------------------------------------------
def _re_common(base_color, incidents_shs, direct_shs, visibility_shs):
    P = base_color.size(0)
    return P, incidents_shs.size(1), direct_shs.size(1), visibility_shs.size(1)


def render_equation_forward(base_color, roughness, metallic, normals, viewdirs, incidents_shs, direct_shs,
                            visibility_shs, sample_num, is_training, debug=False):
    """-> (pbr[P,3], incident_dirs[P,K,3], diffuse_light[P,3])"""
    L = _lib.lib()
    P, Si, Sd, Sv = _re_common(base_color, incidents_shs, direct_shs, visibility_shs)
    dev = base_color.device
    t = [_c(x) for x in (base_color, roughness, metallic, normals, viewdirs, incidents_shs, direct_shs, visibility_shs)]
    pbr = torch.zeros((P, 3), dtype=torch.float32, device=dev)
    incident_dirs = torch.zeros((P, sample_num, 3), dtype=torch.float32, device=dev)
    diffuse_light = torch.zeros((P, 3), dtype=torch.float32, device=dev)
    rand_float = torch.rand((P, sample_num, 1), dtype=torch.float32, device=dev)   # drawn even when unused, like the reference
    with torch.cuda.device(dev):
        st = L.r3dg_render_equation_forward(_lib.current_stream(), P, Si, Sd, Sv, *[x.data_ptr() for x in t],
                                            int(sample_num), rand_float.data_ptr() if is_training else None,
                                            incident_dirs.data_ptr(), pbr.data_ptr(), diffuse_light.data_ptr())
    _lib.check(st, "render_equation_forward")
    render_equation_forward.last_rand = rand_float
    return pbr, incident_dirs, diffuse_light


def render_equation_forward_complex(base_color, roughness, metallic, normals, viewdirs, incidents_shs, direct_shs,
                                    visibility_shs, sample_num):
    """-> (pbr, incident_dirs, incident_lights, local_incident_lights, global_incident_lights, incident_visibility,
    diffuse_light, local_diffuse_light, accum, rgb_d, rgb_s)"""
    L = _lib.lib()
    P, Si, Sd, Sv = _re_common(base_color, incidents_shs, direct_shs, visibility_shs)
    dev = base_color.device
    K = int(sample_num)
    t = [_c(x) for x in (base_color, roughness, metallic, normals, viewdirs, incidents_shs, direct_shs, visibility_shs)]

    def z(*shape):
        return torch.zeros(shape, dtype=torch.float32, device=dev)
    pbr, incident_dirs, lights, local, glob = z(P, 3), z(P, K, 3), z(P, K, 3), z(P, K, 3), z(P, K, 3)
    vis, diffuse, local_diffuse, accum, rgb_d, rgb_s = z(P, K, 1), z(P, 3), z(P, 3), z(P, 1), z(P, 3), z(P, 3)
    with torch.cuda.device(dev):
        st = L.r3dg_render_equation_forward_complex(
            _lib.current_stream(), P, Si, Sd, Sv, *[x.data_ptr() for x in t], K, incident_dirs.data_ptr(),
            pbr.data_ptr(), lights.data_ptr(), local.data_ptr(), glob.data_ptr(), vis.data_ptr(), diffuse.data_ptr(),
            local_diffuse.data_ptr(), accum.data_ptr(), rgb_d.data_ptr(), rgb_s.data_ptr())
    _lib.check(st, "render_equation_forward_complex")
    return pbr, incident_dirs, lights, local, glob, vis, diffuse, local_diffuse, accum, rgb_d, rgb_s


def render_equation_backward(base_color, roughness, metallic, normals, viewdirs, incidents, direct_shs, visibility_shs,
                             sample_num, incident_dirs, dL_drgb, dL_ddiffuse_light, debug=False):
    """-> (dL_dbase_color, dL_droughness, dL_dmetallic, dL_dnormals, dL_dviewdirs, dL_dincidents_shs, dL_ddirect_shs,
    dL_dvisibility_shs)"""
    L = _lib.lib()
    P, Si, Sd, Sv = _re_common(base_color, incidents, direct_shs, visibility_shs)
    dev = base_color.device
    t = [_c(x) for x in (base_color, roughness, metallic, normals, viewdirs, incidents, direct_shs, visibility_shs)]
    dirs, g_rgb, g_dl = _c(incident_dirs), _c(dL_drgb), _c(dL_ddiffuse_light)
    outs = [torch.zeros_like(x) for x in (t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7])]
    with torch.cuda.device(dev):
        st = L.r3dg_render_equation_backward(_lib.current_stream(), P, Si, Sd, Sv, *[x.data_ptr() for x in t],
                                             int(sample_num), dirs.data_ptr(), g_rgb.data_ptr(), g_dl.data_ptr(),
                                             *[o.data_ptr() for o in outs])
    _lib.check(st, "render_equation_backward")
    return tuple(outs)
