"""Relight / eval frames (relighting.py:114-170 -> gaussian_renderer/neilf.py:74-209 with is_training=False): per frame the
shading integral at K samples under a fixed HDR environment map (EnvLight, optional rotation), the S=28 eval feature
row, the rasterizer forward and the environment composites.

`RelightRenderer.frame` runs the whole frame through the C ABI -- activations + view directions (r3dg_stage2_activate),
r3dg_shade_forward, r3dg_relight_pack_features, the rasterizer forward, r3dg_relight_compose -- five streaming passes
around the two hot kernels where the reference (and `frame_reference` below, the parity target) spends a few dozen
PyTorch elementwise launches.  Replicas only under multi-GPU: frames are independent, no collective.
"""
import math

import torch
import torch.nn.functional as F

from . import _lib, rasterizer_ops, sampling, shading_ops
from .train_step import rgb_to_srgb, update_visibility   # noqa: F401  (rgb_to_srgb: part of this module's surface)


def _shs_of(model):
    if hasattr(model, "shs"):
        return model.shs
    return torch.cat([model.features_dc.detach(), model.features_rest.detach()], 1).contiguous()


def normal_order(normals, bits=10):
    """Permutation [P] (int32) that visits the Gaussians in an order in which neighbours have NEIGHBOURING normals: Morton code
    of the octahedral image of the unit normal (2 x `bits` bits).  Used by the split-transport relight kernels, whose 64 lanes
    look up the environment map for the same sample index of 64 consecutive Gaussians."""
    n = F.normalize(normals.detach().float(), dim=-1, eps=1e-12)
    p = n / n.abs().sum(-1, keepdim=True).clamp_min(1e-12)
    u, v = p[:, 0], p[:, 1]
    fold = p[:, 2] < 0
    uf = torch.where(fold, (1 - v.abs()) * torch.where(u >= 0, 1.0, -1.0), u)
    vf = torch.where(fold, (1 - u.abs()) * torch.where(v >= 0, 1.0, -1.0), v)
    q = (1 << bits) - 1
    iu = ((uf * 0.5 + 0.5) * q).round().clamp(0, q).long()
    iv = ((vf * 0.5 + 0.5) * q).round().clamp(0, q).long()
    code = torch.zeros_like(iu)
    for b in range(bits):
        code |= ((iu >> b) & 1) << (2 * b)
        code |= ((iv >> b) & 1) << (2 * b + 1)
    return torch.argsort(code).to(torch.int32).contiguous()


def _incidents_of(model):
    if hasattr(model, "incidents"):
        return model.incidents
    return torch.cat([model.incidents_dc.detach(), model.incidents_rest.detach()], 1).contiguous()


class RelightRenderer:
    """`model`: anything holding the RAW parameters xyz, normal, scaling, rotation, opacity, base_color, roughness,
    SH colour (`shs` or `features_dc`/`features_rest`) and incident light (`incidents` or `incidents_dc`/`_rest`) --
    bench_core.GaussianParams and fused_step.FusedStage2Step both do.  `envmap` [He,We,3] HDR (EnvLight.envmap)."""

    def __init__(self, model, envmap, sample_num, process_group=None, cache="transport", regenerate_dirs=True):
        """`cache` -- what is kept between frames while the light does not change:
        "transport" (default): the whole view-independent part of the integral -- per sample (local + global light) x area
            x n.d, per Gaussian diffuse_light and the mean light / visibility columns -- so that a frame evaluates only the
            GGX lobe (r3dg_shade_forward_transport; 12 bytes and ~80 instructions per sample).  Valid while parameters,
            light and visibility are unchanged: the renderer works on its own SNAPSHOT of the parameters (copies taken
            here), and a light that turns with every frame (configs/nerf_syn_light, configs/tnt) drops to the lookup-in-kernel
            path by itself (see _taps_for), so nothing can go stale.  `regenerate_dirs`: the kernel rebuilds each
            direction from the normal and the Fibonacci table instead of reading the direction cache.
        "radiance": only the sampled environment radiance of every cached direction (12 bytes per sample) is kept and the
            full integral (r3dg_shade_forward_cached) runs per frame."""
        if cache not in ("radiance", "transport"):
            raise RuntimeError("RelightRenderer: cache must be 'radiance' or 'transport'")
        self.cache, self.regenerate_dirs = cache, bool(regenerate_dirs)
        # lookup cache of the direction set for ONE light (see _taps_for)
        self._taps = self._taps_key = self._taps_ref = None
        self._light_key = self._light_ref = None
        self._light_changes = 0
        self._area_key, self._uniform_area = None, None
        self._consts = self._zsamples = None
        self._split = None                          # split-transport cache of a light that turns with every frame (_split_cache)
        self._order_stream = None                   # the instance ordering of a frame runs there, beside the shading kernel
        d = lambda t: t.detach().clone().contiguous()       # a snapshot: the caches below are only valid for THESE values
        self.xyz, self.normal = d(model.xyz), d(model.normal)
        self.scaling, self.rotation, self.opacity = d(model.scaling), d(model.rotation), d(model.opacity)
        self.base_color, self.roughness = d(model.base_color), d(model.roughness)
        self.shs, self.incidents = d(_shs_of(model)), d(_incidents_of(model))
        self.envmap = d(envmap)
        if self.envmap.dim() != 3 or self.envmap.shape[2] != 3:
            raise RuntimeError("envmap must be [He,We,3]")
        for t in (self.xyz, self.envmap):
            if not t.is_cuda:
                raise RuntimeError("RelightRenderer needs device tensors (there is no CPU path)")
        self.dev = dev = self.xyz.device
        self.P = P = self.xyz.shape[0]
        self.K, self.M = sample_num, self.shs.shape[1]
        if self.incidents.shape[1] != self.M:      # the reference gives both the same degree (gaussian_model.py:421, :450)
            raise RuntimeError("RelightRenderer: colour and incident-light SH must hold the same number of coefficients")
        f = dict(dtype=torch.float32, device=dev)
        self.a_scales, self.a_rot = torch.empty(P, 3, **f), torch.empty(P, 4, **f)
        self.a_opacity, self.a_normal = torch.empty(P, 1, **f), torch.empty(P, 3, **f)
        self.a_base, self.a_rough = torch.empty(P, 3, **f), torch.empty(P, 1, **f)
        self.a_viewdirs = torch.empty(P, 3, **f)
        self.shade_out = torch.empty(P, shading_ops.NOUT, **f)
        self.features = torch.empty(P, 28, **f)
        if dev.type == "cuda" and torch.cuda.is_available():
            from .fused_step import shared_stream
            self._order_stream = shared_stream(dev, "order")
        with torch.no_grad():
            self._activate(torch.zeros(3, device=dev))
            self.visibility, self.incident_dirs, self.incident_areas, self.tracer = update_visibility(
                self.xyz, self.a_scales, self.a_rot, self.a_opacity, self.a_normal, sample_num, group=process_group)

    def _taps_for(self, tr, He, We, given=None):
        """shading_ops.build_taps of the direction cache for this light rotation (None = identity); one entry is kept, so
        a static light costs one build.  `tr`: the rotation on the device, `given`: the caller's tensor it came from."""
        # Identity of the light WITHOUT reading device memory back: a host tensor is keyed by its nine values; a device
        # tensor by its storage address + version counter -- and a reference to it is kept for as long as it is the key, so
        # that the allocator cannot hand the same address to the NEXT frame's matrix (relighting.py:162-163 builds a new
        # tensor per frame: without the reference a recycled address would read as "the light did not move").
        if tr is None:
            ident = None
        elif given is not None and not given.is_cuda:
            ident = tuple(float(x) for x in given.detach().reshape(-1).tolist())
        else:
            ident = (tr.data_ptr(), tr._version)
        # (the map and the direction cache are held by this object, so their addresses are theirs alone for as long as the
        # key is; a caller that swaps either in gets a new key through the version / address pair of the new tensor while the
        # old one is still referenced below)
        key = (ident, He, We, self.incident_dirs.data_ptr(), self.incident_dirs._version, self.envmap.data_ptr(),
               self.envmap._version)
        # A light that turns with EVERY frame (configs/nerf_syn_light, configs/tnt): writing the cache costs what the lookup
        # inside the shading kernel costs and the kernel would then still have to read it back -- from the second
        # consecutive change on, no cache: None = r3dg_shade_forward_cached evaluates the lookup itself (measured: 3.6 ms
        # per frame with a rebuild, 2.5 without).  A light that stops turning gets its cache back on
        # the next frame.
        changed = self._light_key != key
        self._light_key, self._light_ref = key, (tr, self.incident_dirs, self.envmap)
        self._light_changes = (self._light_changes + 1) if changed else 0
        if self._light_changes >= 2 and self._area_key is not None:          # (the first frame always builds)
            return None
        if self._taps_key != key:
            # the HDR map is fixed while relighting, so the SAMPLED RADIANCE of every cached direction is cached (not just
            # the lookup coordinates): the shading kernel then reads 12 bytes per sample and no texture
            self._taps = shading_ops.build_taps(self.incident_dirs, He, We, tr, radiance_of=self.envmap)
            self._taps_key, self._taps_ref = key, (tr, self.incident_dirs, self.envmap)
            if self._area_key != self.incident_areas.data_ptr():
                # fibonacci_sphere_sampling gives every sample the area 2 pi: then the area cache need not be read at all
                lo, hi = float(self.incident_areas.min()), float(self.incident_areas.max())
                self._uniform_area = lo if lo == hi else None
                self._area_key = self.incident_areas.data_ptr()
            if self.cache == "transport":
                # radiance -> transport in place, + the per-Gaussian constants (the buffer must not be handed to
                # r3dg_shade_forward_cached any more: frame() takes the transport kernel whenever this cache is live)
                if self._zsamples is None:
                    self._zsamples = sampling.fibonacci_z_samples(self.K, self.dev)[0].t().contiguous()      # [K,3]
                self._consts = shading_ops.build_transport(
                    self.a_normal, self.incidents, self.visibility, self.incident_dirs, self.incident_areas, self._uniform_area,
                    self._taps, self._consts)
        return self._taps

    def _split_cache(self, tr=None, given=None):
        """The light-independent half of the transport, for a light that turns with every frame (relighting.py:160-161 with
        configs/nerf_syn_light / configs/tnt light_transform.json): per sample (local light x a, a) and the visibility,
        sample-major, in an order sorted by normal; the per-frame kernel (r3dg_shade_forward_split) does the lat-long lookup
        of the rotated direction itself.  Only for the configuration the reference's relighting uses (16 incident-light
        coefficients, the Fibonacci ray set with its uniform area, a map of at most 4095 x 4095 texels) and only for a light
        transform that is a ROTATION (the kernel evaluates the GGX lobe in the light's frame; a transform given on the host is
        checked here, one on the device is the caller's promise); None otherwise (the general kernel with the lookup inside
        then runs).  Like _taps_for's entry, the cache is KEYED: the sample half on the direction / visibility caches
        (address + version) and `regenerate_dirs`, the footprints on the map (address + version + size) -- a caller that swaps
        or edits `envmap`, `visibility` or `incident_dirs` between two frames gets the rebuilt half, and the tensors keyed on
        are referenced for as long as they are the key."""
        He, We = self.envmap.shape[0], self.envmap.shape[1]
        if not shading_ops.split_supported(self.K, self.M, He, We, self._uniform_area):
            self._split = False
            return None
        if given is not None and not given.is_cuda:
            t = given.detach().to(torch.float64).reshape(3, 3)
            if float((t @ t.t() - torch.eye(3, dtype=torch.float64)).abs().max()) > 1e-4:
                return None                       # scaled / sheared light transform: the general kernel handles it
        sample_key = (self.incident_dirs.data_ptr(), self.incident_dirs._version, self.visibility.data_ptr(),
                      self.visibility._version, self.regenerate_dirs)
        env_key = (self.envmap.data_ptr(), self.envmap._version, He, We)
        sp = self._split if isinstance(self._split, dict) else None
        if sp is None or sp["sample_key"] != sample_key:
            zs = sampling.fibonacci_z_samples(self.K, self.dev)[0].t().contiguous()
            new = shading_ops.build_split(normal_order(self.a_normal), self.a_normal, self.incidents, self.visibility,
                                          None if self.regenerate_dirs else self.incident_dirs, zs, float(self._uniform_area))
            new.update(sample_key=sample_key, sample_ref=(self.incident_dirs, self.visibility),
                       env4=None if sp is None else sp["env4"], env_key=None if sp is None else sp["env_key"],
                       env_ref=None if sp is None else sp["env_ref"])
            sp = self._split = new
        if sp["env_key"] != env_key:
            sp["env4"] = shading_ops.env_footprints(self.envmap)
            sp["env_key"], sp["env_ref"] = env_key, self.envmap
        return sp

    def _activate(self, campos):
        with torch.cuda.device(self.dev):
            st = _lib.lib().r3dg_stage2_activate(
                _lib.current_stream(), self.P, self.xyz.data_ptr(), self.scaling.data_ptr(), self.rotation.data_ptr(),
                self.opacity.data_ptr(), self.normal.data_ptr(), self.base_color.data_ptr(), self.roughness.data_ptr(),
                campos.contiguous().data_ptr(), self.a_scales.data_ptr(), self.a_rot.data_ptr(),
                self.a_opacity.data_ptr(), self.a_normal.data_ptr(), self.a_base.data_ptr(), self.a_rough.data_ptr(),
                self.a_viewdirs.data_ptr(), None, None)
        _lib.check(st, "stage2_activate")

    def _shade_cached(self, L, stream, P, He, We, tr, taps):
        _lib.check(L.r3dg_shade_forward_cached(
            stream(), P, self.K, self.M, self.a_base.data_ptr(), self.a_rough.data_ptr(), self.a_normal.data_ptr(),
            self.a_viewdirs.data_ptr(), self.incidents.data_ptr(), self.envmap.data_ptr(), He, We, _lib.ptr(tr),
            self.visibility.data_ptr(), self.incident_dirs.data_ptr(),
            None if self._uniform_area is not None else self.incident_areas.data_ptr(), self._uniform_area or 0.0,
            # 2 = R3DG_SHADE_TAPS_ARE_RADIANCE; no cache (a light that changes every frame): lookup in the kernel
            taps.data_ptr() if taps is not None else None, 2 if taps is not None else 0,
            self.shade_out.data_ptr()), "shade_forward")

    @torch.no_grad()
    def frame(self, cam, bg, env_transform=None, outputs=("pbr_env",)):
        """-> dict with the rasterizer's public outputs ("render", "opacity", "depth", "feature", "pseudo_normal",
        "num_rendered", "num_contrib", "radii") and the requested composites out of "pbr_env", "render_env", "env_only"
        (neilf.py:203-207), each [3,H,W]."""
        L = _lib.lib()
        P, dev = self.P, self.dev
        H, W = cam.image_height, cam.image_width
        vm = cam.world_view_transform.contiguous()
        campos = cam.camera_center.contiguous()
        tr = None if env_transform is None else env_transform.to(dev, torch.float32).contiguous()
        He, We = self.envmap.shape[0], self.envmap.shape[1]
        empty = torch.Tensor([])
        stream = _lib.current_stream
        with torch.cuda.device(dev):
            self._activate(campos)
            # The rasterizer's front end depends on the camera and the geometry only: its projection is queued NOW, and
            # finish() below puts the instance ordering on the ordering stream -- both run beside the shading kernel instead of
            # behind it (the feature rows, which the shading produces, are first read by the tile kernel inside finish()).
            pending = rasterizer_ops.rasterize_gaussians_begin(
                bg, self.xyz, self.features, empty, self.a_opacity, self.a_scales, self.a_rot, 1.0, empty, vm,
                cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.cx, cam.cy, H, W, self.shs, 3, campos, False,
                True, False, want_weights=False)                   # (a frame has no use for the per-Gaussian blend weights)
            # the lat-long lookups of the cached directions are constant for a fixed light rotation: cached per transform
            taps = self._taps_for(tr, He, We, env_transform)
            sp = None
            if taps is not None and self.cache == "transport":
                shading_ops.shade_forward_transport(
                    self.a_base, self.a_rough, self.a_normal, self.a_viewdirs, taps, self._consts, self._zsamples,
                    None if self.regenerate_dirs else self.incident_dirs, self.shade_out)
            elif taps is None and (sp := self._split_cache(tr, env_transform)) is not None:
                # a light that turns with every frame: the light-independent half of the transport is cached, the lookup of the
                # rotated direction happens in the kernel (lane = Gaussian, sorted by normal)
                shading_ops.shade_forward_split(sp, self.a_base, self.a_rough, self.a_normal, self.a_viewdirs, tr, sp["env4"],
                                                He, We, self.shade_out)
            else:
                self._shade_cached(L, stream, P, He, We, tr, taps)
            _lib.check(L.r3dg_relight_pack_features(
                stream(), P, self.xyz.data_ptr(), vm.data_ptr(), self.a_normal.data_ptr(), self.a_base.data_ptr(),
                self.a_rough.data_ptr(), self.shade_out.data_ptr(), self.features.data_ptr()), "relight_pack_features")
            fw = pending.finish(self._order_stream)
            R, n_contrib, image, opacity, depth, feature, pseudo_normal, sxyz, weights, radii = fw[:10]
            res = dict(num_rendered=R, num_contrib=n_contrib, render=image, opacity=opacity, depth=depth, feature=feature,
                       pseudo_normal=pseudo_normal, surface_xyz=sxyz, radii=radii)
            want = {k: torch.empty(3, H, W, dtype=torch.float32, device=dev) for k in outputs}
            for k in want:
                if k not in ("pbr_env", "render_env", "env_only"):
                    raise RuntimeError("unknown relight output %r" % k)
            if want:
                g = lambda k: want[k].data_ptr() if k in want else None
                _lib.check(L.r3dg_relight_compose(
                    stream(), W, H, W / (2.0 * cam.tanfovx), H / (2.0 * cam.tanfovy), cam.cx, cam.cy, vm.data_ptr(),
                    _lib.ptr(tr), self.envmap.data_ptr(), He, We, image.data_ptr(), opacity.data_ptr(),
                    feature.data_ptr(), n_contrib.data_ptr(), g("pbr_env"), g("render_env"), g("env_only")),
                    "relight_compose")
            res.update(want)
        return res


def env_directions(cam, envmap, env_transform=None):
    """Per-pixel environment colour [3,H,W]: Camera.get_world_directions (scene/cameras.py:79-91) +
    EnvLight.direct_light (scene/envmap.py:35-53) as plain torch ops."""
    H, W = cam.image_height, cam.image_width
    dev = envmap.device
    fx, fy = W / (2 * cam.tanfovx), H / (2 * cam.tanfovy)
    v, u = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
    d = torch.stack([(u - cam.cx) / fx, (v - cam.cy) / fy, torch.ones_like(u, dtype=torch.float32)], 0)
    d = F.normalize(d, dim=0)
    c2w_rot = cam.world_view_transform[:3, :3]            # (W2C^T)[:3,:3] = R_w2c^T = R_c2w
    d = (c2w_rot @ d.reshape(3, -1)).t()
    if env_transform is not None:
        d = d @ env_transform.to(dev).t()
    phi = torch.arccos(d[:, 2].clamp(-1, 1)) - 1e-6
    theta = torch.atan2(d[:, 1], d[:, 0])
    grid = torch.stack((-theta / math.pi, (phi / math.pi) * 2 - 1), -1)[None, None]
    col = F.grid_sample(envmap.permute(2, 0, 1)[None], grid, align_corners=True)
    return col[0, :, 0].reshape(3, H, W)


@torch.no_grad()
def frame_reference(renderer, cam, bg, env_transform=None, exact_activations=False):
    """The same frame through the drop-in ops + PyTorch glue, shaped like the reference's render_view(is_training=False):
    the parity target of RelightRenderer.frame (tests) and the 'before' of the relight measurement.
    `exact_activations`: take the activated parameters / view directions from the renderer's activation kernel instead
    of torch (exp / sigmoid differ in the last bit, which the rasterizer's alpha >= 1/255 test can turn into a visible
    difference on a few pixels) -- isolates the glue under test."""
    from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    r = renderer
    if exact_activations:
        r._activate(cam.camera_center)
        base_color, roughness, normal = r.a_base.clone(), r.a_rough.clone(), r.a_normal.clone()
        opacity_a, scales, rot, viewdirs = r.a_opacity.clone(), r.a_scales.clone(), r.a_rot.clone(), r.a_viewdirs.clone()
    else:
        base_color = 0.03 + 0.77 * torch.sigmoid(r.base_color)
        roughness = 0.09 + 0.9 * torch.sigmoid(r.roughness)
        normal = F.normalize(r.normal, dim=-1, eps=1e-3)
        opacity_a, scales, rot = torch.sigmoid(r.opacity), torch.exp(r.scaling), F.normalize(r.rotation)
        viewdirs = F.normalize(cam.camera_center - r.xyz, dim=-1)
    tr = None if env_transform is None else env_transform.to(r.dev, torch.float32).contiguous()
    pbr, diffuse, rest = shading_ops.shade(base_color, roughness, normal, viewdirs, r.incidents, r.envmap, r.visibility,
                                           r.incident_dirs, r.incident_areas, tr)
    xyz_h = torch.cat([r.xyz, torch.ones_like(r.xyz[:, :1])], -1)
    depths = (xyz_h @ cam.world_view_transform)[:, 2:3]
    feats = torch.cat([depths, depths.square(), pbr, normal, base_color, roughness, diffuse, rest], -1)      # S = 28
    rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, cam.cx, cam.cy, bg,
                                       1.0, cam.world_view_transform, cam.full_proj_transform, 3, cam.camera_center,
                                       False, True, True, False)
    outs = GaussianRasterizer(rs)(r.xyz, torch.zeros_like(r.xyz), opacity_a, shs=r.shs, scales=scales, rotations=rot,
                                  features=feats)
    _, n_contrib, image, opacity, depth, feature, pn, sxyz, weights, radii = outs
    feat = feature / opacity.clamp_min(1e-5) * (n_contrib > 0)
    env_rgb = env_directions(cam, r.envmap, tr)
    return dict(render=image, opacity=opacity, feature=feature, num_rendered=outs[0],
                pbr_env=rgb_to_srgb(feat[2:5] * opacity + (1 - opacity) * env_rgb),
                render_env=image + (1 - opacity) * rgb_to_srgb(env_rgb), env_only=rgb_to_srgb(env_rgb))


# ---- multi-object composition (relighting.py:28-52 scene_composition; GaussianModel.set_transform
# scene/gaussian_model.py:84-95, create_from_gaussians :344-356) -----------------------------------------------------------
def _quaternion_of(R):
    """Unit quaternion (w, x, y, z) of rotation matrices [n,3,3] the way the reference extracts it
    (utils/general_utils.py:105-117: w from the trace, clamped at 1e-7, then normalised)."""
    qw = torch.sqrt((1 + R[:, 0, 0] + R[:, 1, 1] + R[:, 2, 2]).clamp_min(1e-7)) / 2
    q = torch.stack((qw, (R[:, 2, 1] - R[:, 1, 2]) / (4 * qw), (R[:, 0, 2] - R[:, 2, 0]) / (4 * qw),
                     (R[:, 1, 0] - R[:, 0, 1]) / (4 * qw)), dim=-1)
    return F.normalize(q, dim=-1)


def _quaternion_product(a, b):
    """Hamilton product a * b of (w, x, y, z) rows (utils/general_utils.py:141-151); `a` broadcasts over `b`."""
    w1, x1, y1, z1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    w2, x2, y2, z2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    return torch.stack((w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2), dim=1)


TRANSFORMED = ("xyz", "normal", "scaling", "rotation")


@torch.no_grad()
def transform_object(params, transform):
    """GaussianModel.set_transform(transform=T) on a dict of RAW parameter tensors (xyz, normal, scaling, rotation + any
    others, which pass through): similarity transform T [4,4] -- per-axis scale = row norms of T[:3,:3], rotation =
    T[:3,:3] / scale; positions x T^T, log-scales + log(scale), normals x R^T, rotation quaternions pre-multiplied."""
    T = transform.to(params["xyz"].device, torch.float32).reshape(4, 4)
    scale = T[:3, :3].norm(dim=-1)
    out = dict(params)
    out["scaling"] = torch.log(torch.exp(params["scaling"]) * scale)
    xyz_h = torch.cat([params["xyz"], torch.ones_like(params["xyz"][:, :1])], dim=-1)
    out["xyz"] = (xyz_h @ T.T)[:, :3].contiguous()
    R = T[:3, :3] / scale[:, None]
    out["normal"] = params["normal"] @ R.T
    out["rotation"] = _quaternion_product(_quaternion_of(R[None]), params["rotation"])
    return out


@torch.no_grad()
def compose_scenes(objects, transforms):
    """scene_composition (relighting.py:28-52): every object's parameters under its own 4x4 transform, concatenated row
    wise in the order given; the incident-light coefficients of the composite are zeroed (:49-50).  `objects`: dicts of
    raw parameter tensors with identical keys.  Returns one dict (feed it to RelightRenderer through a namespace)."""
    if not objects or len(objects) != len(transforms):
        raise RuntimeError("compose_scenes needs one transform per object")
    moved = [transform_object(o, t) for o, t in zip(objects, transforms)]
    keys = list(objects[0].keys())
    for m in moved:
        if list(m.keys()) != keys:
            raise RuntimeError("compose_scenes: objects must hold the same parameter names in the same order")
    out = {k: torch.cat([m[k] for m in moved], dim=0).contiguous() for k in keys}
    for k in out:
        if k.startswith("incidents"):
            out[k].zero_()
    return out
