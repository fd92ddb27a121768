"""Host side of the fused shading integral (C ABI r3dg_shade_forward / r3dg_shade_backward).

`rendering_equation(...)` has the signature and return convention of the reference's live
`gaussian_renderer.neilf.rendering_equation` (neilf.py:339-371), so `neilf.rendering_equation = shading_ops.rendering_equation`
swaps the op in without editing the reference file.  The per-sample [P,K,*] tensors of `extra_results` are only ever
consumed through `.mean(-2)` (neilf.py:119-130); they are returned already reduced with a singleton sample axis
([P,1,C]) so that `.mean(-2)` and the eval-time `torch.cat(..., dim=0)` keep working on unchanged caller code.
"""
import torch

from . import _lib

NOUT = 19


def _c(t):
    if t.dtype != torch.float32 or not t.is_cuda:
        raise RuntimeError("shading ops expect float32 CUDA(HIP) tensors")
    return t.contiguous()


def shade_forward(base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs, incident_areas,
                  env_transform=None):
    """-> out[P,19] = pbr3 diffuse3 specular3 lights3 local3 global3 vis1 (see include/r3dg_hip.h)."""
    L = _lib.lib()
    P, K = incident_dirs.shape[0], incident_dirs.shape[1]
    M = incidents.shape[1]
    He, We = env.shape[-3], env.shape[-2]
    t = [_c(x) for x in (base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs,
                         incident_areas)]
    tr = _c(env_transform) if env_transform is not None else None
    out = torch.empty((P, NOUT), dtype=torch.float32, device=base_color.device)
    with torch.cuda.device(base_color.device):
        st = L.r3dg_shade_forward(_lib.current_stream(), P, K, M, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                  t[3].data_ptr(), t[4].data_ptr(), t[5].data_ptr(), He, We,
                                  tr.data_ptr() if tr is not None else None, t[6].data_ptr(), t[7].data_ptr(),
                                  t[8].data_ptr(), out.data_ptr())
    _lib.check(st, "shade_forward")
    return out


def shade_backward(base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs, incident_areas,
                   dL_dpbr, dL_ddiffuse_light, env_transform=None):
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
    d_inc = torch.empty((P, M, 3), dtype=torch.float32, device=dev)
    d_env = torch.zeros_like(t[5])
    with torch.cuda.device(dev):
        st = L.r3dg_shade_backward(_lib.current_stream(), P, K, M, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(),
                                   t[3].data_ptr(), t[4].data_ptr(), t[5].data_ptr(), He, We,
                                   tr.data_ptr() if tr is not None else None, t[6].data_ptr(), t[7].data_ptr(),
                                   t[8].data_ptr(), t[9].data_ptr(), t[10].data_ptr(), d_base.data_ptr(),
                                   d_rough.data_ptr(), d_view.data_ptr(), d_inc.data_ptr(), d_env.data_ptr())
    _lib.check(st, "shade_backward")
    return d_base, d_rough, d_view, d_inc, d_env


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
    """Differentiable fused integral -> (pbr[P,3], diffuse_light[P,3], rest[P,13] = specular3 lights3 local3 global3 vis1)."""
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
