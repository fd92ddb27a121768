"""CPU oracle for the per-Gaussian shading integral -- TEST INFRASTRUCTURE ONLY.

Restates the LIVE stage-2 model of the reference:
    rendering_equation   gaussian_renderer/neilf.py:339-371
    GGX_specular         gaussian_renderer/neilf.py:374-407
    eval_sh              utils/sh_utils.py:71-128
    DirectLightMap.direct_light / EnvLight.direct_light   scene/direct_light_map.py:70-83, scene/envmap.py:35-53
in device-agnostic, dtype-generic torch (float32 or float64) so autograd provides the gradients.

PINNED: tests/golden/shading_reference.npz holds inputs/outputs/gradients produced by importing and running
the reference's own Python functions (tests/golden/make_golden.py); tests/test_oracle_cpu.py checks this
restatement against them.  The product package never imports this module.
"""
import math

import torch
import torch.nn.functional as F

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def sh_basis(deg, d):
    """[...,3] unit dirs -> [..., (deg+1)^2] real SH basis with the reference's signs (sh_utils.py:92-127)."""
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    out = [torch.full_like(x, C0)]
    if deg > 0:
        out += [-C1 * y, C1 * z, -C1 * x]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            out += [C2[0] * xy, C2[1] * yz, C2[2] * (2.0 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
            if deg > 2:
                out += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
                        C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy),
                        C3[5] * z * (xx - yy), C3[6] * x * (xx - 3 * yy)]
    return torch.stack(out, -1)


def env_lookup(env, dirs, transform=None):
    """env [He,We,3] (already activated), dirs [...,3] -> [...,3]; bilinear, align_corners=True, zero padding,
    query (x,y) = (-atan2(dy,dx)/pi, 2*(acos(dz)-1e-6)/pi - 1)   (direct_light_map.py:70-83)."""
    shape = dirs.shape
    d = dirs.reshape(-1, 3)
    if transform is not None:
        d = d @ transform.T
    He, We = env.shape[0], env.shape[1]
    phi = torch.arccos(d[:, 2]) - 1e-6
    theta = torch.atan2(d[:, 1], d[:, 0])
    qy = (phi / math.pi) * 2 - 1
    qx = -theta / math.pi
    ix = (qx + 1) * 0.5 * (We - 1)
    iy = (qy + 1) * 0.5 * (He - 1)
    x0, y0 = torch.floor(ix), torch.floor(iy)
    wx1, wy1 = ix - x0, iy - y0
    wx0, wy0 = 1 - wx1, 1 - wy1
    out = torch.zeros(d.shape[0], 3, dtype=env.dtype, device=env.device)
    for yy, wy in ((y0, wy0), (y0 + 1, wy1)):
        for xx, wx in ((x0, wx0), (x0 + 1, wx1)):
            ok = (xx >= 0) & (xx <= We - 1) & (yy >= 0) & (yy <= He - 1)
            xi = xx.clamp(0, We - 1).long()
            yi = yy.clamp(0, He - 1).long()
            out = out + env[yi, xi] * (wx * wy * ok)[:, None]
    return out.reshape(shape)


def ggx_specular(normal, pts2c, pts2l, roughness, fresnel=0.04):
    """neilf.py:374-407 with out-of-place clamps (identical values and gradients)."""
    L = F.normalize(pts2l, dim=-1)
    V = F.normalize(pts2c, dim=-1)
    H = F.normalize((L + V[:, None, :]) / 2.0, dim=-1)
    N = F.normalize(normal, dim=-1)
    NoV = torch.sum(V * N, dim=-1, keepdim=True)
    N = N * NoV.sign()
    NoL = torch.sum(N[:, None, :] * L, dim=-1, keepdim=True).clamp(1e-6, 1)
    NoV = torch.sum(N * V, dim=-1, keepdim=True).clamp(1e-6, 1)
    NoH = torch.sum(N[:, None, :] * H, dim=-1, keepdim=True).clamp(1e-6, 1)
    VoH = torch.sum(V[:, None, :] * H, dim=-1, keepdim=True).clamp(1e-6, 1)
    alpha = roughness * roughness
    alpha2 = alpha * alpha
    k = (alpha + 2 * roughness + 1.0) / 8.0
    FMi = ((-5.55473) * VoH - 6.98316) * VoH
    frac0 = fresnel + (1 - fresnel) * torch.pow(torch.tensor(2.0, dtype=VoH.dtype, device=VoH.device), FMi)
    frac = frac0 * alpha2[:, None, :]
    nom0 = NoH * NoH * (alpha2[:, None, :] - 1) + 1
    nom1 = NoV * (1 - k) + k
    nom2 = NoL * (1 - k[:, None, :]) + k[:, None, :]
    nom = (4 * math.pi * nom0 * nom0 * nom1[:, None, :] * nom2).clamp(1e-6, 4 * math.pi)
    return frac / nom


def rendering_equation(base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs,
                       incident_areas, transform=None):
    """Returns dict(pbr, diffuse_light, specular, incident_lights, local_incident_lights, global_incident_lights,
    incident_visibility) -- the per-sample tensors already averaged over K (that is the only way
    neilf.py:119-130 consumes them)."""
    deg = int(round(math.sqrt(incidents.shape[1]) - 1))
    glob = env_lookup(env, incident_dirs, transform)
    Y = sh_basis(deg, incident_dirs)                                     # [P,K,M]
    local = torch.einsum("pkm,pmc->pkc", Y, incidents[:, :Y.shape[-1]]).clamp_min(0)
    glob = glob * visibility
    lights = local + glob
    ndi = (normals[:, None] * incident_dirs).sum(-1, keepdim=True).clamp(min=0)
    f_d = base_color[:, None] / math.pi
    f_s = ggx_specular(normals, viewdirs, incident_dirs, roughness, fresnel=0.04)
    transport = lights * incident_areas * ndi
    return dict(pbr=((f_d + f_s) * transport).mean(-2), diffuse_light=transport.mean(-2),
                specular=(f_s * transport).mean(-2), incident_lights=lights.mean(-2),
                local_incident_lights=local.mean(-2), global_incident_lights=glob.mean(-2),
                incident_visibility=visibility.mean(-2))


# ---- ray set of the visibility caches (gaussian_model.py:312-342 -> graphics_utils.py:9-37, sh_utils.py:36-68) ----
def rotation_between_z(vec):
    v1, v2 = -vec[..., 1], vec[..., 0]
    v3 = torch.zeros_like(v1)
    cos_p_1 = (vec[..., 2] + 1).clamp_min(1e-7)
    R = torch.zeros(vec.shape[:-1] + (3, 3), dtype=vec.dtype, device=vec.device)
    R[..., 0, 0] = 1 + (-v3 * v3 - v2 * v2) / cos_p_1
    R[..., 0, 1] = -v3 + v1 * v2 / cos_p_1
    R[..., 0, 2] = v2 + v1 * v3 / cos_p_1
    R[..., 1, 0] = v3 + v1 * v2 / cos_p_1
    R[..., 1, 1] = 1 + (-v3 * v3 - v1 * v1) / cos_p_1
    R[..., 1, 2] = -v1 + v2 * v3 / cos_p_1
    R[..., 2, 0] = -v2 + v1 * v3 / cos_p_1
    R[..., 2, 1] = v1 + v2 * v3 / cos_p_1
    R[..., 2, 2] = 1 + (-v2 * v2 - v1 * v1) / cos_p_1
    return torch.where((vec[..., 2] + 1 > 0)[..., None, None], R, -torch.eye(3, dtype=vec.dtype, device=vec.device).expand_as(R))


def fibonacci_sphere_sampling(normals, sample_num):
    """random_rotate=False variant (the only one the visibility caches use, gaussian_model.py:305-310)."""
    delta = math.pi * (3.0 - math.sqrt(5.0))
    idx = torch.arange(sample_num, dtype=torch.float32, device=normals.device)[None]
    z = (1 - 2 * idx / (2 * sample_num - 1)).clamp_min(math.sin(10 / 180 * math.pi))
    rad = torch.sqrt(1 - z ** 2)
    theta = delta * idx
    y, x = torch.cos(theta) * rad, torch.sin(theta) * rad
    z_samples = torch.stack([x, y, z.expand_as(y)], dim=-2).to(normals.dtype)      # [1,3,K]
    dirs = rotation_between_z(normals) @ z_samples                                  # [P,3,K]
    dirs = F.normalize(dirs, dim=-2).transpose(-1, -2)
    areas = torch.ones_like(dirs)[..., 0:1] * 2 * math.pi
    return dirs, areas
