"""Stage-2 (neilf) hot-path step, restating the op-level content of the reference's training iteration without its
dataset / logging machinery:
    GaussianModel.update_visibility      scene/gaussian_model.py:312-342   (BVH build + K rays per Gaussian, once)
    render_view (is_training=True)       gaussian_renderer/neilf.py:15-209 (shading -> S=16 feature row -> rasterize)
    calculate_loss (core terms)          gaussian_renderer/neilf.py:212-318
Everything heavy runs in the HIP ops; the glue is plain PyTorch on the same stream."""
import math
import os

import torch
import torch.nn.functional as F

from . import sampling
from .bvh import RayTracer
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from .shading_ops import shade


def inverse_covariance(scales, rotations):
    """get_inverse_covariance (gaussian_model.py:257-260): R diag(1/s)^2 R^T as the 6-vector."""
    q = F.normalize(rotations)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    L = R * (1.0 / scales)[:, None, :]
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1).contiguous()


@torch.no_grad()
def update_visibility(xyz, scales, rotations, opacity, normal, sample_num, group=None, tracer_cls=None):
    """-> (visibility[P,K,1], incident_dirs[P,K,3], incident_areas[P,K,1], tracer); chunked like the reference
    (chunk = P // ((K-1)//24 + 1)) so the transient [chunk,K,3] ray tensors stay bounded.

    Data parallel: with an initialised process group of W > 1 ranks every rank builds the same BVH
    (replicated, deterministic), traces only ITS contiguous block of ceil(P/W) ray bundles (of the Morton-ordered bundle
    list, see below) and ONE all-gather assembles the [P,K,1] visibility on every rank; directions and areas are pure
    functions of the normals and are evaluated locally for all rows.  `tracer_cls` (tests) replaces bvh.RayTracer."""
    import torch.distributed as dist
    world = rank = None
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    # a one-rank group takes the all-gather path only for the single-GPU RCCL smoke test (fused_step._world_of)
    gather = bool(world) and (world > 1 or os.environ.get("R3DG_DP_SINGLE_RANK") == "1")
    if not world or world == 1:
        world, rank = 1, 0
    tracer = (tracer_cls or RayTracer)(xyz, scales, rotations)
    cinv = inverse_covariance(scales, rotations)
    op = opacity[:, 0].contiguous()
    P = xyz.shape[0]
    per = -(-P // world)                                   # rows per rank (the last block may be short or empty)
    lo, hi = min(P, rank * per), min(P, (rank + 1) * per)
    chunk = max(1, P // ((sample_num - 1) // 24 + 1))
    # The bundles are traced in MORTON order of their origin Gaussian (the leaf order of the tree just built): consecutive
    # ray blocks then start next to each other and walk the same subtrees, which is what the trace kernel's per-XCD L2s
    # need (csrc/bvh.hip).  Results are scattered back to the caller's order; values are those of any other order.
    order = getattr(tracer, "tree", None)
    order = order[P - 1:, 3].long() if (order is not None and P > 1) else torch.arange(P, device=xyz.device)
    dirs_all, areas_all = [], []
    for off in range(0, P, chunk):
        dirs, areas = sampling.fibonacci_sphere_sampling(normal[off:off + chunk], sample_num)
        dirs_all.append(dirs)
        areas_all.append(areas)
    dirs_all, areas_all = torch.cat(dirs_all, 0), torch.cat(areas_all, 0)
    mine = order[lo:hi]                                        # this rank's share of the Morton-ordered bundles
    vis_mine = torch.empty(mine.numel(), sample_num, 1, dtype=torch.float32, device=xyz.device)
    for a in range(0, mine.numel(), chunk):
        sel = mine[a:a + chunk]
        d = dirs_all[sel]
        res = tracer.trace_visibility(xyz[sel][:, None].expand_as(d), d, xyz, cinv, op, normal)
        vis_mine[a:a + sel.numel()] = res["visibility"]
    if not gather:
        full = torch.empty(P, sample_num, 1, dtype=torch.float32, device=xyz.device)
        full[order] = vis_mine
        return full, dirs_all, areas_all, tracer
    padded = torch.zeros(per, sample_num, 1, dtype=torch.float32, device=xyz.device)
    padded[:mine.numel()] = vis_mine
    gathered = torch.empty(world * per, sample_num, 1, dtype=torch.float32, device=xyz.device)
    dist.all_gather(list(gathered.view(world, per, sample_num, 1).unbind(0)), padded, group=group)
    # rank r traced order[r*per : (r+1)*per]: the first P gathered rows are the visibilities in Morton order
    full = torch.empty(P, sample_num, 1, dtype=torch.float32, device=xyz.device)
    full[order] = gathered[:P]
    return full, dirs_all, areas_all, tracer


LAMBDA_DSSIM = 0.2          # arguments/__init__.py:125


def ssim(img1, img2, window_size=11):
    """utils/loss_utils.py:20-63 restated: Gaussian window (sigma 1.5), zero padding, mean over channels and pixels."""
    C = img1.size(-3)
    g = torch.tensor([math.exp(-(x - window_size // 2) ** 2 / (2 * 1.5 ** 2)) for x in range(window_size)],
                     dtype=torch.float32)
    g = (g / g.sum()).to(img1.device)
    window = (g[:, None] @ g[None, :])[None, None].expand(C, 1, window_size, window_size).contiguous()
    x, y = img1[None] if img1.dim() == 3 else img1, img2[None] if img2.dim() == 3 else img2
    pad = window_size // 2
    mu1 = F.conv2d(x, window, padding=pad, groups=C)
    mu2 = F.conv2d(y, window, padding=pad, groups=C)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(x * x, window, padding=pad, groups=C) - mu1_sq
    sigma2_sq = F.conv2d(y * y, window, padding=pad, groups=C) - mu2_sq
    sigma12 = F.conv2d(x * y, window, padding=pad, groups=C) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean()


def image_loss(img, gt):
    """(1 - lambda_dssim) * L1 + lambda_dssim * (1 - SSIM)  (neilf.py:225-239, render.py likewise)."""
    return (1.0 - LAMBDA_DSSIM) * (img - gt).abs().mean() + LAMBDA_DSSIM * (1.0 - ssim(img, gt))


def psnr(img1, img2):
    """utils/image_utils.py:25-30: one value per slice of the leading axis (the reference logs `.mean()` of it,
    neilf.py:229): 20 log10(1 / sqrt(mean squared error of that slice))."""
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def rgb_to_srgb(img):
    """utils/graphics_utils.py:207-213 with clip=True (what render_view puts into results["pbr"], neilf.py:179): the sRGB
    curve, then clamp to [0,1] -- the clamp stops the gradient of saturated pixels.  Pinned by tests/golden/ssim_reference.npz."""
    curve = torch.where(img > 0.0031308, torch.pow(torch.clamp_min(img, 0.0031308), 1.0 / 2.4) * 1.055 - 0.055, 12.92 * img)
    return curve.clamp(0.0, 1.0)


def tv_loss(x):
    """utils/loss_utils.py:113-117: mean squared forward difference along the last two axes (pinned by
    tests/golden/ssim_reference.npz)."""
    return (x[..., 1:, :] - x[..., :-1, :]).square().mean() + (x[..., :, 1:] - x[..., :, :-1]).square().mean()


def spatial_gradient(x):
    """kornia.filters.spatial_gradient(x[B,C,H,W], mode='sobel', order=1, normalized=True) of kornia 0.6.12 (the version
    readme.md:31-32 pins; the package is not in this image, so its published algorithm is restated): cross-correlation
    with the 3x3 Sobel kernel [[-1,0,1],[-2,0,2],[-1,0,1]] (x) and its transpose (y), each divided by the sum of its
    absolute values (8), on a replicate-padded input -> [B,C,2,H,W]."""
    b, c, h, w = x.shape
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]], dtype=x.dtype, device=x.device) / 8.0
    k = torch.stack([kx, kx.t()])[:, None]
    out = F.conv2d(F.pad(x.reshape(b * c, 1, h, w), (1, 1, 1, 1), mode="replicate"), k)
    return out.reshape(b, c, 2, h, w)


def first_order_edge_aware_loss(data, img):
    """utils/loss_utils.py:104-105."""
    return (spatial_gradient(data[None])[0].abs() * torch.exp(-spatial_gradient(img[None])[0].abs())).sum(1).mean()


# script/run_nerf.sh:7-14 (stage 1): --lambda_normal_render_depth 0.01 --lambda_normal_smooth 0.01 --lambda_mask_entropy 0.1
# --lambda_depth_var 1e-2; lambda_dssim 0.2 (arguments/__init__.py:125)
STAGE1_WEIGHTS = dict(l1=1.0, mask_entropy=0.1, normal_render_depth=0.01, normal_smooth=0.01, depth_var=1e-2)


def depth_var_weight(lambda_depth_var, iteration):
    """render.py:202: lambda_depth_var * min(10^(iteration/5000), 100)."""
    return lambda_depth_var * min(math.pow(10, iteration / 5000), 100)


def stage1_loss(outs, gt, image_mask=None, weights=None, iteration=0):
    """calculate_loss of gaussian_renderer/render.py:137-223 on the rasterizer's 10 public outputs (render_view :107-115):
    the parity target of fused_step.FusedStage1Step (plain PyTorch, autograd)."""
    w = dict(STAGE1_WEIGHTS)
    if weights:
        w.update(weights)
    num_rendered, n_contrib, image, opacity, depth, feature, pseudo_normal, xyz, weights_, radii = outs
    mask = (n_contrib > 0)
    feat = feature / opacity.clamp_min(1e-5) * mask
    normal, r_depth, r_depth2 = feat[:3], feat[3:4], feat[4:5]
    m = torch.ones_like(opacity) if image_mask is None else image_mask
    loss = w["l1"] * image_loss(image, gt)
    if w["mask_entropy"] > 0:
        o = opacity.clamp(1e-6, 1 - 1e-6)
        loss = loss + w["mask_entropy"] * -(m * torch.log(o) + (1 - m) * torch.log(1 - o)).mean()
    if w["normal_render_depth"] > 0:
        loss = loss + w["normal_render_depth"] * F.mse_loss(normal * m, pseudo_normal.detach() * m)
    if w["normal_smooth"] > 0:
        loss = loss + w["normal_smooth"] * first_order_edge_aware_loss(normal, gt)
    if w["depth_var"] > 0:
        var = r_depth2 - r_depth.square()
        loss = loss + depth_var_weight(w["depth_var"], iteration) * var.clamp_min(1e-6).sqrt().mean()
    return loss


# The stage-2 objectives of the three run scripts (lambda_* of neilf.py:212-318; lambda_dssim 0.2 and lambda_pbr 1 are the
# defaults of arguments/__init__.py:125-126):
#   script/run_nerf.sh:20-39   lambda_light 0.01, lambda_env_smooth 0.01, the three edge-aware smoothness terms 0
#   script/run_syn4.sh:22-42   + lambda_base_color_smooth 1, lambda_roughness_smooth 0.5, lambda_light_smooth 1; every geometry
#   script/run_dtu.sh:24-45      rate (position, normal, sh, opacity, scaling, rotation) 0: only base colour, roughness,
#                                incident light and the environment texture train
STAGE2_WEIGHTS = dict(l1=1.0, pbr=1.0, normal=0.0, light=0.01, env_smooth=0.01, base_color_smooth=0.0, roughness_smooth=0.0,
                      light_smooth=0.0)
STAGE2_WEIGHTS_SYN4 = dict(STAGE2_WEIGHTS, base_color_smooth=1.0, roughness_smooth=0.5, light_smooth=1.0)
# parameter groups the frozen-geometry schedules leave at learning rate 0 (run_syn4.sh:27-33 / run_dtu.sh:29-35)
FROZEN_GEOMETRY_GROUPS = ("xyz", "normal", "scaling", "rotation", "opacity", "shs")


def stage2_smoothness(feat, gt, image_mask, w, maps_are_srgb=False):
    """The three edge-aware terms of calculate_loss (neilf.py:275-292) on the divided feature maps `feat` [16,H,W]:
    base colour and roughness against the target image, diffuse light against the RENDERED NORMAL (not detached: the term
    also pulls on the normal map).  The base-colour and diffuse-light maps the loss sees are results["base_color"] /
    results["diffuse"] = rgb_to_srgb(.) of the rendered maps -- the sRGB curve AND its clip to [0,1] (neilf.py:153-155) --
    the roughness map is used as rendered.  `image_mask` [1,H,W] or None (all ones); `maps_are_srgb`: feat[8:11] and
    feat[12:15] already hold the sRGB-mapped maps (the reference's result dict)."""
    m = 1.0 if image_mask is None else image_mask
    curve = (lambda x: x) if maps_are_srgb else rgb_to_srgb
    loss = feat.new_zeros(())
    if w["base_color_smooth"] != 0.0:
        loss = loss + w["base_color_smooth"] * first_order_edge_aware_loss(curve(feat[8:11]) * m, gt)
    if w["roughness_smooth"] != 0.0:
        loss = loss + w["roughness_smooth"] * first_order_edge_aware_loss(feat[11:12] * m, gt)
    if w["light_smooth"] != 0.0:
        loss = loss + w["light_smooth"] * first_order_edge_aware_loss(curve(feat[12:15]) * m, feat[5:8])
    return loss


# ---- the shading integral in plain PyTorch: what an UNPATCHED train.py runs between the drop-in ops ---------------------------
# The reference's `rendering_equation` (gaussian_renderer/neilf.py:339-371, GGX_specular :374-407, eval_sh utils/sh_utils.py:71-128,
# DirectLightMap.direct_light scene/direct_light_map.py:70-83) is pure PyTorch over [P,K,3] temporaries -- it is NOT one of the
# three extensions this repo replaces, so a user who only swaps the extension packages keeps running it.  This restatement (same
# sequence of elementwise / reduction ops over the same temporaries, autograd for the backward) exists so that bench.py can put
# a number on that mode at the headline size (`other_configs["reference loop shape"]`); the product path is shading_ops.shade.
_SH_C0, _SH_C1 = 0.28209479177387814, 0.4886025119029199
_SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
          1.445305721320277, -0.5900435899266435)


def _eval_sh3(sh, dirs):
    """sh [P,1,3,16] (channels, coefficients), dirs [P,K,3] -> [P,K,3] (degree 3, reference sign convention)."""
    x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
    r = _SH_C0 * sh[..., 0]
    r = r - _SH_C1 * y * sh[..., 1] + _SH_C1 * z * sh[..., 2] - _SH_C1 * x * sh[..., 3]
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    r = (r + _SH_C2[0] * xy * sh[..., 4] + _SH_C2[1] * yz * sh[..., 5] + _SH_C2[2] * (2.0 * zz - xx - yy) * sh[..., 6] +
         _SH_C2[3] * xz * sh[..., 7] + _SH_C2[4] * (xx - yy) * sh[..., 8])
    r = (r + _SH_C3[0] * y * (3 * xx - yy) * sh[..., 9] + _SH_C3[1] * xy * z * sh[..., 10] +
         _SH_C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + _SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12] +
         _SH_C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + _SH_C3[5] * z * (xx - yy) * sh[..., 14] +
         _SH_C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return r


def rendering_equation_pytorch(base_color, roughness, normals, viewdirs, incidents, env, visibility, incident_dirs,
                               incident_areas):
    """-> (pbr [P,3], diffuse_light [P,3], mean visibility [P,1]); env [He,We,3] activated (get_env[0])."""
    import math
    P, K = incident_dirs.shape[:2]
    # DirectLightMap.direct_light: lat-long lookup, bilinear, align_corners=True
    phi = torch.arccos(incident_dirs[..., 2]).reshape(-1) - 1e-6
    theta = torch.atan2(incident_dirs[..., 1], incident_dirs[..., 0]).reshape(-1)
    grid = torch.stack([-theta / math.pi, (phi / math.pi) * 2 - 1], -1).view(1, 1, -1, 2)
    glob = F.grid_sample(env.permute(2, 0, 1)[None], grid, align_corners=True).view(3, -1).t().view(P, K, 3)
    local = _eval_sh3(incidents.transpose(1, 2).view(P, 1, 3, -1), incident_dirs).clamp_min(0)
    lights = local + glob * visibility
    n_d_i = (normals[:, None] * incident_dirs).sum(-1, keepdim=True).clamp(min=0)
    f_d = base_color[:, None] / math.pi
    # GGX_specular
    L = F.normalize(incident_dirs, dim=-1)
    V = F.normalize(viewdirs, dim=-1)
    H = F.normalize((L + V[:, None]) / 2.0, dim=-1)
    N = F.normalize(normals, dim=-1)
    NoV = torch.sum(V * N, dim=-1, keepdim=True)
    N = N * NoV.sign()
    NoL = torch.sum(N[:, None] * L, dim=-1, keepdim=True).clamp(1e-6, 1)
    NoV = torch.sum(N * V, dim=-1, keepdim=True).clamp(1e-6, 1)
    NoH = torch.sum(N[:, None] * H, dim=-1, keepdim=True).clamp(1e-6, 1)
    VoH = torch.sum(V[:, None] * H, dim=-1, keepdim=True).clamp(1e-6, 1)
    alpha = roughness * roughness
    alpha2 = alpha * alpha
    k = (alpha + 2 * roughness + 1.0) / 8.0
    FMi = ((-5.55473) * VoH - 6.98316) * VoH
    frac = (0.04 + 0.96 * torch.pow(2.0, FMi)) * alpha2[:, None]
    nom0 = NoH * NoH * (alpha2[:, None] - 1) + 1
    nom = (4 * math.pi * nom0 * nom0 * (NoV * (1 - k) + k)[:, None] * (NoL * (1 - k[:, None]) + k[:, None])).clamp(1e-6, 4 * math.pi)
    f_s = frac / nom
    transport = lights * incident_areas * n_d_i
    return ((f_d + f_s) * transport).mean(-2), transport.mean(-2), visibility.mean(-2)


class Stage2Step:
    def __init__(self, params, scene, device, sample_num, loss_weights=None, shading="hip"):
        """`loss_weights`: as fused_step.FusedStage2Step (defaults = script/run_nerf.sh:20-39, i.e. the
        normal_render_depth term and the three smoothness terms off; STAGE2_WEIGHTS_SYN4 = run_syn4.sh / run_dtu.sh).
        `shading`: "hip" = shading_ops.shade (INTEGRATION.md's one-line rendering_equation patch), "pytorch" = the reference's
        own pure-PyTorch rendering_equation restated above (what an unpatched train.py keeps running)."""
        self.p = params
        self.shading = shading
        self.K = sample_num
        self.w = dict(STAGE2_WEIGHTS)
        if loss_weights:
            self.w.update(loss_weights)
        with torch.no_grad():
            self.visibility, self.incident_dirs, self.incident_areas, self.tracer = update_visibility(
                params.xyz.detach(), params.get_scaling().detach(), params.get_rotation().detach(),
                params.get_opacity().detach(), params.get_normal().detach(), sample_num)
        self.P = params.xyz.shape[0]

    def render(self, cam, bg):
        p = self.p
        means3D = p.xyz
        means2D = torch.zeros_like(means3D, requires_grad=True)
        base_color = 0.03 + 0.77 * torch.sigmoid(p.base_color)          # gaussian_model.py:51
        roughness = 0.09 + 0.9 * torch.sigmoid(p.roughness)             # gaussian_model.py:52
        normal = p.get_normal()
        incidents = torch.cat([p.incidents_dc, p.incidents_rest], 1)
        viewdirs = F.normalize(cam.camera_center - means3D, dim=-1)
        env = F.softplus(p.env)[0]                                       # DirectLightMap.get_env
        if self.shading == "pytorch":
            pbr, diffuse_light, mean_vis = rendering_equation_pytorch(base_color, roughness, normal.detach(), viewdirs, incidents,
                                                                      env, self.visibility, self.incident_dirs,
                                                                      self.incident_areas)
        else:
            pbr, diffuse_light, rest = shade(base_color, roughness, normal.detach(), viewdirs, incidents, env,
                                             self.visibility, self.incident_dirs, self.incident_areas)
            mean_vis = rest[:, 12:13]
        xyz_h = torch.cat([means3D, torch.ones_like(means3D[:, :1])], -1)
        depths = (xyz_h @ cam.world_view_transform)[:, 2:3]
        features = torch.cat([depths, depths.square(), pbr, normal, base_color, roughness, diffuse_light,
                              mean_vis], -1)                              # S = 16 (neilf.py:120-122)
        rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, cam.tanfovx, cam.tanfovy, cam.cx, cam.cy,
                                           bg, 1.0, cam.world_view_transform, cam.full_proj_transform, 3,
                                           cam.camera_center, False, True, True, False)
        outs = GaussianRasterizer(rs)(means3D, means2D, p.get_opacity(), shs=p.get_shs(), scales=p.get_scaling(),
                                      rotations=p.get_rotation(), features=features)
        return outs, diffuse_light, env

    def __call__(self, cam, bg, gt, image_mask=None):
        outs, diffuse_light, env = self.render(cam, bg)
        num_rendered, n_contrib, image, opacity, depth, feature, pseudo_normal, xyz, weights, radii = outs
        mask = n_contrib > 0
        feat = feature / opacity.clamp_min(1e-5) * mask
        r_depth, r_depth2, r_pbr, r_normal, r_base, r_rough, r_diffuse, r_vis = feat.split([1, 1, 3, 3, 3, 1, 3, 1], 0)
        pbr_img = r_pbr * opacity + (1 - opacity) * bg[:, None, None]
        pbr_srgb = rgb_to_srgb(pbr_img)
        w = self.w
        loss = w["l1"] * image_loss(image, gt) + w["pbr"] * image_loss(pbr_srgb, gt)      # L1/SSIM mix on both images
        if w["normal"] != 0.0:                                                                         # normal_render_depth
            m = 1.0 if image_mask is None else image_mask
            loss = loss + w["normal"] * F.mse_loss(r_normal * m, pseudo_normal.detach() * m)
        loss = loss + stage2_smoothness(feat, gt, image_mask, w)                                       # neilf.py:275-292
        mean_light = diffuse_light.mean(-1, keepdim=True).expand_as(diffuse_light)
        loss = loss + w["light"] * F.l1_loss(diffuse_light, mean_light)                              # lambda_light
        loss = loss + w["env_smooth"] * tv_loss(env.permute(2, 0, 1))                                # lambda_env_smooth
        return loss, outs

    # --- roofline bookkeeping for bench.py (live model fwd (260+16K) B, bwd (476+16K) B per Gaussian)
    def stage_names(self):
        return ("shade_forward", "shade_backward")

    def algorithmic_bytes(self, name):
        per = (260.0 + 16 * self.K) if name == "shade_forward" else (476.0 + 16 * self.K)
        return per * self.P

    def profile(self):
        return {}
