"""Pure-PyTorch restatement of the rasterizer op -- TEST INFRASTRUCTURE ONLY.

Second, independent restatement of the reference semantics, written with
vectorised torch ops so that (a) autograd supplies the backward that oracle/rasterizer_oracle.c's
hand-restated backward is cross-checked against, and (b) bench.py can time "a pure-PyTorch CPU
rasterize of the same view" next to the GPU number (BASELINE.json north_star).

Reference lines followed: preprocess forward.cu:156-258, binning rasterizer_impl.cu:70-138, blend
forward.cu:334-394, pseudo normal forward.cu:398-491.  Deliberate deviations of the reference's backward
from the true derivative are reproduced so autograd matches it: straight-through min(0.99, .)
(backward.cu:528,594-611), no gradient through the +-1.3*tan_fov clamp (backward.cu:176-177,250-252),
quaternion used un-normalised (backward.cu:342).
"""
import math

import torch

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def sh_to_rgb(deg, sh, dirs):
    """sh [P,M,3], dirs [P,3] (unit) -> [P,3] before the +0.5 / clamp (forward.cu:20-62)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    r = SH_C0 * sh[:, 0]
    if deg > 0:
        r = r - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            r = (r + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5] + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                 + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                r = (r + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                     + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                     + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                     + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                     + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return r


def cov3d_from_scale_rot(scales, mod, rot):
    """forward.cu:119-153: Sigma = R diag(mod*s)^2 R^T with R from the quaternion as given. -> [P,6]"""
    r, x, y, z = rot[:, 0], rot[:, 1], rot[:, 2], rot[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    Mm = R * (mod * scales)[:, None, :]
    Sig = Mm @ Mm.transpose(1, 2)
    return torch.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2], Sig[:, 2, 2]], -1)


def preprocess(means3D, scales, scale_modifier, rotations, opacities, shs, degree, cov3D_precomp, colors_precomp,
               viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy):
    dt = means3D.dtype
    P = means3D.shape[0]
    fx, fy = W / (2.0 * tan_fovx), H / (2.0 * tan_fovy)
    ones = torch.ones(P, 1, dtype=dt)
    hom = torch.cat([means3D, ones], -1)
    p_view = hom @ viewmatrix[:, :3]
    p_hom = hom @ projmatrix
    p_w = 1.0 / (p_hom[:, 3] + 0.0000001)
    p_proj = p_hom[:, :3] * p_w[:, None]
    p_proj_xy = p_proj[:, :2]
    in_front = p_view[:, 2] > 0.2
    cov3D = cov3D_precomp if cov3D_precomp is not None else cov3d_from_scale_rot(scales, scale_modifier, rotations)
    # cov2D (forward.cu:74-113); no gradient through the clamp (backward.cu:176-177)
    tz = p_view[:, 2]
    safe_tz = torch.where(in_front, tz, torch.ones_like(tz))
    limx, limy = 1.3 * tan_fovx, 1.3 * tan_fovy
    txtz, tytz = p_view[:, 0] / safe_tz, p_view[:, 1] / safe_tz
    cl_x = (txtz < -limx) | (txtz > limx)
    cl_y = (tytz < -limy) | (tytz > limy)
    tx = torch.where(cl_x, (txtz.clamp(-limx, limx) * safe_tz).detach(), p_view[:, 0])
    ty = torch.where(cl_y, (tytz.clamp(-limy, limy) * safe_tz).detach(), p_view[:, 1])
    zeros = torch.zeros_like(tz)
    J = torch.stack([fx / safe_tz, zeros, -(fx * tx) / (safe_tz * safe_tz),
                     zeros, fy / safe_tz, -(fy * ty) / (safe_tz * safe_tz)], -1).reshape(P, 2, 3)
    Rw2c = viewmatrix[:3, :3].transpose(0, 1)       # viewmatrix is W2C^T
    A = J @ Rw2c
    Sig = torch.stack([cov3D[:, 0], cov3D[:, 1], cov3D[:, 2], cov3D[:, 1], cov3D[:, 3], cov3D[:, 4],
                       cov3D[:, 2], cov3D[:, 4], cov3D[:, 5]], -1).reshape(P, 3, 3)
    cov = A @ Sig @ A.transpose(1, 2)
    a, b, c = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = in_front & (det != 0)
    det_s = torch.where(ok, det, torch.ones_like(det))
    conic = torch.stack([c / det_s, -b / det_s, a / det_s], -1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    pix = ((p_proj_xy.double() + 1.0) * torch.tensor([W, H], dtype=torch.float64) - 1.0) * 0.5
    pix = pix.to(dt)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    rad_i = radius.clamp(max=2.0e9).to(torch.int64)
    pd = pix.detach()

    def trunc_div(v):
        return torch.trunc(v.to(torch.float32) / 16).clamp(-2.0e9, 2.0e9).to(torch.int64)
    rmin_x = trunc_div(pd[:, 0] - rad_i).clamp(0, gx)
    rmin_y = trunc_div(pd[:, 1] - rad_i).clamp(0, gy)
    rmax_x = trunc_div(pd[:, 0] + rad_i + 15).clamp(0, gx)
    rmax_y = trunc_div(pd[:, 1] + rad_i + 15).clamp(0, gy)
    tiles = (rmax_x - rmin_x) * (rmax_y - rmin_y)
    ok = ok & (tiles > 0)
    if colors_precomp is None:
        d = means3D - campos[None]
        d = d / d.norm(dim=-1, keepdim=True)
        raw = sh_to_rgb(degree, shs, d) + 0.5
        clamped = raw < 0
        rgb = torch.clamp_min(raw, 0.0)
    else:
        rgb, clamped = colors_precomp, torch.zeros(P, 3, dtype=torch.bool)
    radii = torch.where(ok, rad_i, torch.zeros_like(rad_i)).to(torch.int32)
    tiles = torch.where(ok, tiles, torch.zeros_like(tiles))
    return dict(radii=radii, tiles_touched=tiles, means2D=pix, p_proj_xy=p_proj_xy, depths=p_view[:, 2], conic=conic,
                rgb=rgb, clamped=clamped, cov3D=cov3D, rect=(rmin_x, rmin_y, rmax_x, rmax_y), visible=ok)


def bin_tiles(pre, W, H):
    """rasterizer_impl.cu:70-138: (tile<<32 | depth bits) keys, stable ascending sort, per-tile ranges."""
    gx, gy = (W + 15) // 16, (H + 15) // 16
    rmin_x, rmin_y, rmax_x, rmax_y = pre["rect"]
    vis = pre["visible"]
    idx = torch.nonzero(vis).flatten()
    w = (rmax_x - rmin_x)[idx]
    cnt = pre["tiles_touched"][idx]
    R = int(cnt.sum())
    gid = torch.repeat_interleave(idx, cnt)
    start = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(R) - torch.repeat_interleave(start, cnt)
    wrep = torch.repeat_interleave(w, cnt)
    ty = torch.repeat_interleave(rmin_y[idx], cnt) + local // wrep
    tx = torch.repeat_interleave(rmin_x[idx], cnt) + local % wrep
    tile = ty * gx + tx
    dbits = pre["depths"].detach().to(torch.float32).view(torch.int32).to(torch.int64)[gid]
    keys = (tile << 32) | dbits
    order = torch.sort(keys, stable=True).indices
    keys_sorted = keys[order]
    point_list = gid[order]
    tiles_sorted = keys_sorted >> 32
    T = gx * gy
    counts = torch.bincount(tiles_sorted, minlength=T)
    ends = torch.cumsum(counts, 0)
    starts = ends - counts
    ranges = torch.stack([torch.where(counts > 0, starts, torch.zeros_like(starts)),
                          torch.where(counts > 0, ends, torch.zeros_like(ends))], -1)
    return dict(keys_unsorted=keys, vals_unsorted=gid, keys=keys_sorted, point_list=point_list, ranges=ranges,
                num_rendered=R)


def render(pre, binning, colors, features, bg, W, H, backward_geometry=True, chunk=4096):
    dt = pre["means2D"].dtype
    S = features.shape[1]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    out = torch.zeros(5 + S, H, W, dtype=dt)       # color3, opacity, depth, feature S
    final_T = torch.ones(H, W, dtype=dt)
    n_contrib = torch.zeros(H, W, dtype=torch.int32)
    P = pre["means2D"].shape[0]
    weights = torch.zeros(P, dtype=dt)
    opac = pre["opacity"]
    payload = torch.cat([colors, torch.ones(P, 1, dtype=dt), pre["depths"][:, None], features], -1)  # [P,5+S]
    ranges, plist = binning["ranges"], binning["point_list"]
    pieces = []
    for t in range(gx * gy):
        r0, r1 = int(ranges[t, 0]), int(ranges[t, 1])
        ty0, tx0 = (t // gx) * 16, (t % gx) * 16
        th, tw = min(16, H - ty0), min(16, W - tx0)
        ys = torch.arange(ty0, ty0 + th, dtype=dt)[:, None].expand(th, tw).reshape(-1)
        xs = torch.arange(tx0, tx0 + tw, dtype=dt)[None, :].expand(th, tw).reshape(-1)
        npx = ys.numel()
        acc = torch.zeros(npx, 5 + S, dtype=dt)
        Tcur = torch.ones(npx, dtype=dt)
        alive = torch.ones(npx, dtype=torch.bool)
        last = torch.zeros(npx, dtype=torch.int32)
        for c0 in range(r0, r1, chunk):
            g = plist[c0:min(c0 + chunk, r1)]
            n = g.numel()
            dx = pre["means2D"][g, 0][None, :] - xs[:, None]
            dy = pre["means2D"][g, 1][None, :] - ys[:, None]
            con = pre["conic"][g]
            power = -0.5 * (con[:, 0][None] * dx * dx + con[:, 2][None] * dy * dy) - con[:, 1][None] * dx * dy
            a_raw = opac[g][None, :] * torch.exp(torch.clamp_max(power, 0.0))
            alpha = a_raw + (torch.clamp_max(a_raw, 0.99) - a_raw).detach()     # straight-through min(0.99, .)
            valid = (power <= 0) & (alpha.detach() >= 1.0 / 255.0) & alive[:, None]
            a_eff = torch.where(valid, alpha, torch.zeros_like(alpha))
            incl = Tcur[:, None] * torch.cumprod(1 - a_eff, 1)
            excl = torch.cat([Tcur[:, None], incl[:, :-1]], 1)
            stop = valid & (incl.detach() < 0.0001)
            stopped_before = torch.cumsum(stop.to(torch.int32), 1) > 0           # inclusive: the stopper is dropped
            use = valid & ~stopped_before
            wgt = torch.where(use, alpha * excl, torch.zeros_like(alpha))
            pay = payload[g]
            if not backward_geometry and S > 0:
                acc = acc + torch.cat([wgt @ pay[:, :5], wgt.detach() @ pay[:, 5:]], 1)
            else:
                acc = acc + wgt @ pay
            weights = weights.index_add(0, g, wgt.detach().sum(0))
            pos = torch.arange(c0 - r0 + 1, c0 - r0 + n + 1, dtype=torch.int32)[None, :].expand(npx, n)
            lastc = torch.where(use, pos, torch.zeros_like(pos)).max(1).values
            last = torch.maximum(last, lastc)
            n_use = use.to(torch.int64).sum(1)
            # T after the last used entry of this chunk
            any_stop = stopped_before[:, -1]
            idx_last = (torch.where(use, pos, torch.zeros_like(pos)).max(1).indices)
            T_after = torch.where(n_use > 0, incl.gather(1, idx_last[:, None])[:, 0], Tcur)
            Tcur = T_after
            alive = alive & ~any_stop
        pieces.append((ty0, tx0, th, tw, acc, Tcur, last))
    for ty0, tx0, th, tw, acc, Tcur, last in pieces:
        col = acc[:, :3] + Tcur[:, None] * bg[None, :]
        blk = torch.cat([col, acc[:, 3:]], 1).t().reshape(5 + S, th, tw)
        out[:, ty0:ty0 + th, tx0:tx0 + tw] = blk
        final_T[ty0:ty0 + th, tx0:tx0 + tw] = Tcur.detach().reshape(th, tw)
        n_contrib[ty0:ty0 + th, tx0:tx0 + tw] = last.reshape(th, tw)
    return out[:3], out[3:4], out[4:5], out[5:], final_T, n_contrib, weights[:, None]


def pseudo_normal(opacity, depth, viewmatrix, fx, fy, cx, cy):
    """forward.cu:398-491."""
    _, H, W = depth.shape
    dt = depth.dtype
    d = depth[0] / torch.clamp_min(opacity[0], 0.0000001)
    xs = torch.arange(W, dtype=dt)[None, :]
    ys = torch.arange(H, dtype=dt)[:, None]
    xyz = torch.stack([(xs - cx) / fx * d, (ys - cy) / fy * d, d], 0)
    pad = torch.nn.functional.pad(xyz[None], (1, 1, 1, 1), mode="replicate")[0]

    def s(dy, dx):
        return pad[:, 1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    ga = -0.125 * s(-1, -1) + 0.125 * s(-1, 1) - 0.25 * s(0, -1) + 0.25 * s(0, 1) - 0.125 * s(1, -1) + 0.125 * s(1, 1)
    gb = -0.125 * s(-1, -1) - 0.25 * s(-1, 0) - 0.125 * s(-1, 1) + 0.125 * s(1, -1) + 0.25 * s(1, 0) + 0.125 * s(1, 1)
    n = torch.stack([ga[1] * gb[2] - ga[2] * gb[1], -ga[0] * gb[2] + ga[2] * gb[0], ga[0] * gb[1] - ga[1] * gb[0]], 0)
    norm = n.norm(dim=0, keepdim=True)
    nn = torch.where(norm > 0, -n / torch.where(norm > 0, norm, torch.ones_like(norm)), torch.zeros_like(n))
    vm = viewmatrix
    world = torch.stack([vm[0, 0] * nn[0] + vm[0, 1] * nn[1] + vm[0, 2] * nn[2],
                         vm[1, 0] * nn[0] + vm[1, 1] * nn[1] + vm[1, 2] * nn[2],
                         vm[2, 0] * nn[0] + vm[2, 1] * nn[1] + vm[2, 2] * nn[2]], 0)
    return world, xyz


def rasterize(bg, means3D, features, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
              projmatrix, tan_fovx, tan_fovy, cx, cy, H, W, sh, degree, campos, computer_pseudo_normal=True,
              backward_geometry=True):
    """Differentiable (autograd) restatement of `_C.rasterize_gaussians`; all tensors CPU, one dtype."""
    def opt(t):
        return None if (t is None or t.numel() == 0) else t
    colors, scales, rotations, cov3D_precomp, sh = map(opt, (colors, scales, rotations, cov3D_precomp, sh))
    pre = preprocess(means3D, scales, scale_modifier, rotations, opacity, sh, degree, cov3D_precomp, colors,
                     viewmatrix, projmatrix, campos, W, H, tan_fovx, tan_fovy)
    pre["p_proj_xy"].retain_grad() if pre["p_proj_xy"].requires_grad else None
    pre["opacity"] = opacity.reshape(-1)
    binning = bin_tiles(pre, W, H)
    P = means3D.shape[0]
    feats = features.reshape(P, -1)
    color, opac, depth, feat, final_T, n_contrib, weights = render(pre, binning, pre["rgb"], feats, bg, W, H,
                                                                   backward_geometry)
    if computer_pseudo_normal:
        fx, fy = W / (2.0 * tan_fovx), H / (2.0 * tan_fovy)
        normal, xyz = pseudo_normal(opac.detach(), depth.detach(), viewmatrix, fx, fy, cx, cy)
    else:
        normal, xyz = torch.zeros(3, H, W, dtype=color.dtype), torch.zeros(3, H, W, dtype=color.dtype)
    return dict(num_rendered=binning["num_rendered"], n_contrib=n_contrib, color=color, opacity=opac, depth=depth,
                feature=feat, normal=normal, surface_xyz=xyz, weights=weights, radii=pre["radii"], pre=pre,
                binning=binning, final_T=final_T)
