"""Autograd face of the rasterizer op: the same public surface as the reference's live wrapper
(gaussian_renderer/r3dg_rasterization.py:58-261) -- `GaussianRasterizationSettings` (16 fields, same order),
`GaussianRasterizer(raster_settings)(means3D, means2D, opacities, shs=, colors_precomp=, scales=, rotations=,
cov3D_precomp=, features=)` returning the 10-tuple (num_rendered, num_contrib, color, opacity, depth, feature,
normal, surface_xyz, weights, radii), gradients flowing to (means3D, means2D, features, sh, colors_precomp,
opacities, scales, rotations, cov3Ds_precomp).  The reference's own wrapper file also works unchanged against
`r3dg_rasterization._C`; this module exists so the package is usable without the reference tree.
"""
from typing import NamedTuple

import torch
import torch.nn as nn

from . import rasterizer_ops as _ops


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    cx: float
    cy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    backward_geometry: bool
    computer_pseudo_normal: bool
    debug: bool


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, features, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                rs):
        (num_rendered, num_contrib, color, opacity, depth, feature, normal, surface_xyz, weights, radii, geom, binning,
         img) = _ops.rasterize_gaussians(
            rs.bg, means3D, features, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.cx, rs.cy, rs.image_height, rs.image_width, sh,
            rs.sh_degree, rs.campos, rs.prefiltered, rs.computer_pseudo_normal, rs.debug)
        ctx.rs = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, features, scales, rotations, cov3Ds_precomp, radii, sh, geom,
                              binning, img)
        ctx.mark_non_differentiable(num_contrib, normal, surface_xyz, weights, radii)
        return num_rendered, num_contrib, color, opacity, depth, feature, normal, surface_xyz, weights, radii

    @staticmethod
    def backward(ctx, _g_num, _g_contrib, g_color, g_opacity, g_depth, g_feature, *_unused):
        rs = ctx.rs
        colors_precomp, means3D, features, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img = \
            ctx.saved_tensors

        def z(g, like_shape):
            return g if g is not None else torch.zeros(like_shape, dtype=torch.float32, device=means3D.device)
        H, W = rs.image_height, rs.image_width
        (g_means2D, g_colors, g_opac, g_means3D, g_feat, g_cov3D, g_sh, g_scales, g_rot) = \
            _ops.rasterize_gaussians_backward(
                rs.bg, means3D, features, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
                rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, z(g_color, (3, H, W)), z(g_opacity, (1, H, W)),
                z(g_depth, (1, H, W)), z(g_feature, (features.shape[1], H, W)), sh, rs.sh_degree, rs.campos, geom,
                ctx.num_rendered, binning, img, rs.backward_geometry, rs.debug)
        return g_means3D, g_means2D, g_feat, g_sh, g_colors, g_opac, g_scales, g_rot, g_cov3D, None


def rasterize_gaussians(means3D, means2D, features, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _Rasterize.apply(means3D, means2D, features, sh, colors_precomp, opacities, scales, rotations,
                            cov3Ds_precomp, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _ops.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, features=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        empty = torch.Tensor([])       # absent optionals travel as empty CPU tensors, like the reference's
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        if features is None:
            features = torch.empty_like(means3D[..., :0])
        return rasterize_gaussians(means3D, means2D, features, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings)
