"""Package name the reference imports (`from r3dg_rasterization import _C`,
gaussian_renderer/r3dg_rasterization.py:7-8).  Also re-exports the autograd wrapper API of the live wrapper
(GaussianRasterizationSettings / GaussianRasterizer / rasterize_gaussians, same fields and call signature)."""
from . import _C  # noqa: F401
from relightable3dgaussian_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                                  rasterize_gaussians)
