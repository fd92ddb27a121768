"""Drop-in for the reference's compiled `r3dg_rasterization._C` extension (r3dg-rasterization/ext.cpp:15-19):
`from r3dg_rasterization import _C` resolves here, and `_C.rasterize_gaussians`, `_C.rasterize_gaussians_backward`,
`_C.mark_visible` run the MI355X HIP kernels through the C ABI.  The render-equation ops named by
render_equation.h (never bound in the reference snapshot) are exposed under the same module."""
from relightable3dgaussian_amd.rasterizer_ops import (mark_visible, rasterize_gaussians,  # noqa: F401
                                                      rasterize_gaussians_backward)

from relightable3dgaussian_amd.shading_ops import (render_equation_backward, render_equation_forward,  # noqa: F401
                                                   render_equation_forward_complex, rendering_equation, shade)
