"""Package name the reference's bvh/__init__.py imports (`from bvh_tracing import _C`).  Also re-exports the
RayTracer class with the reference's interface (bvh/__init__.py:28-71)."""
from . import _C  # noqa: F401
from relightable3dgaussian_amd.bvh import RayTracer  # noqa: F401
