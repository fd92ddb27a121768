"""Drop-in for the reference's compiled `bvh_tracing._C` extension (bvh/src/bindings.cpp:8-12): resolved by
`from bvh_tracing import _C` in bvh/__init__.py:9."""
from relightable3dgaussian_amd.bvh_ops import create_bvh, trace_bvh, trace_bvh_opacity  # noqa: F401
