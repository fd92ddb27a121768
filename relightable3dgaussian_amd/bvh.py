"""`RayTracer` with the interface of the reference's bvh/__init__.py:28-71: builds the LBVH over the 3-sigma oriented
boxes of the Gaussians and traces visibility rays.  Leaf-box construction follows bvh/__init__.py:28-57 (pinned by
tests/golden/bvh_leaf_reference.npz); build and trace run the HIP kernels through `bvh_ops`."""
import torch

from . import _lib, bvh_ops


def build_rotation(r):
    """utils/general_utils.py:82-103"""
    norm = torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)
    return R.reshape(-1, 3, 3)


def leaf_boxes(means3D, scales, rotations):
    """-> (nodes int32[2P-1,5], aabbs float32[2P-1,6]) as RayTracer.__init__ prepares them (bvh/__init__.py:29-57)."""
    P = means3D.shape[0]
    dev = means3D.device
    rot = build_rotation(rotations)
    nodes = torch.full((2 * P - 1, 5), -1, dtype=torch.int32, device=dev)
    nodes[:P - 1, 4] = 0
    nodes[P - 1:, 4] = 1
    aabbs = torch.zeros(2 * P - 1, 6, dtype=torch.float32, device=dev)
    aabbs[:, :3] = 100000
    aabbs[:, 3:] = -100000
    a, b, c = rot[:, :, 0], rot[:, :, 1], rot[:, :, 2]
    sa, sb, sc = 3 * scales[:, 0:1], 3 * scales[:, 1:2], 3 * scales[:, 2:3]
    lo, hi = None, None
    for s0 in (1.0, -1.0):
        for s1 in (1.0, -1.0):
            for s2 in (1.0, -1.0):
                x = means3D + s0 * a * sa + s1 * b * sb + s2 * c * sc
                lo = x if lo is None else torch.minimum(lo, x)
                hi = x if hi is None else torch.maximum(hi, x)
    aabbs[P - 1:] = torch.cat([lo, hi], dim=-1)
    return nodes, aabbs


class RayTracer:
    def __init__(self, means3D, scales, rotations):
        nodes, aabbs = leaf_boxes(means3D, scales, rotations)
        self.tree, self.aabb, self.morton = bvh_ops.create_bvh(means3D, scales, rotations, nodes, aabbs)
        # packed traversal records of the arrays the LAST trace_visibility call was given (update_visibility traces its
        # ray bundles chunk by chunk against the same arrays: packed once, owned by this tracer)
        self._records = self._records_key = self._records_ref = None

    def _records_for(self, means3D, symm_inv, opacity, normals):
        if self.tree.shape[0] != 2 * means3D.shape[0] - 1 or means3D.shape[0] == 0:
            return None
        if _lib.get_option("TRACE_FORMULATION") < 2:         # (experiments: the formulations that walk the reference's tables)
            return None
        arrays = (means3D, symm_inv, opacity, normals)
        key = tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in arrays)
        if self._records is None or key != self._records_key:
            self._records = bvh_ops.trace_records(self.tree, self.aabb, *arrays)
            # (references kept: an address cannot be recycled for other values while it is part of the key)
            self._records_key, self._records_ref = key, arrays
        return self._records

    @torch.no_grad()
    def trace_visibility(self, rays_o, rays_d, means3D, symm_inv, opacity, normals):
        rays_o = rays_o + rays_d * 0.05
        contrib, opa = bvh_ops.trace_bvh_opacity(self.tree, self.aabb, rays_o, rays_d, means3D, symm_inv, opacity,
                                                 normals, records=self._records_for(means3D, symm_inv, opacity, normals))
        return {"visibility": opa.unsqueeze(-1), "contribute": contrib.unsqueeze(-1)}
