"""CPU oracle for the BVH ops -- TEST INFRASTRUCTURE ONLY (Python face of oracle/bvh_oracle.c).

`leaf_boxes` restates the Python half of the reference's RayTracer.__init__ (bvh/__init__.py:28-57) and is pinned by
tests/golden/bvh_leaf_reference.npz; `create_bvh` / `trace_bvh_opacity` mirror `bvh_tracing._C` (bvh/include/bvh.h:5-18)
on numpy arrays.  The product package never imports this module."""
import ctypes as C

import numpy as np

from . import _build

_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_build.build())
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def build_rotation(q):
    """utils/general_utils.py:82-103 (normalises q)."""
    norm = np.sqrt(q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1] + q[:, 2] * q[:, 2] + q[:, 3] * q[:, 3])
    q = q / norm[:, None]
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.zeros((q.shape[0], 3, 3), np.float32)
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - r * z)
    R[:, 0, 2] = 2 * (x * z + r * y)
    R[:, 1, 0] = 2 * (x * y + r * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - r * x)
    R[:, 2, 0] = 2 * (x * z - r * y)
    R[:, 2, 1] = 2 * (y * z + r * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def leaf_boxes(means3D, scales, rotations):
    """-> (nodes int32[2P-1,5], aabbs float32[2P-1,6]) exactly as RayTracer.__init__ prepares them."""
    means3D, scales, rotations = (np.asarray(a, np.float32) for a in (means3D, scales, rotations))
    P = means3D.shape[0]
    rot = build_rotation(rotations)
    nodes = np.full((2 * P - 1, 5), -1, np.int32)
    nodes[:P - 1, 4] = 0
    nodes[P - 1:, 4] = 1
    aabbs = np.zeros((2 * P - 1, 6), np.float32)
    aabbs[:, :3] = 100000
    aabbs[:, 3:] = -100000
    a, b, c = rot[:, :, 0], rot[:, :, 1], rot[:, :, 2]
    sa, sb, sc = (np.float32(3) * scales[:, i] for i in range(3))
    corners = []
    for s0 in (1, -1):
        for s1 in (1, -1):
            for s2 in (1, -1):
                x = means3D + np.float32(s0) * a * sa[:, None]
                x = x + np.float32(s1) * b * sb[:, None]
                x = x + np.float32(s2) * c * sc[:, None]
                corners.append(x)
    corners = np.stack(corners, 0)
    aabbs[P - 1:] = np.concatenate([corners.min(0), corners.max(0)], -1)
    return nodes, aabbs


def create_bvh(nodes, aabbs):
    P = (nodes.shape[0] + 1) // 2
    nodes = np.ascontiguousarray(nodes, np.int32).copy()
    aabbs = np.ascontiguousarray(aabbs, np.float32).copy()
    morton = np.zeros(P, np.uint64)
    lib().bvho_build(P, _p(nodes), _p(aabbs), _p(morton))
    return nodes, aabbs, morton


def _ray_args(rays_o, rays_d, means3D, covs, opac, normals):
    o = np.ascontiguousarray(rays_o, np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(rays_d, np.float32).reshape(-1, 3)
    return (o, d, np.ascontiguousarray(means3D, np.float32), np.ascontiguousarray(covs, np.float32),
            np.ascontiguousarray(opac, np.float32).reshape(-1), np.ascontiguousarray(normals, np.float32))


def trace_bvh_opacity(nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities, normals):
    shape = np.asarray(rays_o).shape[:-1]
    o, d, mu, ci, op, n = _ray_args(rays_o, rays_d, means3D, covs3D, opacities, normals)
    nr, P = o.shape[0], mu.shape[0]
    cnt = np.zeros(nr, np.int32)
    out = np.ones(nr, np.float32)
    lib().bvho_trace_opacity(nr, P, _p(np.ascontiguousarray(nodes, np.int32)), _p(np.ascontiguousarray(aabbs, np.float32)),
                             _p(o), _p(d), _p(mu), _p(ci), _p(op), _p(n), _p(cnt), _p(out))
    return cnt.reshape(shape), out.reshape(shape)


def trace_bruteforce(nodes, aabbs, rays_o, rays_d, means3D, covs3D, opacities, normals):
    """-> (count, product as float64) without the tree and without the 0.9 early-out."""
    shape = np.asarray(rays_o).shape[:-1]
    o, d, mu, ci, op, n = _ray_args(rays_o, rays_d, means3D, covs3D, opacities, normals)
    nr, P = o.shape[0], mu.shape[0]
    cnt = np.zeros(nr, np.int32)
    prod = np.ones(nr, np.float64)
    lib().bvho_trace_bruteforce(nr, P, _p(np.ascontiguousarray(nodes, np.int32)),
                                _p(np.ascontiguousarray(aabbs, np.float32)), _p(o), _p(d), _p(mu), _p(ci), _p(op), _p(n),
                                _p(cnt), _p(prod))
    return cnt.reshape(shape), prod.reshape(shape)
