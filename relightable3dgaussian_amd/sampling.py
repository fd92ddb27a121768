"""Ray set of the visibility caches: device-agnostic restatement of the reference's
`fibonacci_sphere_sampling(normals, K, random_rotate=False)` (utils/graphics_utils.py:9-37) and `rotation_between_z`
(utils/sh_utils.py:36-68) -- the reference versions hard-code device="cuda" tensors; these follow the input's device.
Pinned against tests/golden/fibonacci_reference.npz through oracle/shading.py (same formulas)."""
import math

import torch
import torch.nn.functional as F


def rotation_between_z(vec):
    v1, v2 = -vec[..., 1], vec[..., 0]
    cos_p_1 = (vec[..., 2] + 1).clamp_min(1e-7)
    zero = torch.zeros_like(v1)
    R = torch.stack([1 + (-v2 * v2) / cos_p_1, v1 * v2 / cos_p_1, v2 + zero,
                     v1 * v2 / cos_p_1, 1 + (-v1 * v1) / cos_p_1, -v1 + zero,
                     -v2 + zero, v1 + zero, 1 + (-v2 * v2 - v1 * v1) / cos_p_1], -1).reshape(vec.shape[:-1] + (3, 3))
    eye = -torch.eye(3, dtype=vec.dtype, device=vec.device).expand_as(R)
    return torch.where((vec[..., 2] + 1 > 0)[..., None, None], R, eye)


def fibonacci_z_samples(sample_num, device):
    """The Fibonacci set around +z before it is rotated to a normal: [1,3,K] (graphics_utils.py:14-24, random_rotate=False)."""
    delta = math.pi * (3.0 - math.sqrt(5.0))
    idx = torch.arange(sample_num, dtype=torch.float32, device=device)[None]
    z = (1 - 2 * idx / (2 * sample_num - 1)).clamp_min(math.sin(10 / 180 * math.pi))
    rad = torch.sqrt(1 - z ** 2)
    theta = delta * idx
    y, x = torch.cos(theta) * rad, torch.sin(theta) * rad
    return torch.stack([x, y, z.expand_as(y)], dim=-2)


def fibonacci_sphere_sampling(normals, sample_num):
    z_samples = fibonacci_z_samples(sample_num, normals.device)                     # [1,3,K]
    dirs = rotation_between_z(normals) @ z_samples                                  # [P,3,K]
    dirs = F.normalize(dirs, dim=-2).transpose(-1, -2).contiguous()
    areas = torch.ones_like(dirs[..., 0:1]) * 2 * math.pi
    return dirs, areas
