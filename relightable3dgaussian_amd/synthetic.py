"""Seeded synthetic scenes and cameras: no dataset ships with the image, so every
benchmark and parity test renders procedurally generated Gaussian sets from orbit cameras.

Conventions follow the reference's camera code so the tensors can be handed to the ops unchanged:
`world_view_transform` = W2C transposed, `full_proj_transform` = W2C^T @ P^T (scene/cameras.py:62-73,
utils/graphics_utils.py:148-168), z-near 0.01 / z-far 100 (scene/cameras.py:56-57).
"""
import math
from typing import NamedTuple

import numpy as np
import torch


class SynthCamera(NamedTuple):
    image_height: int
    image_width: int
    FoVx: float
    FoVy: float
    tanfovx: float
    tanfovy: float
    cx: float
    cy: float
    world_view_transform: torch.Tensor  # [4,4] = W2C^T
    full_proj_transform: torch.Tensor   # [4,4]
    camera_center: torch.Tensor         # [3]

    def to(self, device):
        return self._replace(world_view_transform=self.world_view_transform.to(device),
                             full_proj_transform=self.full_proj_transform.to(device),
                             camera_center=self.camera_center.to(device))


def _projection(znear, zfar, fovx, fovy):
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    Pm = torch.zeros(4, 4)
    Pm[0, 0] = 2.0 * znear / (2 * right)
    Pm[1, 1] = 2.0 * znear / (2 * top)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


def look_at_camera(eye, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0), width=800, height=800,
                   fovx=0.6911112070083618) -> SynthCamera:
    """OpenCV-style camera (x right, y down, z forward) looking from `eye` at `target`."""
    eye = np.asarray(eye, np.float64)
    fwd = np.asarray(target, np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    Rw2c = np.stack([right, down, fwd], 0)          # rows = camera axes in world
    w2c = np.eye(4)
    w2c[:3, :3] = Rw2c
    w2c[:3, 3] = -Rw2c @ eye
    fovy = 2 * math.atan(height / (2 * (width / (2 * math.tan(fovx / 2)))))
    wvt = torch.tensor(w2c, dtype=torch.float32).transpose(0, 1).contiguous()
    proj = _projection(0.01, 100.0, fovx, fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wvt.inverse()[3, :3].contiguous()
    return SynthCamera(height, width, fovx, fovy, math.tan(fovx * 0.5), math.tan(fovy * 0.5), width / 2.0,
                       height / 2.0, wvt, full, center)


def orbit_cameras(n=100, radius=4.03, width=800, height=800, fovx=0.6911112070083618, elevation_deg=25.0):
    cams = []
    for i in range(n):
        az = 2 * math.pi * i / n
        el = math.radians(elevation_deg) * (0.6 + 0.4 * math.sin(3 * az))
        eye = (radius * math.cos(el) * math.cos(az), radius * math.cos(el) * math.sin(az), radius * math.sin(el))
        cams.append(look_at_camera(eye, width=width, height=height, fovx=fovx))
    return cams


def _normalize(v, eps=1e-12):
    return v / v.norm(dim=-1, keepdim=True).clamp_min(eps)


def make_scene(P=300_000, seed=0, stage2=True, sh_degree=3, scale_log_mean=-4.6, device="cpu"):
    """Lego-like synthetic Gaussian set inside [-1.3,1.3]^3 (scene/dataset_readers.py:294): 70 % of the means
    on the shells of three primitives (sphere, box, torus), 30 % volume noise; flat splats; activated values
    (scales exp'd, rotations normalised, opacities sigmoid'ed) exactly as the reference hands them to the op
    (gaussian_renderer/render.py:52-65)."""
    g = torch.Generator().manual_seed(seed)
    n_shell = int(0.7 * P)
    n_each = [n_shell // 3, n_shell // 3, n_shell - 2 * (n_shell // 3)]
    pts, nrm = [], []
    # sphere r=0.8
    d = _normalize(torch.randn(n_each[0], 3, generator=g))
    pts.append(0.8 * d + torch.tensor([0.0, 0.0, 0.1]))
    nrm.append(d)
    # box half-size (1.0, 0.6, 0.4)
    hs = torch.tensor([1.0, 0.6, 0.4])
    u = torch.rand(n_each[1], 3, generator=g) * 2 - 1
    face = torch.randint(0, 3, (n_each[1],), generator=g)
    sign = (torch.randint(0, 2, (n_each[1],), generator=g) * 2 - 1).float()
    u[torch.arange(n_each[1]), face] = sign
    nb = torch.zeros(n_each[1], 3)
    nb[torch.arange(n_each[1]), face] = sign
    pts.append(u * hs + torch.tensor([0.0, 0.0, -0.5]))
    nrm.append(nb)
    # torus R=0.9 r=0.25 around z
    a = torch.rand(n_each[2], generator=g) * 2 * math.pi
    b = torch.rand(n_each[2], generator=g) * 2 * math.pi
    ring = torch.stack([torch.cos(a), torch.sin(a), torch.zeros_like(a)], -1)
    nt = ring * torch.cos(b)[:, None] + torch.tensor([0.0, 0.0, 1.0]) * torch.sin(b)[:, None]
    pts.append(0.9 * ring + 0.25 * nt + torch.tensor([0.0, 0.0, 0.6]))
    nrm.append(nt)
    n_vol = P - n_shell
    pts.append(torch.rand(n_vol, 3, generator=g) * 2.6 - 1.3)
    nrm.append(_normalize(torch.randn(n_vol, 3, generator=g)))
    xyz = torch.cat(pts).clamp(-1.3, 1.3)
    normal = _normalize(torch.cat(nrm))
    perm = torch.randperm(P, generator=g)            # interleave primitives so index order is not spatial
    xyz, normal = xyz[perm].contiguous(), normal[perm].contiguous()

    log_s = scale_log_mean + 0.5 * torch.randn(P, 3, generator=g)
    flat = torch.randint(0, 3, (P,), generator=g)
    log_s[torch.arange(P), flat] += math.log(0.2)
    scales = torch.exp(log_s)
    rot = _normalize(torch.randn(P, 4, generator=g))
    opacity = torch.sigmoid(1.0 + 2.0 * torch.randn(P, 1, generator=g))
    M = (sh_degree + 1) ** 2
    shs = torch.zeros(P, 16, 3)
    shs[:, 0] = (torch.rand(P, 3, generator=g) - 0.5) / 0.28209479177387814
    shs[:, 1:] = 0.05 * torch.randn(P, 15, 3, generator=g)
    scene = dict(xyz=xyz, normal=normal, scales=scales, rotations=rot, opacity=opacity, shs=shs.contiguous(),
                 sh_degree=sh_degree, M=M)
    if stage2:
        scene["base_color"] = 0.03 + 0.77 * torch.sigmoid(torch.randn(P, 3, generator=g))   # gaussian_model.py:51
        scene["roughness"] = 0.09 + 0.9 * torch.sigmoid(torch.randn(P, 1, generator=g))     # gaussian_model.py:52
        scene["incidents"] = 0.02 * torch.randn(P, 16, 3, generator=g)
        scene["env"] = 0.5 * torch.rand(1, 16, 32, 3, generator=g)                          # direct_light_map.py:14
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in scene.items()}


def make_random_init_scene(P=2000, seed=0, device="cpu"):
    """BASELINE config 1: uniform xyz in the lego bounds, opacity 0.1, isotropic scale from the mean 3-NN
    distance (scene/dataset_readers.py:290-297, scene/gaussian_model.py:427-432)."""
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(P, 3, generator=g) * 2.6 - 1.3
    d2 = torch.cdist(xyz, xyz).square()
    d2.fill_diagonal_(float("inf"))
    nn3 = d2.topk(3, largest=False).values.mean(-1).clamp_min(1e-7)
    scales = torch.sqrt(nn3)[:, None].repeat(1, 3)
    rot = torch.zeros(P, 4)
    rot[:, 0] = 1
    shs = torch.zeros(P, 16, 3)
    shs[:, 0] = (torch.rand(P, 3, generator=g) / 255.0 - 0.5) / 0.28209479177387814
    scene = dict(xyz=xyz, normal=_normalize(torch.randn(P, 3, generator=g)), scales=scales, rotations=rot,
                 opacity=torch.full((P, 1), 0.1), shs=shs, sh_degree=0, M=16)
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in scene.items()}


def write_blender_dataset(root, cameras, images, split="train", name_format="r_%d"):
    """Write views in the NeRF-synthetic (Blender) layout the reference reads (scene/dataset_readers.py:215-270,
    readCamerasFromTransforms): `<root>/transforms_<split>.json` with `camera_angle_x` and one frame per view
    (`file_path` without extension, `transform_matrix` = camera-to-world in the OpenGL/Blender axes: the reader flips the
    Y and Z columns to get back to the OpenCV axes these cameras use) plus `<root>/<split>/<name>.png` (8-bit RGB).
    `cameras`: SynthCamera list sharing one FoVx; `images`: [3,H,W] float tensors in [0,1].  Square images only: the reader
    derives FovY from FovX with the image HEIGHT as the focal-length base (:266), which is the true FovY only when H == W."""
    import json
    import os
    from PIL import Image
    if len(cameras) != len(images) or not cameras:
        raise RuntimeError("write_blender_dataset needs one image per camera")
    fovx = cameras[0].FoVx
    os.makedirs(os.path.join(root, split), exist_ok=True)
    frames = []
    for i, (cam, img) in enumerate(zip(cameras, images)):
        if abs(cam.FoVx - fovx) > 1e-12:
            raise RuntimeError("all views of a Blender-format split share camera_angle_x")
        if cam.image_height != cam.image_width:
            raise RuntimeError("square images only (the reference reader's FovY quirk, dataset_readers.py:266)")
        w2c = cam.world_view_transform.detach().cpu().double().t()            # world_view_transform is W2C transposed
        c2w = torch.linalg.inv(w2c)
        c2w[:3, 1:3] *= -1                                                    # OpenCV -> OpenGL/Blender camera axes
        name = name_format % i
        arr = (img.detach().cpu().clamp(0, 1).permute(1, 2, 0) * 255.0).round().to(torch.uint8).numpy()
        Image.fromarray(arr, "RGB").save(os.path.join(root, split, name + ".png"))
        frames.append({"file_path": "./%s/%s" % (split, name), "transform_matrix": c2w.tolist()})
    with open(os.path.join(root, "transforms_%s.json" % split), "w") as fh:
        json.dump({"camera_angle_x": fovx, "frames": frames}, fh, indent=1)
    return os.path.join(root, "transforms_%s.json" % split)


def cameras_extent(cameras):
    """`scene.cameras_extent` (scene/__init__.py -> getNerfppNorm, scene/dataset_readers.py:45-66): 1.1 x the largest
    distance of a camera centre from the mean camera centre -- the `extent` every densification threshold is scaled by
    (train.py:170).  Returns (radius, translate) with translate = -mean centre."""
    centers = torch.stack([c.camera_center.detach().cpu().double() for c in cameras], 0)
    mean = centers.mean(0)
    radius = float((centers - mean).norm(dim=-1).max()) * 1.1
    return radius, (-mean).numpy()
