"""Generate golden vectors by running the REFERENCE's own Python code on CPU (this container only).

    python tests/golden/make_golden.py        # needs /root/reference; writes tests/golden/*.npz

The reference's CUDA extensions cannot be built here, but the pure-PyTorch parts of its hot path can be imported
and executed: `rendering_equation` / `GGX_specular` (gaussian_renderer/neilf.py:339-407), `DirectLightMap.direct_light`
(scene/direct_light_map.py:70-83), `EnvLight.direct_light` (scene/envmap.py:35-53), `eval_sh` (utils/sh_utils.py:71-128),
`fibonacci_sphere_sampling` / `rotation_between_z` (utils/graphics_utils.py:9-37, utils/sh_utils.py:36-68),
`build_scaling_rotation` + `strip_symmetric` (utils/general_utils.py) and the leaf-AABB construction of
`RayTracer.__init__` (bvh/__init__.py:28-57).  Missing third-party imports (torchvision, kornia, plyfile, nvdiffrast,
pyexr, imageio, cv2, ...) and the compiled extensions are replaced by auto-mocks; torch factory functions are patched
to ignore the hard-coded device="cuda".  Nothing from the reference is copied: only inputs and outputs are stored.
The fixtures travel to the GPU box; /root/reference does not.
"""
import importlib.abc
import importlib.machinery
import os
import sys
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
MOCK_ROOTS = {"torchvision", "kornia", "plyfile", "nvdiffrast", "pyexr", "imageio", "cv2", "tensorboard", "simple_knn",
              "bvh_tracing", "r3dg_rasterization", "dearpygui", "open3d", "trimesh", "matplotlib"}


class _Loader(importlib.abc.Loader):
    def create_module(self, spec):
        m = mock.MagicMock()
        m.__path__ = []
        m.__spec__ = spec
        m.__name__ = spec.name
        return m

    def exec_module(self, module):
        pass


class _Finder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in MOCK_ROOTS:
            return importlib.machinery.ModuleSpec(name, _Loader(), is_package=True)
        return None


def _cpu_factories():
    """torch.zeros(..., device='cuda') etc. -> CPU."""
    patches = []
    for fn in ("zeros", "ones", "arange", "eye", "rand", "full", "tensor", "empty", "zeros_like", "ones_like"):
        orig = getattr(torch, fn)

        def wrapped(*a, __orig=orig, **kw):
            kw.pop("device", None)
            return __orig(*a, **kw)
        patches.append(mock.patch.object(torch, fn, wrapped))
    patches.append(mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self))
    return patches


def main():
    sys.meta_path.append(_Finder())
    sys.path.insert(0, REF)
    for p in _cpu_factories():
        p.start()
    import gaussian_renderer.neilf as nf
    from scene.direct_light_map import DirectLightMap
    from scene.envmap import EnvLight
    from utils.general_utils import build_scaling_rotation, strip_symmetric
    from utils.graphics_utils import fibonacci_sphere_sampling
    from utils.sh_utils import eval_sh

    g = torch.Generator().manual_seed(1234)

    def rnd(*s):
        return torch.randn(*s, generator=g)

    # ---------------- shading (live GGX model) ----------------
    P, K, He = 48, 24, 16

    class FakeLight:
        def __init__(self, env):
            self.env = env
        get_env = property(lambda self: torch.nn.functional.softplus(self.env))
        direct_light = DirectLightMap.direct_light

    env_raw = (0.5 * torch.rand(1, He, 2 * He, 3, generator=g)).requires_grad_(True)
    base = (0.03 + 0.77 * torch.sigmoid(rnd(P, 3))).requires_grad_(True)
    rough = (0.09 + 0.9 * torch.sigmoid(rnd(P, 1))).requires_grad_(True)
    normals = torch.nn.functional.normalize(rnd(P, 3), dim=-1)
    viewdirs = torch.nn.functional.normalize(rnd(P, 3), dim=-1).requires_grad_(True)
    viewdirs_raw = viewdirs
    incidents = (0.3 * rnd(P, 16, 3)).requires_grad_(True)
    dirs, areas = fibonacci_sphere_sampling(normals, K, random_rotate=False)
    # a few grazing / back-facing samples so the clamps are exercised
    dirs = torch.nn.functional.normalize(dirs + 0.35 * rnd(P, K, 3), dim=-1)
    vis = torch.rand(P, K, 1, generator=g)
    vis = torch.where(vis < 0.3, torch.zeros_like(vis), 0.9 + 0.1 * vis)
    pbr, extra = nf.rendering_equation(base, rough, normals, viewdirs_raw, incidents, FakeLight(env_raw),
                                       visibility_precompute=vis, incident_dirs_precompute=dirs,
                                       incident_areas_precompute=areas)
    g_pbr, g_diff = rnd(P, 3), rnd(P, 3)
    loss = (pbr * g_pbr).sum() + (extra["diffuse_light"] * g_diff).sum()
    loss.backward()
    np.savez(os.path.join(HERE, "shading_reference.npz"),
             base_color=base.detach().numpy(), roughness=rough.detach().numpy(), normals=normals.numpy(),
             viewdirs=viewdirs.detach().numpy(), incidents=incidents.detach().numpy(), env_raw=env_raw.detach().numpy(),
             visibility=vis.numpy(), incident_dirs=dirs.numpy(), incident_areas=areas.numpy(),
             pbr=pbr.detach().numpy(), diffuse_light=extra["diffuse_light"].detach().numpy(),
             specular=extra["specular"].detach().numpy(),
             incident_lights_mean=extra["incident_lights"].detach().mean(-2).numpy(),
             local_incident_lights_mean=extra["local_incident_lights"].detach().mean(-2).numpy(),
             global_incident_lights_mean=extra["global_incident_lights"].detach().mean(-2).numpy(),
             incident_visibility_mean=extra["incident_visibility"].mean(-2).numpy(),
             g_pbr=g_pbr.numpy(), g_diffuse_light=g_diff.numpy(),
             d_base_color=base.grad.numpy(), d_roughness=rough.grad.numpy(), d_viewdirs=viewdirs.grad.numpy(),
             d_incidents=incidents.grad.numpy(), d_env_raw=env_raw.grad.numpy())

    # relight variant: fixed HDR map + 3x3 light rotation (EnvLight.direct_light)
    hdr = 3.0 * torch.rand(32, 64, 3, generator=g) ** 2
    tr = torch.linalg.qr(rnd(3, 3)).Q
    fake = mock.MagicMock()
    fake.envmap, fake.transform = hdr, None
    light_hdr = EnvLight.direct_light(fake, dirs, transform=tr)
    np.savez(os.path.join(HERE, "envlight_reference.npz"), envmap=hdr.numpy(), transform=tr.numpy(), dirs=dirs.numpy(),
             light=light_hdr.numpy())

    # ---------------- SH evaluation, covariance, ray set ----------------
    sh = rnd(200, 3, 16)
    d = torch.nn.functional.normalize(rnd(200, 3), dim=-1)
    sh_out = {"deg%d" % k: eval_sh(k, sh, d).numpy() for k in range(4)}
    np.savez(os.path.join(HERE, "eval_sh_reference.npz"), sh=sh.numpy(), dirs=d.numpy(), **sh_out)

    s = torch.exp(-3 + 0.5 * rnd(200, 3))
    q = torch.nn.functional.normalize(rnd(200, 4), dim=-1)
    Lm = build_scaling_rotation(1.7 * s, q)
    cov = strip_symmetric(Lm @ Lm.transpose(1, 2))
    Li = build_scaling_rotation(1 / s, q)
    cov_inv = strip_symmetric(Li @ Li.transpose(1, 2))
    np.savez(os.path.join(HERE, "covariance_reference.npz"), scales=s.numpy(), rotations=q.numpy(), modifier=1.7,
             cov3D=cov.numpy(), cov3D_inverse=cov_inv.numpy())

    nrm = torch.nn.functional.normalize(rnd(64, 3), dim=-1)
    nrm[0] = torch.tensor([0.0, 0.0, -1.0])       # the n_z + 1 <= 0 branch of rotation_between_z
    nrm[1] = torch.tensor([0.0, 0.0, 1.0])
    fd, fa = fibonacci_sphere_sampling(nrm, 64, random_rotate=False)
    np.savez(os.path.join(HERE, "fibonacci_reference.npz"), normals=nrm.numpy(), dirs=fd.numpy(), areas=fa.numpy())

    # ---------------- BVH leaf boxes (Python part of RayTracer.__init__) ----------------
    import bvh as ref_bvh
    captured = {}

    def fake_create_bvh(means3D, scales, rotations, nodes, aabbs):
        captured["nodes"], captured["aabbs"] = nodes.clone(), aabbs.clone()
        return nodes, aabbs, None
    ref_bvh._C.create_bvh = fake_create_bvh
    mu = rnd(100, 3)
    ref_bvh.RayTracer(mu, s[:100], q[:100])
    np.savez(os.path.join(HERE, "bvh_leaf_reference.npz"), means3D=mu.numpy(), scales=s[:100].numpy(),
             rotations=q[:100].numpy(), nodes_init=captured["nodes"].numpy(), aabbs_init=captured["aabbs"].numpy())
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
