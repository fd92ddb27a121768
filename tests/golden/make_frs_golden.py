"""Golden vectors for the fixed-ray-set shading kernels, produced by the REFERENCE's own Python (this container only):

    python tests/golden/make_frs_golden.py        # needs /root/reference; writes tests/golden/shading_reference_frs.npz

tests/golden/shading_reference.npz (make_golden.py) perturbs its directions, so it can only pin the general kernels.  This
fixture is the configuration the training loop actually runs: `fibonacci_sphere_sampling(ray_normal, K, random_rotate=False)`
(utils/graphics_utils.py:9-37 -- what GaussianModel.update_visibility caches, scene/gaussian_model.py:312-342) UNPERTURBED,
fed to the reference's own `rendering_equation` (gaussian_renderer/neilf.py:339-407) with a shading normal that has moved
away from the normal the rays were generated from (the trained normal moves on between visibility updates), a handful of ray
normals next to -z (rotation_between_z loses orthonormality there: those Gaussians leave the rotated path) and smooth
Gaussians (roughness 0.1: the GGX lobe that amplifies any error of a direction).  Inputs, outputs and autograd gradients are
stored; nothing of the reference is copied.  Same import machinery as make_golden.py."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402


def main():
    sys.meta_path.append(mg._Finder())
    sys.path.insert(0, mg.REF)
    for p in mg._cpu_factories():
        p.start()
    import gaussian_renderer.neilf as nf
    from scene.direct_light_map import DirectLightMap
    from utils.graphics_utils import fibonacci_sphere_sampling

    g = torch.Generator().manual_seed(4321)
    rnd = lambda *s: torch.randn(*s, generator=g)
    P, K, He = 83, 24, 16                      # P not a multiple of 16, K not a multiple of 16

    class FakeLight:
        def __init__(self, env):
            self.env = env
        get_env = property(lambda self: torch.nn.functional.softplus(self.env))
        direct_light = DirectLightMap.direct_light

    env_raw = (0.5 * torch.rand(1, He, 2 * He, 3, generator=g)).requires_grad_(True)
    base = (0.03 + 0.77 * torch.sigmoid(rnd(P, 3))).requires_grad_(True)
    rough0 = 0.09 + 0.9 * torch.sigmoid(rnd(P, 1))
    rough0[:12] = 0.1
    rough = rough0.clone().requires_grad_(True)
    ray_normals = torch.nn.functional.normalize(rnd(P, 3), dim=-1)
    special = torch.tensor([[0.0, 0.0, -1.0], [0.0, 3e-3, -1.0], [2e-2, 1e-2, -1.0], [0.0, 0.0, 1.0], [0.08, 0.05, -1.0],
                            [0.0, 0.12, -1.0]])
    ray_normals[:special.shape[0]] = torch.nn.functional.normalize(special, dim=-1)
    # the shading normal: get_normal-like (unit to eps), a few degrees away from the ray normal
    normals = torch.nn.functional.normalize(ray_normals + 0.08 * rnd(P, 3), dim=-1)
    viewdirs = (torch.nn.functional.normalize(rnd(P, 3), dim=-1) * (1 + torch.rand(P, 1, generator=g))).requires_grad_(True)
    incidents = (0.3 * rnd(P, 16, 3)).requires_grad_(True)
    dirs, areas = fibonacci_sphere_sampling(ray_normals, K, random_rotate=False)
    vis = torch.rand(P, K, 1, generator=g)
    vis = torch.where(vis < 0.3, torch.zeros_like(vis), 0.9 + 0.1 * vis)
    pbr, extra = nf.rendering_equation(base, rough, normals, viewdirs, incidents, FakeLight(env_raw),
                                       visibility_precompute=vis, incident_dirs_precompute=dirs,
                                       incident_areas_precompute=areas)
    g_pbr, g_diff = rnd(P, 3), rnd(P, 3)
    ((pbr * g_pbr).sum() + (extra["diffuse_light"] * g_diff).sum()).backward()
    np.savez(os.path.join(HERE, "shading_reference_frs.npz"),
             base_color=base.detach().numpy(), roughness=rough.detach().numpy(), normals=normals.numpy(),
             ray_normals=ray_normals.numpy(), viewdirs=viewdirs.detach().numpy(), incidents=incidents.detach().numpy(),
             env_raw=env_raw.detach().numpy(), visibility=vis.numpy(), incident_dirs=dirs.numpy(),
             incident_areas=areas.numpy(), pbr=pbr.detach().numpy(), diffuse_light=extra["diffuse_light"].detach().numpy(),
             incident_visibility_mean=extra["incident_visibility"].detach().mean(-2).numpy(),
             g_pbr=g_pbr.numpy(), g_diffuse_light=g_diff.numpy(), d_base_color=base.grad.numpy(),
             d_roughness=rough.grad.numpy(), d_viewdirs=viewdirs.grad.numpy(), d_incidents=incidents.grad.numpy(),
             d_env_raw=env_raw.grad.numpy())
    print("wrote shading_reference_frs.npz: P=%d K=%d, pbr max %.4f" % (P, K, float(pbr.abs().max())))


if __name__ == "__main__":
    main()
