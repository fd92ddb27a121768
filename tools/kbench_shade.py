"""Time the shading kernels alone at full size (GPU only).  ROWS=0 selects the round-1 16-lane forward kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightable3dgaussian_amd import _lib, shading_ops as so, sampling
P = int(os.environ.get("P", 300000)); dev = "cuda"
L = _lib.lib()
for name in _lib.OPTIONS:
    if os.environ.get("R3DG_OPT_" + name):
        _lib.set_option(name, int(os.environ["R3DG_OPT_" + name]))
g = torch.Generator().manual_seed(0)
CASES = ((64, 16),) if os.environ.get("ONLY64") else ((64, 16), (384, 256))
for K, He in CASES:
    nrm = torch.nn.functional.normalize(torch.randn(P, 3, generator=g), dim=-1).to(dev)
    dirs, areas = sampling.fibonacci_sphere_sampling(nrm, K)
    vis = (torch.rand(P, K, 1, device=dev) > 0.3).float()
    base = torch.rand(P, 3, device=dev); rough = 0.1 + 0.8 * torch.rand(P, 1, device=dev)
    view = torch.nn.functional.normalize(torch.randn(P, 3, device=dev), dim=-1)
    inc = 0.1 * torch.randn(P, 16, 3, device=dev)
    env = torch.rand(He, 2 * He, 3, device=dev)
    gp, gd = torch.randn(P, 3, device=dev), torch.randn(P, 3, device=dev)
    taps = so.build_taps(dirs, He, 2 * He)
    rad = so.build_taps(dirs, He, 2 * He, radiance_of=env)
    res = {}
    for name, kw in (("forward (all 19 outputs, lookup in kernel)", {}), ("forward (19 outputs, cached taps)", dict(taps=taps)),
                     ("forward (train outputs, cached taps)", dict(taps=taps, train_outputs=True)),
                     ("forward (train outputs, cached taps, uniform area)", dict(taps=taps, train_outputs=True, uniform_area=6.283185307179586)),
                     ("forward (19 outputs, cached radiance, uniform area)", dict(taps=rad, taps_are_radiance=True, uniform_area=6.283185307179586))):
        for it in range(8):
            if it == 3:
                torch.cuda.synchronize(); L.r3dg_profile_enable(1)
            so.shade_forward(base, rough, nrm, view, inc, env, vis, dirs, areas, **kw)
        torch.cuda.synchronize()
        pr = _lib.profile_read(); L.r3dg_profile_enable(0)
        res[name] = pr["shade_forward"][0] / max(pr["shade_forward"][1], 1)
    if K == 64:
        for it in range(8):
            if it == 3:
                torch.cuda.synchronize(); L.r3dg_profile_enable(1)
            so.shade_backward(base, rough, nrm, view, inc, env, vis, dirs, areas, gp, gd)
        torch.cuda.synchronize()
        pr = _lib.profile_read(); L.r3dg_profile_enable(0)
        res["backward"] = pr["shade_backward"][0] / max(pr["shade_backward"][1], 1)
        for it in range(8):
            if it == 3:
                torch.cuda.synchronize(); L.r3dg_profile_enable(1)
            so.shade_backward(base, rough, nrm, view, inc, env, vis, dirs, areas, gp, gd, taps=taps)
        torch.cuda.synchronize()
        pr = _lib.profile_read(); L.r3dg_profile_enable(0)
        res["backward (cached taps)"] = pr["shade_backward"][0] / max(pr["shade_backward"][1], 1)
    # the fixed-ray-set kernels on the same caches (csrc/shading_frs.hpp): coefficient rotation + MFMA kernel (+ the
    # wave-per-Gaussian
    # kernels on the few Gaussians off the rotated path) in one profiled stage each
    if so.FixedRaySet.supported(K, 16, He, 2 * He):
        frs = so.FixedRaySet.try_build(nrm, dirs)
        out = torch.empty(P, so.NOUT, device=dev)
        args = (base, rough, nrm, view, inc, env, vis)
        for it in range(8):
            if it == 3:
                torch.cuda.synchronize(); L.r3dg_profile_enable(1)
            frs.forward(*args, out, uniform_area=6.283185307179586)
        torch.cuda.synchronize()
        pr = _lib.profile_read(); L.r3dg_profile_enable(0)
        res["frs forward (train outputs; %d Gaussians off the rotated path)" % frs.n_invalid] = pr["shade_forward"][0] / max(pr["shade_forward"][1], 1)
        res["frs forward listed"] = pr["shade_frs_listed"][0] / max(pr["shade_frs_listed"][1], 1)
        if K == 64:
            for it in range(8):
                if it == 3:
                    torch.cuda.synchronize(); L.r3dg_profile_enable(1)
                frs.backward(*args, gp, gd, uniform_area=6.283185307179586)
            torch.cuda.synchronize()
            pr = _lib.profile_read(); L.r3dg_profile_enable(0)
            res["frs backward"] = pr["shade_backward"][0] / max(pr["shade_backward"][1], 1)
            res["frs backward listed"] = pr["shade_frs_listed"][0] / max(pr["shade_frs_listed"][1], 1)
            # the incident-light chain as one kernel (rotation back + Adam + rotation forward), alone
            grad, m, v = torch.zeros_like(inc), torch.zeros_like(inc), torch.zeros_like(inc)
            for it in range(8):
                if it == 3:
                    torch.cuda.synchronize(); L.r3dg_profile_enable(1)
                frs.incident_chain(inc, grad, m, v, 1e-4, 1e-5, (0.9, 0.999), 1e-15, it + 1)
            torch.cuda.synchronize()
            pr = _lib.profile_read(); L.r3dg_profile_enable(0)
            res["frs incident chain"] = pr["shade_frs_aux"][0] / max(pr["shade_frs_aux"][1], 1)
    print("K=%d He=%d  " % (K, He) + "  ".join("%s %.4f ms" % kv for kv in res.items()))
