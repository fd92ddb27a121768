"""Shared helpers for the parity tests (oracle <-> HIP)."""
import numpy as np
import torch

from relightable3dgaussian_amd import synthetic as syn


def to_np(t):
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().numpy()
    return np.asarray(t)


def report(name, got, ref, rtol, atol):
    """Returns (ok, message): |got-ref| <= atol + rtol*max|ref| elementwise (scale-relative tolerance: sums of
    float atomics are compared against the double-accumulated oracle, so the natural scale is the array's)."""
    got = to_np(got).astype(np.float64)
    ref = to_np(ref).astype(np.float64)
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (name, got.shape, ref.shape)
    if ref.size == 0:
        return True, "%s: empty" % name
    scale = np.abs(ref).max()
    err = np.abs(got - ref)
    bound = atol + rtol * scale
    bad = err > bound
    msg = "%-14s max|err| %.3e  scale %.3e  rel %.2e  bad %d/%d (bound %.2e)" % (
        name, err.max(), scale, err.max() / max(scale, 1e-30), bad.sum(), ref.size, bound)
    if not np.isfinite(got).all():
        return False, msg + "  NON-FINITE values in result"
    return not bad.any(), msg


def make_case(P=3000, W=128, H=128, S=5, seed=1, scale_log_mean=-3.0, eye=(3.2, 1.0, 1.5), use_colors=False,
              use_cov=False, bg=(1.0, 0.5, 0.2), sh_degree=3):
    """A seeded scene + camera + feature block; everything as CPU fp32 torch tensors (dict)."""
    sc = syn.make_scene(P=P, seed=seed, scale_log_mean=scale_log_mean)
    cam = syn.look_at_camera(eye, width=W, height=H)
    g = torch.Generator().manual_seed(seed + 100)
    feat = torch.rand(P, S, generator=g) if S > 0 else torch.zeros(P, 0)
    case = dict(P=P, W=W, H=H, S=S, bg=torch.tensor(bg, dtype=torch.float32), means3D=sc["xyz"], features=feat,
                opacity=sc["opacity"], scales=sc["scales"], rotations=sc["rotations"], shs=sc["shs"],
                degree=sh_degree, cam=cam, colors=None, cov3D=None)
    if use_colors:
        case["colors"] = torch.rand(P, 3, generator=g)
        case["shs"] = None
    if use_cov:
        from oracle import torch_rasterizer as trz
        case["cov3D"] = trz.cov3d_from_scale_rot(sc["scales"], 1.0, sc["rotations"]).contiguous()
        case["scales"] = None
        case["rotations"] = None
    return case


def case_from_scene(sc, W=128, H=128, S=5, seed=1, eye=(3.2, 1.0, 1.5), bg=(1.0, 0.5, 0.2), sh_degree=3):
    """make_case for a GIVEN scene (synthetic.make_scene's format; e.g. relightable3dgaussian_amd.trained_scene)."""
    P = sc["xyz"].shape[0]
    g = torch.Generator().manual_seed(seed + 100)
    feat = torch.rand(P, S, generator=g) if S > 0 else torch.zeros(P, 0)
    cpu = lambda t: t.detach().cpu().contiguous()
    return dict(P=P, W=W, H=H, S=S, bg=torch.tensor(bg, dtype=torch.float32), means3D=cpu(sc["xyz"]), features=feat,
                opacity=cpu(sc["opacity"]), scales=cpu(sc["scales"]), rotations=cpu(sc["rotations"]), shs=cpu(sc["shs"]),
                degree=sh_degree, cam=syn.look_at_camera(eye, width=W, height=H), colors=None, cov3D=None)


def fwd_args(case, device=None, debug=False):
    """Positional args of `_C.rasterize_gaussians` for a case (optionals as empty CPU tensors, like the reference)."""
    cam = case["cam"]
    empty = torch.Tensor([])

    def dv(t):
        if t is None:
            return empty
        return t.to(device) if device is not None else t
    return (dv(case["bg"]), dv(case["means3D"]), dv(case["features"]), dv(case["colors"]), dv(case["opacity"]),
            dv(case["scales"]), dv(case["rotations"]), 1.0, dv(case["cov3D"]), dv(cam.world_view_transform),
            dv(cam.full_proj_transform), cam.tanfovx, cam.tanfovy, cam.cx, cam.cy, case["H"], case["W"],
            dv(case["shs"]), case["degree"], dv(cam.camera_center), False, True, debug)
