"""Run the rasterizer fwd+bwd alone at full size a few times (for rocprofv3 PMC runs).
ACTIVE=2,3,4: the backward carries only these feature channels (the other columns of the upstream feature gradient are zero) --
the launch configuration of the stage-2 training iteration (run_nerf.sh objective: the three pbr maps), S stays 16."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightable3dgaussian_amd import _lib, synthetic as syn
from relightable3dgaussian_amd import rasterizer_ops
from r3dg_rasterization import _C
P = int(os.environ.get("P", 300000)); RES = int(os.environ.get("RES", 800)); S = int(os.environ.get("S", 16))
dev = "cuda"; L = _lib.lib()
sc = syn.make_scene(P=P, seed=0, stage2=False)
cam = syn.orbit_cameras(100, width=RES, height=RES)[0].to(dev)
empty = torch.Tensor([]); bg = torch.ones(3, device=dev)
d = {k: v.to(dev) for k, v in sc.items() if torch.is_tensor(v)}
feat = torch.rand(P, S, device=dev)
gC, gO, gD, gF = [torch.randn(c, RES, RES, device=dev) for c in (3, 1, 1, S)]
ACTIVE = [int(c) for c in os.environ["ACTIVE"].split(",")] if os.environ.get("ACTIVE") else None
if ACTIVE is not None:
    keep = torch.zeros(S, 1, 1, device=dev)
    keep[ACTIVE] = 1.0
    gF = gF * keep
    gD = torch.zeros_like(gD)                     # (the depth image carries no loss term in stage 2)
    if os.environ.get("NODEPTH"):
        gD = empty                                # ... and the fused iterations say so: no depth gradient at all
L.r3dg_profile_enable(1)
for name in _lib.OPTIONS:                                # R3DG_OPT_<NAME>=<value>, e.g. R3DG_OPT_CULL=0
    if os.environ.get("R3DG_OPT_" + name):
        _lib.set_option(name, int(os.environ["R3DG_OPT_" + name]))
for it in range(3 + int(os.environ.get("ITERS", 10))):
    if it == 3:
        torch.cuda.synchronize(); _lib.profile_read(); L.r3dg_profile_enable(1)
    out = _C.rasterize_gaussians(bg, d["xyz"], feat, empty, d["opacity"], d["scales"], d["rotations"], 1.0, empty,
                                 cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.cx,
                                 cam.cy, RES, RES, d["shs"], 3, cam.camera_center, False, True, False)
    if ACTIVE is None:
        _C.rasterize_gaussians_backward(bg, d["xyz"], feat, out[9], empty, d["scales"], d["rotations"], 1.0, empty,
                                        cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy,
                                        gC, gO, gD, gF, d["shs"], 3, cam.camera_center, out[10], out[0], out[11],
                                        out[12], True, False)
    else:
        rasterizer_ops.rasterize_gaussians_backward(bg, d["xyz"], feat, out[9], empty, d["scales"], d["rotations"], 1.0, empty,
                                                    cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy,
                                                    gC, gO, gD, gF, d["shs"], 3, cam.camera_center, out[10], out[0], out[11],
                                                    out[12], True, False, active_features=ACTIVE)
torch.cuda.synchronize()
pr = _lib.profile_read()
print({k: round(v[0] / max(v[1], 1), 4) for k, v in pr.items() if v[1]})
