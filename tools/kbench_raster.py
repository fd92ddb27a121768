"""Run the rasterizer fwd+bwd alone at full size a few times (for rocprofv3 PMC runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightable3dgaussian_amd import _lib, synthetic as syn
from r3dg_rasterization import _C
P = int(os.environ.get("P", 300000)); RES = int(os.environ.get("RES", 800)); S = int(os.environ.get("S", 16))
dev = "cuda"; L = _lib.lib()
sc = syn.make_scene(P=P, seed=0, stage2=False)
cam = syn.orbit_cameras(100, width=RES, height=RES)[0].to(dev)
empty = torch.Tensor([]); bg = torch.ones(3, device=dev)
d = {k: v.to(dev) for k, v in sc.items() if torch.is_tensor(v)}
feat = torch.rand(P, S, device=dev)
gC, gO, gD, gF = [torch.randn(c, RES, RES, device=dev) for c in (3, 1, 1, S)]
L.r3dg_profile_enable(1)
e = os.environ
L.r3dg_set_tuning(int(e.get("FPPL", 0)), int(e.get("BPPL", 0)), -1)
L.r3dg_set_tuning2(int(e.get("FU", 0)), int(e.get("BU", 0)), int(e.get("ORDER", -1)))
if "WAVE8" in os.environ:
    L.r3dg_set_tuning3(int(os.environ["WAVE8"]) & 1, int(os.environ["WAVE8"]) >> 1, -1)
if "BIN" in os.environ:
    L.r3dg_set_tuning4(int(os.environ["BIN"]))
if "STAGE" in os.environ:
    L.r3dg_set_tuning5(int(os.environ["STAGE"]))
if "CULL" in os.environ:
    L.r3dg_set_tuning3(-1, -1, int(os.environ["CULL"]))
for it in range(3 + int(os.environ.get("ITERS", 10))):
    if it == 3:
        torch.cuda.synchronize(); _lib.profile_read(); L.r3dg_profile_enable(1)
    out = _C.rasterize_gaussians(bg, d["xyz"], feat, empty, d["opacity"], d["scales"], d["rotations"], 1.0, empty,
                                 cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.cx,
                                 cam.cy, RES, RES, d["shs"], 3, cam.camera_center, False, True, False)
    _C.rasterize_gaussians_backward(bg, d["xyz"], feat, out[9], empty, d["scales"], d["rotations"], 1.0, empty,
                                    cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy,
                                    gC, gO, gD, gF, d["shs"], 3, cam.camera_center, out[10], out[0], out[11],
                                    out[12], True, False)
torch.cuda.synchronize()
pr = _lib.profile_read()
print({k: round(v[0] / max(v[1], 1), 4) for k, v in pr.items() if v[1]})
