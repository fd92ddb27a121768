"""The rasterizer forward alone at P W H, a few times (for rocprofv3 --kernel-trace --stats runs of the binning kernels):
    P=2000000 W=1800 H=700 ITERS=6 python tools/kbench_binning.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightable3dgaussian_amd import _lib, synthetic as syn
from r3dg_rasterization import _C
P, W, H = int(os.environ.get("P", 300000)), int(os.environ.get("W", 800)), int(os.environ.get("H", 800))
dev = "cuda"
for name in _lib.OPTIONS:
    if os.environ.get("R3DG_OPT_" + name):
        _lib.set_option(name, int(os.environ["R3DG_OPT_" + name]))
sc = syn.make_scene(P=P, seed=0, stage2=False)
cam = syn.orbit_cameras(8, width=W, height=H)[1].to(dev)
empty = torch.Tensor([]); bg = torch.ones(3, device=dev)
d = {k: v.to(dev) for k, v in sc.items() if torch.is_tensor(v)}
feat = torch.rand(P, 16, device=dev)
for it in range(2 + int(os.environ.get("ITERS", 6))):
    out = _C.rasterize_gaussians(bg, d["xyz"], feat, empty, d["opacity"], d["scales"], d["rotations"], 1.0, empty,
                                 cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.cx,
                                 cam.cy, H, W, d["shs"], 3, cam.camera_center, False, True, False)
torch.cuda.synchronize()
print("num_rendered", out[0])
