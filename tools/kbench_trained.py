"""The rasterizer forward alone on the TRAINED scene (trained_scene.train_scene) and on the i.i.d. scene, stage times
from the library's HIP events; for rocprofv3 runs:  [SCENE=trained|iid|heavy] python tools/kbench_trained.py [iters]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightable3dgaussian_amd import _lib, synthetic as syn, trained_scene as ts, rasterizer_ops as ro
dev = torch.device("cuda", 0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
kind = os.environ.get("SCENE", "trained")
# SCENE_SAVE=<file>: train (or build) the scene, save it and exit; SCENE_FILE=<file>: run on a saved scene -- so that a counter pass
# (rocprofv3 --pmc profiles EVERY dispatch of the process) sees the forward on the trained scene only, not the 75 000 launches that trained it
if os.environ.get("SCENE_FILE"):
    sc = torch.load(os.environ["SCENE_FILE"])
else:
    sc = ts.train_scene(dev, stage2=False) if kind == "trained" else ts.heavy_tail_scene(stage2=False) if kind == "heavy" else \
        syn.make_scene(P=300_000, seed=0, stage2=False)
if os.environ.get("SCENE_SAVE"):
    torch.save({k: v for k, v in sc.items() if torch.is_tensor(v)}, os.environ["SCENE_SAVE"])
    print("saved", os.environ["SCENE_SAVE"], sc["xyz"].shape[0])
    sys.exit(0)
P = sc["xyz"].shape[0]
W = H = int(os.environ.get("RES", 800))
cams = [c.to(dev) for c in syn.orbit_cameras(100, width=W, height=H)[:4]]
t = lambda k: sc[k].to(dev).contiguous()
xyz, op, scl, rot, shs = t("xyz"), t("opacity"), t("scales"), t("rotations"), t("shs")
S = int(os.environ.get("S", 16))
feat = torch.rand(P, S, device=dev)
empty = torch.Tensor([])
bg = torch.ones(3, device=dev)
L = _lib.lib()


def fwd(cam):
    return ro.rasterize_gaussians(bg, xyz, feat, empty, op, scl, rot, 1.0, empty, cam.world_view_transform.contiguous(),
                                  cam.full_proj_transform.contiguous(), cam.tanfovx, cam.tanfovy, cam.cx, cam.cy, H, W, shs, 3,
                                  cam.camera_center.contiguous(), False, True, False)
for i in range(3):
    out = fwd(cams[i])
torch.cuda.synchronize()
L.r3dg_profile_enable(1)
for i in range(iters):
    out = fwd(cams[i % 4])
torch.cuda.synchronize()
prof = _lib.profile_read()
L.r3dg_profile_enable(0)
print(kind, "P=%d num_rendered=%d" % (P, int(out[0])), {k: round(ms / max(n, 1), 4) for k, (ms, n) in prof.items() if n})
