"""The instance-ordering stages of the rasterizer forward at the composition scale (BASELINE configs[4]: 2 M Gaussians,
1800x700) with the distribution of tile lengths: HIP-event ms per stage (under rocprofv3 --kernel-trace: the kernels)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightable3dgaussian_amd import _lib, synthetic as syn
from relightable3dgaussian_amd.rasterizer_ops import decode_state
from r3dg_rasterization import _C

P = int(os.environ.get("P", 2_000_000)); W = int(os.environ.get("W", 1800)); H = int(os.environ.get("H", 700)); S = 16
dev = "cuda"; L = _lib.lib()
sc = syn.make_scene(P=P, seed=0, stage2=False)
cam = syn.orbit_cameras(100, width=W, height=H)[0].to(dev)
empty = torch.Tensor([]); bg = torch.ones(3, device=dev)
d = {k: v.to(dev) for k, v in sc.items() if torch.is_tensor(v)}
feat = torch.rand(P, S, device=dev)


def forward():
    return _C.rasterize_gaussians(bg, d["xyz"], feat, empty, d["opacity"], d["scales"], d["rotations"], 1.0, empty,
                                  cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.cx, cam.cy,
                                  H, W, d["shs"], 3, cam.camera_center, False, True, False)


doc = {"points": P, "image": "%dx%d" % (W, H)}
for _ in range(3):
    out = forward()
torch.cuda.synchronize()
L.r3dg_profile_enable(1)
for _ in range(8):
    out = forward()
torch.cuda.synchronize()
pr = _lib.profile_read()
L.r3dg_profile_enable(0)
st = decode_state(out[10], out[11], out[12], P, out[0], H, W)
doc["stage_ms"] = {k: round(v[0] / max(v[1], 1), 4) for k, v in pr.items() if v[1]}
lens = torch.as_tensor(st["ranges"]).long()
lens = (lens[:, 1] - lens[:, 0])
doc["num_rendered"] = int(out[0])
doc["tiles"] = int(lens.numel())
doc["tile_length"] = {"max": int(lens.max()), "mean": round(float(lens.float().mean()), 1),
                      "over_4096": int((lens > 4096).sum()), "over_16384": int((lens > 16384).sum()),
                      "instances_in_tiles_over_4096": int(lens[lens > 4096].sum())}
print(json.dumps(doc, indent=1))
