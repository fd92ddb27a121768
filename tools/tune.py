"""Sweep the tile-kernel tuning knobs at full size and print per-stage HIP-event times (GPU only)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightable3dgaussian_amd import _lib, synthetic as syn
from r3dg_rasterization import _C

P = int(os.environ.get("P", 300000)); RES = int(os.environ.get("RES", 800))
dev = "cuda"
L = _lib.lib()
sc = syn.make_scene(P=P, seed=0, stage2=False)
cam = syn.orbit_cameras(100, width=RES, height=RES)[0].to(dev)
empty = torch.Tensor([])
bg = torch.ones(3, device=dev)
d = {k: v.to(dev) for k, v in sc.items() if torch.is_tensor(v)}

def run(S, fppl, bppl, dpp, iters=10):
    _lib.set_option("FWD_PIXELS_PER_LANE", fppl)
    _lib.set_option("BWD_PIXELS_PER_LANE", bppl)
    feat = torch.rand(P, S, device=dev)
    gC, gO, gD, gF = [torch.randn(c, RES, RES, device=dev) for c in (3, 1, 1, S)]
    for it in range(iters + 3):
        if it == 3:
            torch.cuda.synchronize(); L.r3dg_profile_enable(1)
        out = _C.rasterize_gaussians(bg, d["xyz"], feat, empty, d["opacity"], d["scales"], d["rotations"], 1.0, empty,
                                     cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.cx,
                                     cam.cy, RES, RES, d["shs"], 3, cam.camera_center, False, True, False)
        _C.rasterize_gaussians_backward(bg, d["xyz"], feat, out[9], empty, d["scales"], d["rotations"], 1.0, empty,
                                        cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy,
                                        gC, gO, gD, gF, d["shs"], 3, cam.camera_center, out[10], out[0], out[11],
                                        out[12], True, False)
    torch.cuda.synchronize()
    prof = _lib.profile_read(); L.r3dg_profile_enable(0)
    return out[0], {k: round(v[0] / max(v[1], 1), 4) for k, v in prof.items()}

for S in (5, 16):
    for fu in (1, 2, 4):
        for order in (0, 1):
            _lib.set_option("FWD_UNROLL", fu); _lib.set_option("BWD_UNROLL", 2); _lib.set_option("TILE_ORDER", order)
            R, pr = run(S, 1, 1, 1)
            print("S=%d fwd_unroll=%d order=%d R=%d render_forward %.4f ms" % (S, fu, order, R, pr["render_forward"]))
    for bu in (1, 2, 4):
        for order in (0, 1):
            _lib.set_option("FWD_UNROLL", 4); _lib.set_option("BWD_UNROLL", bu); _lib.set_option("TILE_ORDER", order)
            R, pr = run(S, 1, 1, 1)
            print("S=%d bwd_unroll=%d order=%d render_backward %.4f ms" % (S, bu, order, pr["render_backward"]))
    print("S=%d all stages:" % S, json.dumps(pr))
