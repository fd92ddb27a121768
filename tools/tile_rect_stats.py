"""Distribution of the tile rectangles of a bench scene (GPU): how many Gaussians cover more than 32 tiles (the wave-cooperative
path of the binning kernels) and what share of the instances they carry.   python tools/tile_rect_stats.py P W H"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightable3dgaussian_amd import synthetic as syn, rasterizer_ops as ro
from relightable3dgaussian_amd.bench_core import GaussianParams, render_stage1
P, W, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
sc = syn.make_scene(P=P, seed=0, stage2=False)
params = GaussianParams(sc, dev, False)
cam = syn.orbit_cameras(8, width=W, height=H)[1].to(dev)
empty = torch.Tensor([])
feats = torch.zeros(P, 5, device=dev)
out = ro.rasterize_gaussians(torch.ones(3, device=dev), params.xyz, feats, empty, params.get_opacity(), params.get_scaling(),
                             params.get_rotation(), 1.0, empty, cam.world_view_transform.contiguous(),
                             cam.full_proj_transform.contiguous(), cam.tanfovx, cam.tanfovy, cam.cx, cam.cy, H, W,
                             params.get_shs().contiguous(), 3, cam.camera_center.contiguous(), False, True, False)
R, geom, binning, img = out[0], out[-3], out[-2], out[-1]
st = ro.decode_state(geom, binning, img, P, R, H, W)
t = st["tiles_touched"].long()
live = t > 0
print("P=%d %dx%d num_rendered=%d live=%d mean tiles per live Gaussian %.2f" % (P, W, H, int(t.sum()), int(live.sum()), float(t[live].float().mean())))
for thr in (8, 16, 32, 64, 128, 256, 1024):
    big = t > thr
    print("  > %4d tiles: %7d Gaussians (%.2f %% of live), %.1f %% of the instances" % (thr, int(big.sum()), 100.0 * int(big.sum()) / max(1, int(live.sum())), 100.0 * float(t[big].sum()) / max(1.0, float(t.sum()))))
print("  max", int(t.max()))
# per wave of 64 consecutive Gaussians: serial steps of the small path (max over lanes of min(t,32)-ish) and big ones
tw = torch.nn.functional.pad(t, (0, (-P) % 64)).view(-1, 64)
small = torch.where(tw > 32, torch.zeros_like(tw), tw)
print("  per wave: mean of max small-rect length %.1f, mean number of big rects %.2f, mean big-rect iterations %.1f" % (
    float(small.max(1).values.float().mean()), float((tw > 32).sum(1).float().mean()),
    float(torch.where(tw > 32, (tw + 63) // 64, torch.zeros_like(tw)).sum(1).float().mean())))
