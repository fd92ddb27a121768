import sys, os, time, torch
sys.path.insert(0, os.getcwd())
from relightable3dgaussian_amd import synthetic as syn
from relightable3dgaussian_amd.bench_core import GaussianParams, render_stage1
from relightable3dgaussian_amd.fused_step import FusedStage2Step
dev = torch.device("cuda", 0)
P = 300000
scene = syn.make_scene(P=P, seed=0, stage2=True)
cams = [c.to(dev) for c in syn.orbit_cameras(100, width=800, height=800)[:8]]
bg = torch.ones(3, device=dev)
params = GaussianParams(scene, dev, True)
with torch.no_grad():
    teacher = GaussianParams(syn.make_scene(P=P, seed=0, stage2=False), dev, False)
    gts = [render_stage1(teacher, c, bg)[2].clone() for c in cams]
step = FusedStage2Step(params, 64, lr=1e-4)
for i in range(10): step(cams[i % 8], bg, gts[i % 8])
torch.cuda.synchronize()
N = 300
t0 = time.perf_counter()
for i in range(N): step(cams[i % 8], bg, gts[i % 8])
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.3f ms/iter, total %.3f ms/iter (GPU drained %.3f ms after the last enqueue)" % (1e3*(t1-t0)/N, 1e3*(t2-t0)/N, 1e3*(t2-t1)))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(100): step(cams[i % 8], bg, gts[i % 8])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
pstats.Stats(pr).sort_stats("tottime").print_stats(45)
