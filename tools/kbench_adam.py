"""The multi-group Adam launch alone, at the stage-2 sizes (127 floats per Gaussian in 10 groups), COLD: a 600 MB write between
launches evicts the 256 MB last-level cache, so every launch streams its 28 bytes per parameter float from / to HBM.
    [R3DG_LIB_PATH=<variant>] python tools/kbench_adam.py [P] [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightable3dgaussian_amd.fused_step import FusedAdam
P = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
shapes = [(P, 3), (P, 3), (P, 3), (P, 4), (P, 1), (P, 16, 3), (P, 3), (P, 1), (P, 16, 3), (1, 16, 32, 3)]
params = [torch.randn(*s, device=dev) for s in shapes]
grads = [torch.randn(*s, device=dev) for s in shapes]
opt = FusedAdam([dict(param=p, lr=1e-4, period=48 if p.dim() == 3 else 0, split=3, lr_tail=5e-6 if p.dim() == 3 else None) for p in params])
evict = torch.empty(150_000_000, device=dev)
floats = sum(p.numel() for p in params)
for warm in (True, False):
    ts = []
    for i in range(3 if warm else n):
        evict.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        opt.step(grads)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
ts.sort()
med = ts[len(ts) // 2]
print("adam P=%d: %d floats, %.1f MB per launch; median %.4f ms (min %.4f) = %.0f GB/s = %.3f of 8 TB/s" % (
    P, floats, floats * 28 / 1e6, med, ts[0], floats * 28 / med / 1e6, floats * 28 / med / 1e6 / 8000))
# warm: back to back (what the iteration sees when the gradients were just written)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(n):
    opt.step(grads)
e1.record()
torch.cuda.synchronize()
print("   back to back: %.4f ms per launch" % (e0.elapsed_time(e1) / n))
