"""Time the visibility trace alone (GPU only): P Gaussians, K rays each; PACKET=0 selects the thread-per-ray kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightable3dgaussian_amd import _lib, synthetic as syn
from relightable3dgaussian_amd.train_step import update_visibility
P = int(os.environ.get("P", 300000)); K = int(os.environ.get("K", 64)); dev = "cuda"
L = _lib.lib()
sc = syn.make_scene(P=P, seed=0, stage2=False)
d = {k: v.to(dev) for k, v in sc.items() if torch.is_tensor(v)}
res = {}
MODES = tuple(int(m) for m in os.environ.get('MODES', '4,3,2,0').split(','))
for packet in MODES:
    _lib.set_option("TRACE_FORMULATION", packet)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        vis, dirs, areas, tracer = update_visibility(d["xyz"], d["scales"], d["rotations"], d["opacity"], d["normal"], K)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    res[packet] = (dt, vis)
    print("packet=%d  P=%d K=%d  update_visibility %.3f s  = %.1f Mrays/s  (visible fraction %.3f)" % (
        packet, P, K, dt, P * K / dt / 1e6, (vis > 0).float().mean().item()))
if 3 in res and 4 in res:
    print('phased == persistent bitwise:', bool(torch.equal(res[4][1], res[3][1])))
if 0 not in res:
    sys.exit(0)
a, b = res[MODES[0]][1], res[0][1]
cls = ((a == 0) != (b == 0))
print("class mismatches packet vs per-ray: %d / %d; max |diff| elsewhere %.3e" % (cls.sum().item(), a.numel(), (a - b)[~cls].abs().max().item()))
