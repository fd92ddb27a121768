"""Explore the trained-scene workload on the GPU: python tools/trained_scene_probe.py [key=value ...]  (train_scene's arguments)"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from relightable3dgaussian_amd import synthetic as syn, trained_scene as ts
kw = {}
for a in sys.argv[1:]:
    k, v = a.split("=")
    kw[k] = float(v) if "." in v or "e" in v else int(v)
dev = torch.device("cuda", 0)
hist = []
t0 = time.perf_counter()
sc = ts.train_scene(dev, history_out=hist, **kw)
dt = time.perf_counter() - t0
print("trained in %.1f s: P=%d; densify rows %s; dropped %s" % (dt, sc["xyz"].shape[0], [r for _, e, r in hist if e == "densify"][::3],
                                                              sum(r for _, e, r in hist if e == "dropped_views")))
cam = syn.orbit_cameras(100, width=kw.get("res", 800), height=kw.get("res", 800))[0].to(dev)
print(json.dumps(ts.binning_stats(sc, cam, dev)))
print("i.i.d. 300k:", json.dumps(ts.binning_stats(syn.make_scene(P=300_000, seed=0, stage2=False), cam, dev)))
print("heavy tail :", json.dumps(ts.binning_stats(ts.heavy_tail_scene(stage2=False), cam, dev)))
