"""Compact view of a bench.py JSON line:  python tools/bench_summary.py gpurun_out/bench.json"""
import json
import sys

import os

d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
if "kernels" not in d and "full" in d:          # the compact stdout line: the document behind it is in gpurun_out/bench_full.json
    d = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_full.json")))
print("value %.1f %s  %.3f ms/step  spread %s" % (d["value"], d["unit"], d["ms_per_step"],
                                                  {k: d.get("spread_iters_per_s", {}).get(k) for k in ("min", "median", "max")}))
r = d["roofline"]
print("roofline:", {k: r.get(k) for k in ("bound", "kernel", "achieved", "frac", "traffic", "avg_kernel_ms", "algorithmic_MB",
                                          "feature_channels_in_backward")}, "valu_bound:", r.get("valu_bound"))
for k, v in d["kernels"].items():
    print("  %-26s %.4f ms x%d  %s MB  hbm %s" % (k, v["avg_ms"], v.get("launches_per_iteration", 1), v.get("algorithmic_MB"), v.get("hbm_frac")))
rl = d.get("relight") or {}
print("relight:", {k: rl.get(k) for k in ("relight_fps", "relight_fps_radiance_cache", "relight_fps_pytorch_glue", "visibility_Mrays_per_s")},
      "rotating:", (rl.get("relight_rotating_light") or {}).get("fps"))
for k, v in (d.get("other_configs") or {}).items():
    if isinstance(v, dict):
        keep = {a: v[a] for a in ("iters_per_s", "ms_per_step", "relight_fps", "exposed_comm_ms", "seconds", "failed", "skipped", "meets_target",
                                  "peak_memory_GB") if a in v}
        print("  [%s] %s" % (k[:90], keep))
        if "stage_ms" in v:
            top = sorted(v["stage_ms"].items(), key=lambda kv: -kv[1])[:8]
            print("      stage_ms:", {a: b for a, b in top})
        for a, b in (v.get("priced_all_reduce_8_ranks") or {}).items():
            print("      priced %s: %s" % (a, b))
    else:
        print("  [%s] %s" % (k[:90], v))
cb = d.get("cpu_baseline") or {}
print("cpu_baseline:", {k: cb.get(k) for k in ("value", "unit", "cores", "kind")})
print("host_cpu:", d.get("host_cpu"))
print("device_clock:", d.get("device_clock"))
