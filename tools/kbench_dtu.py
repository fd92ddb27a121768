"""The DTU configuration of bench.py alone (configs[3]: 1600x1200, sample_num 32, run_dtu.sh objective, frozen geometry), for
rocprofv3 runs:  python tools/kbench_dtu.py [steps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from relightable3dgaussian_amd import bench_core as bc   # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
r = bc.config_rate("cuda", int(os.environ.get("P", 300000)), int(os.environ.get("W", 1600)), int(os.environ.get("H", 1200)),
                   sample_num=int(os.environ.get("K", 32)), objective=os.environ.get("OBJECTIVE", "syn4"), steps=steps, warmup=3,
                   stage_ms=True)
print(r["iters_per_s"], r["ms_per_step"], {k: v for k, v in sorted(r["stage_ms"].items(), key=lambda kv: -kv[1])[:10]})
