#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
show() { python - "$1" <<'P'
import json, sys
d=json.load(open("gpurun_out/bench_full.json"))
print(sys.argv[1], d["value"], d["spread_iters_per_s"]["median"], {k:v["ms_per_iteration"] for k,v in d["kernels"].items() if k in ("duplicate_with_keys","sort_pairs","shade_forward")})
P
}
B="--no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 2"
for v in 1 0; do
  export R3DG_SHADE_LEAVE_ROOM=$v
  timeout 300 python bench.py --steps 100 --warmup 10 --objective syn4 $B < /dev/null > /dev/null 2> gpurun_out/u_err.txt; show "syn4 K=64 leave_room=$v"
  timeout 300 python bench.py --steps 60 --warmup 10 --sample-num 384 $B < /dev/null > /dev/null 2> gpurun_out/u_err.txt; show "nerf K=384 leave_room=$v"
  timeout 300 python bench.py --steps 60 --warmup 10 --sample-num 384 --objective syn4 $B < /dev/null > /dev/null 2> gpurun_out/u_err.txt; show "syn4 K=384 leave_room=$v"
  timeout 300 python bench.py --width 1600 --height 1200 --sample-num 32 --objective syn4 --steps 60 --warmup 10 $B < /dev/null > /dev/null 2> gpurun_out/u_err.txt; show "DTU leave_room=$v"
  timeout 300 python bench.py --points 2000000 --width 1800 --height 700 --steps 24 --warmup 6 $B < /dev/null > /dev/null 2> gpurun_out/u_err.txt; show "2M leave_room=$v"
done
