#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
show() { python - "$1" <<'P'
import json, sys
d=json.load(open("gpurun_out/bench_full.json"))
print(sys.argv[1], d["value"], d["spread_iters_per_s"]["median"], {k:v["ms_per_iteration"] for k,v in d["kernels"].items() if k in ("duplicate_with_keys","sort_pairs","shade_forward","preprocess")})
P
}
A="--points 2000000 --width 1800 --height 700 --steps 24 --warmup 6 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 2"
for b in 1 2 3; do
  R3DG_OPT_SHADE_FWD_BLOCKS_PER_CU=$b timeout 300 python bench.py $A < /dev/null > /dev/null 2> gpurun_out/ai_err.txt; show "2M fwd blocks per CU $b"
done
R3DG_SHADE_LEAVE_ROOM=0 timeout 300 python bench.py $A < /dev/null > /dev/null 2> gpurun_out/ai_err.txt; show "2M uncapped"
timeout 300 python bench.py $A < /dev/null > /dev/null 2> gpurun_out/ai_err.txt; show "2M default"
