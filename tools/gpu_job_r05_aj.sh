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
A="--steps 200 --warmup 20 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 3"
for b in 1 2 3 1 2; do
  R3DG_OPT_SHADE_FWD_BLOCKS_PER_CU=$b timeout 300 python bench.py $A < /dev/null > /dev/null 2> gpurun_out/aj_err.txt; show "fwd blocks per CU $b"
done
