#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
show() { python - "$1" <<'P'
import json, sys
d=json.load(open("gpurun_out/bench_full.json"))
print(sys.argv[1], d["value"], d["spread_iters_per_s"]["median"], d["device_clock"]["shader_clock_ghz_under_valu_load"], {k:v["ms_per_iteration"] for k,v in d["kernels"].items() if k in ("adam_step","shade_frs_aux","shade_forward","duplicate_with_keys","sort_pairs","stage2_activate","preprocess")})
P
}
A="--steps 200 --warmup 20 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 3"
for kb in 0 56 81 0 56 81; do
  R3DG_CHAIN_LDS_KB=$kb timeout 300 python bench.py $A < /dev/null > /dev/null 2> gpurun_out/ab_err.txt; show "chain LDS $kb KB"
done
ONLY64=1 ITERS=4 timeout 200 python tools/kbench_shade.py 2>&1 | tail -1 | tr ' ' '\n' | grep -A3 "incident" | tr '\n' ' '
