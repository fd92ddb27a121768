#!/bin/bash
# chain kernel for the frozen-SH schedules (run_syn4.sh / run_dtu.sh): A/B + tests
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
show() { python - "$1" <<'P'
import json, sys
d=json.load(open("gpurun_out/bench_full.json"))
print(sys.argv[1], d["value"], d["spread_iters_per_s"]["median"], {k:v["ms_per_iteration"] for k,v in d["kernels"].items() if k in ("adam_step","shade_frs_aux","shade_forward","stage2_activate")})
P
}
B="--no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 2"
for v in 0 1 0 1; do
  export R3DG_EARLY_INCIDENTS=$v
  timeout 300 python bench.py --steps 100 --warmup 10 --objective syn4 $B < /dev/null > /dev/null 2> gpurun_out/ac_err.txt; show "syn4 K=64 chain=$v"
  timeout 300 python bench.py --steps 60 --warmup 10 --sample-num 384 --objective syn4 $B < /dev/null > /dev/null 2> gpurun_out/ac_err.txt; show "syn4 K=384 chain=$v"
  timeout 300 python bench.py --width 1600 --height 1200 --sample-num 32 --objective syn4 --steps 60 --warmup 10 $B < /dev/null > /dev/null 2> gpurun_out/ac_err.txt; show "DTU chain=$v"
done
unset R3DG_EARLY_INCIDENTS
timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_reference_pipeline_gpu.py -q -x -p no:cacheprovider < /dev/null > gpurun_out/ac_pytest.txt 2>&1; tail -3 gpurun_out/ac_pytest.txt | cut -c1-200
