#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
show() { python - "$1" <<'P'
import json, sys
d=json.load(open("gpurun_out/bench_full.json"))
print(sys.argv[1], d["value"], d["spread_iters_per_s"]["median"], d["device_clock"]["shader_clock_ghz_under_valu_load"], {k:v["ms_per_iteration"] for k,v in d["kernels"].items() if k in ("duplicate_with_keys","sort_pairs","shade_frs_listed","shade_forward","render_forward")})
P
}
A="--steps 200 --warmup 20 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 3"
for v in 0 1 0 1; do
  R3DG_LISTED_EARLY=$v timeout 300 python bench.py $A < /dev/null > /dev/null 2> gpurun_out/w_err.txt; show "listed_early=$v"
done
timeout 600 python -m pytest tests/test_fused_step_gpu.py tests/test_shading_gpu.py -q -x -p no:cacheprovider -k "fused or listed or frs or fixed" < /dev/null > gpurun_out/w_pytest.txt 2>&1; tail -2 gpurun_out/w_pytest.txt
