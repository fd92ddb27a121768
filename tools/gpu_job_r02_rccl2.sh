#!/bin/bash
# One gpurun call (~2.5 GPU-minutes): every data-parallel and relight GPU test, then the one-rank-RCCL bench line and a short
# default bench.   gpurun --timeout 400 -- 'bash tools/gpu_job_r02_rccl2.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 240 python -m pytest tests/test_fused_dp_gpu.py tests/test_relight_gpu.py -x -q > gpurun_out/rccl2_pytest.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/rccl2_pytest.txt
tail -8 gpurun_out/rccl2_pytest.txt
R3DG_DP_SINGLE_RANK=1 timeout 120 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs \
    --relight-frames 0 --repeats 0 > gpurun_out/bench_dp1_rccl.json 2> gpurun_out/bench_dp1_rccl.err
echo "dp1 bench rc=$?"; cut -c1-260 gpurun_out/bench_dp1_rccl.json
timeout 150 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs --relight-frames 12 --repeats 1 \
    > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
echo "quick bench rc=$?"; tail -3 gpurun_out/bench_quick.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_quick.json"))
    print("value", d["value"], "relight", d["relight"]["relight_fps"], d["relight"].get("relight_rotating_light"))
except Exception as e:
    print("no quick bench line:", e)
PY
