#!/bin/bash
# One gpurun call (~3 GPU-minutes): the data-parallel path over a ONE-rank RCCL group (tests + bench line) and a short
# default bench (relight incl. the rotating-light frames).   gpurun --timeout 400 -- 'bash tools/gpu_job_r02_rccl1.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 170 python -m pytest tests/test_fused_dp_gpu.py -x -q -k "single_rank" > gpurun_out/rccl1_pytest.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/rccl1_pytest.txt
tail -5 gpurun_out/rccl1_pytest.txt
R3DG_DP_SINGLE_RANK=1 timeout 120 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs \
    --relight-frames 0 --repeats 0 > gpurun_out/bench_dp1_rccl.json 2> gpurun_out/bench_dp1_rccl.err
echo "dp1 bench rc=$?"; tail -3 gpurun_out/bench_dp1_rccl.err; cut -c1-400 gpurun_out/bench_dp1_rccl.json
timeout 150 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs --relight-frames 12 --repeats 1 \
    > gpurun_out/bench_quick.json 2> gpurun_out/bench_quick.err
echo "quick bench rc=$?"; tail -3 gpurun_out/bench_quick.err; cut -c1-300 gpurun_out/bench_quick.json
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_quick.json"))
    print("value", d["value"], "relight", d["relight"]["relight_fps"], d["relight"].get("relight_rotating_light"))
except Exception as e:
    print("no quick bench line:", e)
PY
