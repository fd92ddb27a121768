#!/bin/bash
# First GPU call for the opt-in kernels that were written without GPU access: their gated parity tests,
# then the default bench (whose child processes time them after an on-device parity check).   ~3 GPU-minutes.
#   gpurun --timeout 420 -- 'bash tools/gpu_job_experimental.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
R3DG_EXPERIMENTAL=1 timeout 240 python -m pytest tests/test_relight_gpu.py tests/test_fused_step_gpu.py -q \
    -k "transport_cache or saved_shading" > gpurun_out/experimental_pytest.txt 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/experimental_pytest.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_experimental.json 2> gpurun_out/bench_experimental.err
echo "bench rc=$?"; tail -3 gpurun_out/bench_experimental.err
python - <<'PY'
import json
try:
    d = json.load(open("gpurun_out/bench_experimental.json"))
    print("value", d["value"])
    print("saved_shading_intermediates", json.dumps(d["other_configs"].get("saved_shading_intermediates"))[:900])
    print("data_parallel_path_one_rank_rccl", d["other_configs"].get("data_parallel_path_one_rank_rccl"))
    print("relight", d["relight"]["relight_fps"], json.dumps(d["relight"].get("relight_transport_cache"))[:900])
except Exception as e:
    print("no bench line:", e)
PY
