#!/bin/bash
# flag ring / replay of dropped views: the tests that touch the bounded forward + a short bench
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_fused_dp_gpu.py tests/test_train_loop_gpu.py tests/test_densify_gpu.py -q -x -p no:cacheprovider < /dev/null > gpurun_out/j_pytest.txt 2>&1; tail -15 gpurun_out/j_pytest.txt
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 2 < /dev/null 2> gpurun_out/j_bench.err | tail -1 | cut -c1-400
