#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_fused_dp_gpu.py -q -x -p no:cacheprovider < /dev/null > gpurun_out/r_pytest.txt 2>&1; tail -12 gpurun_out/r_pytest.txt
