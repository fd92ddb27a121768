#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > gpurun_out/r05_smoke.txt 2>&1; tail -1 gpurun_out/r05_smoke.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider < /dev/null > gpurun_out/r05_pytest_gpu.txt 2>&1; tail -3 gpurun_out/r05_pytest_gpu.txt | cut -c1-200
