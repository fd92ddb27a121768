#!/bin/bash
# after the chain kernel: the whole GPU suite + the default bench
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
O=gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider < /dev/null > $O/aa_pytest_gpu.txt 2>&1; tail -4 $O/aa_pytest_gpu.txt | cut -c1-200
timeout 900 python bench.py < /dev/null > $O/aa_bench_default.out 2> $O/aa_bench_default.err; tail -1 $O/aa_bench_default.out > $O/aa_bench_default_compact.json
cp $O/bench_full.json $O/aa_bench_default.json
python tools/bench_summary.py $O/aa_bench_default.json | head -40
