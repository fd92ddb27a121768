#!/bin/bash
# the default bench line once more (the pool's boxes differ by ~7 % on memory-bound kernels: profiles/README.md)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
O=gpurun_out
timeout 900 python bench.py < /dev/null > $O/s_bench_default.out 2> $O/s_bench_default.err; tail -1 $O/s_bench_default.out > $O/s_bench_default_compact.json
cp $O/bench_full.json $O/s_bench_default.json
cut -c1-300 $O/s_bench_default_compact.json
python tools/bench_summary.py $O/s_bench_default.json | head -3
