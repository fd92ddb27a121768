#!/bin/bash
# Round 4, last check of the tree: smoke, the full GPU suite, kernel stats + timeline of the bench command, the default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
R=$PWD
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > gpurun_out/r04_smoke.txt 2>&1; tail -1 gpurun_out/r04_smoke.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider < /dev/null > gpurun_out/r04_pytest_gpu.txt 2>&1; tail -2 gpurun_out/r04_pytest_gpu.txt
cd /tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0"
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- $CMD < /dev/null > $R/gpurun_out/prof_bench.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd $R
python tools/rocpd_summary.py "$f" gpurun_out/r04_stage2_fused_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0" < /dev/null
python tools/rocpd_timeline.py "$f" 15 < /dev/null > gpurun_out/r04_stage2_fused_step_timeline.txt 2>&1
python tools/rocpd_timeline.py "$f" seq < /dev/null > gpurun_out/r04_stage2_fused_step_sequence.txt 2>&1
timeout 900 python bench.py < /dev/null > gpurun_out/r04_bench_default.log 2>&1; tail -1 gpurun_out/r04_bench_default.log > gpurun_out/r04_bench_default.json
cut -c1-300 gpurun_out/r04_bench_default.json
