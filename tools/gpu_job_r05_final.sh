#!/bin/bash
# final tree: kernel stats + timeline + sequence of the training iteration, then the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
R=$PWD; O=$R/gpurun_out; RD=r05
cd /tmp; rm -rf /tmp/prof
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0"
R3DG_BENCH_NO_ALONE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- $CMD < /dev/null > $O/prof_bench.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd $R
python tools/rocpd_summary.py "$f" $O/${RD}_stage2_fused_bench_kernel_stats.md "R3DG_BENCH_NO_ALONE=1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0" < /dev/null
python tools/rocpd_timeline.py "$f" 15 < /dev/null > $O/${RD}_stage2_fused_step_timeline.txt 2>&1
python tools/rocpd_timeline.py "$f" seq < /dev/null > $O/${RD}_stage2_fused_step_sequence.txt 2>&1
timeout 900 python bench.py < /dev/null > $O/${RD}_bench_default.out 2> $O/${RD}_bench_default.err; tail -1 $O/${RD}_bench_default.out > $O/${RD}_bench_default_compact.json
cp $O/bench_full.json $O/${RD}_bench_default.json
cut -c1-300 $O/${RD}_bench_default_compact.json
python tools/bench_summary.py $O/${RD}_bench_default.json | head -60
