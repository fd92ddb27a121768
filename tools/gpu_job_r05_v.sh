#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
R=$PWD; O=$R/gpurun_out
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 3 < /dev/null 2> $O/v_err.txt | tail -1 | cut -c1-200
cd /tmp; rm -rf /tmp/prof
R3DG_BENCH_NO_ALONE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 < /dev/null > $O/v_prof.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd $R
python tools/rocpd_timeline.py "$f" 15 < /dev/null > $O/v_timeline.txt 2>&1
python tools/rocpd_timeline.py "$f" seq < /dev/null > $O/v_sequence.txt 2>&1
head -8 $O/v_timeline.txt | cut -c1-160
cut -c1-130 $O/v_sequence.txt | head -34
