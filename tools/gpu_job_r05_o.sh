#!/bin/bash
# 2M-Gaussian configuration: one step launch by launch
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
R=$PWD; O=$R/gpurun_out
cd /tmp; rm -rf /tmp/prof
R3DG_BENCH_NO_ALONE=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python $R/bench.py --points 2000000 --width 1800 --height 700 --steps 12 --warmup 4 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 < /dev/null > $O/o_prof.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd $R
python tools/rocpd_timeline.py "$f" 8 < /dev/null > $O/o_timeline_2m.txt 2>&1
python tools/rocpd_timeline.py "$f" seq < /dev/null > $O/o_sequence_2m.txt 2>&1
cut -c1-150 $O/o_sequence_2m.txt | head -45
