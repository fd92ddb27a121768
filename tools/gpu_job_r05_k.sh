#!/bin/bash
# folded launches: tests that touch the glue + rasterizer backward + pipeline parity, bench, one-step sequence
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
R=$PWD; O=$R/gpurun_out
timeout 1200 python -m pytest tests/test_fused_step_gpu.py tests/test_fused_dp_gpu.py tests/test_train_loop_gpu.py tests/test_rasterizer_gpu.py tests/test_reference_pipeline_gpu.py tests/test_relight_gpu.py -q -x -p no:cacheprovider < /dev/null > $O/k_pytest.txt 2>&1; tail -8 $O/k_pytest.txt
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 2 < /dev/null 2> $O/k_bench.err | tail -1 | cut -c1-300
cd /tmp; rm -rf /tmp/prof
R3DG_BENCH_NO_ALONE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 < /dev/null > $O/k_prof.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd $R
python tools/rocpd_timeline.py "$f" 15 < /dev/null > $O/k_timeline.txt 2>&1
python tools/rocpd_timeline.py "$f" seq < /dev/null > $O/k_sequence.txt 2>&1
head -12 $O/k_timeline.txt | cut -c1-160
