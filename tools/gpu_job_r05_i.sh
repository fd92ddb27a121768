# round 5, call I: late zero fills / cached softplus / flag pair: fused tests + A/B bench + timeline + relight test
set -x
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_fused_dp_gpu.py tests/test_reference_pipeline_gpu.py tests/test_relight_gpu.py tests/test_train_loop_gpu.py tests/test_psnr_vs_reference_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/r05_i_tests.log 2>&1; tail -3 gpurun_out/r05_i_tests.log; grep -n "^E  \|bad [1-9]" gpurun_out/r05_i_tests.log | head -20
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 3"
run() { env "$@" $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d.get('spread_iters_per_s'), d['roofline']['avg_kernel_ms'])"; }
run A=1; run A=2; run A=3
cd /tmp
rm -rf /tmp/prof
R3DG_BENCH_NO_ALONE=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof -o bench -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 < /dev/null > /root/repo/gpurun_out/r05_i_prof.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_timeline.py "$f" seq < /dev/null > gpurun_out/r05_i_sequence.txt 2>&1
python tools/rocpd_timeline.py "$f" 12 < /dev/null > gpurun_out/r05_i_timeline.txt 2>&1
cat gpurun_out/r05_i_sequence.txt | cut -c1-150
head -4 gpurun_out/r05_i_timeline.txt
