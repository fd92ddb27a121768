# round 5, call H: after deleting the chain kernel: long-tile sort on a side stream, small launches on main -- A/B + timeline + tests
set -x
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
timeout 600 python -m pytest tests/test_rasterizer_gpu.py tests/test_relight_gpu.py tests/test_fused_step_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/r05_h_tests.log 2>&1; tail -3 gpurun_out/r05_h_tests.log; grep -n "^E  \|bad [1-9]" gpurun_out/r05_h_tests.log | head -20
timeout 300 python tools/variants_bwd.py run > gpurun_out/r05_h_bwd_ablation.txt 2>&1; cat gpurun_out/r05_h_bwd_ablation.txt
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 3"
run() { env "$@" $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d.get('spread_iters_per_s'), d['roofline']['avg_kernel_ms'])"; }
run A=1; run R3DG_OPT_SORT_LONG_SIDE_STREAM=0; run R3DG_FWD_STAGGER=aux; run A=1; run R3DG_OPT_SORT_LONG_SIDE_STREAM=0; run R3DG_FWD_STAGGER=aux; run R3DG_SHADE_LEAVE_ROOM=0
cd /tmp
rm -rf /tmp/prof
R3DG_BENCH_NO_ALONE=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof -o bench -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 < /dev/null > /root/repo/gpurun_out/r05_h_prof.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_timeline.py "$f" seq < /dev/null > gpurun_out/r05_h_sequence.txt 2>&1
python tools/rocpd_timeline.py "$f" 12 < /dev/null > gpurun_out/r05_h_timeline.txt 2>&1
cat gpurun_out/r05_h_sequence.txt | cut -c1-150
head -12 gpurun_out/r05_h_timeline.txt
