# round 5, call G: LDS-transposing scatter, 12-channel reduction, small launches on main, incident chain kernel: tests + A/B + timeline
set -x
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
timeout 900 python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_gpu.py -q -p no:cacheprovider -x < /dev/null > gpurun_out/r05_g_raster.log 2>&1; tail -3 gpurun_out/r05_g_raster.log
timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_reference_pipeline_gpu.py tests/test_fused_dp_gpu.py tests/test_train_loop_gpu.py tests/test_psnr_vs_reference_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/r05_g_fused.log 2>&1; tail -3 gpurun_out/r05_g_fused.log
timeout 600 python -m pytest tests/test_shading_gpu.py tests/test_relight_gpu.py -q -p no:cacheprovider -s -k "incident_chain or reference_python or fixed_ray_set_kernels_match_oracle" < /dev/null > gpurun_out/r05_g_shading.log 2>&1; tail -3 gpurun_out/r05_g_shading.log; grep -n "^E  \|incident chain:" gpurun_out/r05_g_shading.log | head
timeout 300 python tools/variants_bwd.py run > gpurun_out/r05_g_bwd_ablation.txt 2>&1; cat gpurun_out/r05_g_bwd_ablation.txt
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 3"
run() { env "$@" $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d.get('spread_iters_per_s'), d['roofline']['avg_kernel_ms'])"; }
run A=1; run R3DG_INCIDENT_CHAIN_KERNEL=0; run R3DG_FWD_STAGGER=aux; run A=1; run R3DG_INCIDENT_CHAIN_KERNEL=0; run R3DG_SHADE_LEAVE_ROOM=0
cd /tmp
rm -rf /tmp/prof
R3DG_BENCH_NO_ALONE=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof -o bench -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 < /dev/null > /root/repo/gpurun_out/r05_g_prof.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_timeline.py "$f" seq < /dev/null > gpurun_out/r05_g_sequence.txt 2>&1
python tools/rocpd_timeline.py "$f" 12 < /dev/null > gpurun_out/r05_g_timeline.txt 2>&1
cat gpurun_out/r05_g_sequence.txt | cut -c1-150
head -12 gpurun_out/r05_g_timeline.txt
