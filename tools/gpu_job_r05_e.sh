# round 5, call E: gradient records in the tile backward: tests + bench + ablations
set -x
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
timeout 900 python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_gpu.py -q -p no:cacheprovider -x < /dev/null > gpurun_out/r05_e_raster.log 2>&1; tail -4 gpurun_out/r05_e_raster.log
timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_reference_pipeline_gpu.py tests/test_fused_dp_gpu.py tests/test_densify_gpu.py tests/test_train_loop_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/r05_e_fused.log 2>&1; tail -4 gpurun_out/r05_e_fused.log
timeout 300 python -m pytest tests/test_relight_gpu.py -q -p no:cacheprovider -s -k "reference_python" < /dev/null > gpurun_out/r05_e_relight.log 2>&1; tail -3 gpurun_out/r05_e_relight.log; grep -n "visibility classes\|^E  " gpurun_out/r05_e_relight.log | head
timeout 300 python tools/variants_bwd.py run > gpurun_out/r05_e_bwd_ablation.txt 2>&1; cat gpurun_out/r05_e_bwd_ablation.txt
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 3"
for i in 1 2; do $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('head', d['value'], d.get('spread_iters_per_s'), d['roofline'])"; done
python tools/bench_summary.py gpurun_out/bench_full.json | head -24
