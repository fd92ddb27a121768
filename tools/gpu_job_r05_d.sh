# round 5, call D: attribution of the tile backward's time (ablation builds: tools/variants_bwd.py), relight reference test
set -x
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
timeout 600 python tools/variants_bwd.py run > gpurun_out/r05_d_bwd_ablation.txt 2>&1; cat gpurun_out/r05_d_bwd_ablation.txt
timeout 600 python -m pytest tests/test_relight_gpu.py -q -p no:cacheprovider -s -k "reference_python" < /dev/null > gpurun_out/r05_d_relight.log 2>&1; tail -4 gpurun_out/r05_d_relight.log
grep -n "visibility classes\|^E  " gpurun_out/r05_d_relight.log | head
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 3"
for i in 1 2; do $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('head', d['value'], d.get('spread_iters_per_s'), d['roofline']['avg_kernel_ms'])"; done
