cd /root/repo
timeout 900 python -m pytest tests/test_rasterizer_gpu.py tests/test_fused_step_gpu.py tests/test_fused_dp_gpu.py -m gpu -x -q -p no:cacheprovider -k "bounded or fused or dp" 2>&1 | tail -15
for b in 0 1; do
R3DG_BOUNDED=$b python bench.py --steps 40 --warmup 8 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print(d['spread_iters_per_s']['min'], d['spread_iters_per_s']['median'], d['spread_iters_per_s']['max'], d['ms_per_step'], d['config']['workload'][-30:])
print({a:b['ms_per_iteration'] for a,b in k.items()})"
done
