cd /root/repo
timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_fused_dp_gpu.py tests/test_reference_pipeline_gpu.py tests/test_psnr_vs_reference_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --relight-frames 0 --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print(d['spread_iters_per_s']['min'], d['spread_iters_per_s']['median'], d['spread_iters_per_s']['max'], d['ms_per_step'])
print({a:b['ms_per_iteration'] for a,b in k.items()})
print({a:(b.get('value') if isinstance(b,dict) else b) for a,b in d.get('other_configs',{}).items()})"
