cd /root/repo
timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_fused_dp_gpu.py tests/test_densify_gpu.py tests/test_reference_pipeline_gpu.py tests/test_rasterizer_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --relight-frames 0 --repeats 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1])
print(d['spread_iters_per_s']['median'], d['ms_per_step'])
print({a:(b if not isinstance(b,dict) else '...') for a,b in d.get('other_configs',{}).items()})"
