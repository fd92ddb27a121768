cd /root/repo
timeout 600 python -m pytest tests/test_rasterizer_gpu.py -m gpu -x -q -p no:cacheprovider -k "binned or parity or state" 2>&1 | tail -5
for b in 1 258 514 1026 2050; do echo BIN=$b; BIN=$b ITERS=10 python tools/kbench_raster.py 2>&1 | tail -1 | cut -c1-100; done
for b in 1 258 514 1026; do
  echo "BIN=$b"; R3DG_BIN=$b python bench.py --steps 40 --warmup 8 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print(d['spread_iters_per_s']['median'], d['ms_per_step'], 'shade_fwd', k['shade_forward']['ms_per_iteration'], 'sort', k['sort_pairs']['ms_per_iteration'], 'dup', k['duplicate_with_keys']['ms_per_iteration'])"
done
