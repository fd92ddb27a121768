cd /root/repo
for b in 0 2 3 4; do
  echo "BPC=$b"; R3DG_SHADE_FWD_BPC=$b python bench.py --steps 40 --warmup 8 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print(d['spread_iters_per_s']['median'], d['ms_per_step'], 'shade_fwd', k['shade_forward']['ms_per_iteration'], 'sort', k['sort_pairs']['ms_per_iteration'], 'dup', k['duplicate_with_keys']['ms_per_iteration'])"
done
