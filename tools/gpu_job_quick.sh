cd /root/repo
for r in 1 2 1 2; do
R3DG_SHADE_ROWS=$r python bench.py --steps 40 --warmup 8 --no-cpu-baseline --relight-frames 20 --no-other-configs --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print('ROWS=$r', d['spread_iters_per_s']['min'], d['spread_iters_per_s']['median'], d['spread_iters_per_s']['max'], d['ms_per_step'], 'shade_fwd', k['shade_forward']['ms_per_iteration'], 'relight fps', {a:b for a,b in d.get('relight',{}).items() if 'fps' in a})"
done
