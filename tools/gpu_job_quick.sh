cd /root/repo
for d in 0 1 0 1; do
R3DG_DEFER_B=$d python bench.py --steps 40 --warmup 8 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']
print('defer=$d', d['spread_iters_per_s']['min'], d['spread_iters_per_s']['median'], d['spread_iters_per_s']['max'], d['ms_per_step'])"
done
