B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --relight-frames 0"
for v in dev sys dev sys; do
  if [ $v = sys ]; then export R3DG_EXP_EVENT_SYSTEM=1; else unset R3DG_EXP_EVENT_SYSTEM; fi
  $B 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('EV_$v', d['value'], d['spread_iters_per_s']['min'], d['spread_iters_per_s']['median'], d['spread_iters_per_s']['max'])"
done
unset R3DG_EXP_EVENT_SYSTEM
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "fused or rasterizer or pipeline" 2>&1 | tail -3
