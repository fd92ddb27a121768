timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "ssim or loss or fused or stage2 or train_step" 2>&1 | tail -3
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --relight-frames 0"
for i in 1 2; do
  $B 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K20W5', d['value'], d['spread_iters_per_s']['min'], d['spread_iters_per_s']['median'], d['spread_iters_per_s']['max'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items() if k in ('ssim','stage2_loss')})"
done
