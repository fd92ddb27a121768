B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --relight-frames 0"
for v in 0 1 0 1; do
  R3DG_EXP_GATE=$v $B 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('GATE$v', d['value'], d['spread_iters_per_s']['min'], d['spread_iters_per_s']['median'], d['spread_iters_per_s']['max'], {k:round(v['avg_ms'],3) for k,v in d['kernels'].items() if k in ('duplicate_with_keys','sort_pairs','shade_forward','preprocess','render_forward')})"
done
for v in 0 1; do
R3DG_EXP_GATE=$v python - <<'PY'
import torch, json
from relightable3dgaussian_amd import bench_core
dev = torch.device("cuda:0")
for args in [dict(points=300000, width=800, height=800, sample_num=384), dict(points=2_000_000, width=1800, height=700, sample_num=64), dict(points=300000, width=1600, height=1200, sample_num=32, objective="syn4")]:
    r = bench_core.config_rate(dev, args.pop("points"), args.pop("width"), args.pop("height"), steps=12, warmup=4, **args)
    print("cfg", r.get("image"), r.get("sample_num"), r["iters_per_s"])
PY
done
