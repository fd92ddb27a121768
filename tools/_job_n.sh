python - <<'PY'
import torch, json
from relightable3dgaussian_amd import bench_core
dev = torch.device("cuda:0")
for K in (384, 128):
    r = bench_core.config_rate(dev, 300000, 800, 800, sample_num=K, steps=10, warmup=3, stage_ms=True)
    print("cfg K", K, r["iters_per_s"], json.dumps(r.get("stage_ms")))
PY
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof
cat > /tmp/k384.py <<'PY'
import torch
from relightable3dgaussian_amd import bench_core
r = bench_core.config_rate(torch.device("cuda:0"), 300000, 800, 800, sample_num=384, steps=10, warmup=3)
print(r["iters_per_s"])
PY
PYTHONPATH=/root/repo timeout 300 rocprofv3 --kernel-trace -d /tmp/prof -o k -- python /tmp/k384.py > /dev/null 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd /root/repo; python tools/rocpd_timeline.py "$f" seq > gpurun_out/seq384.txt 2>&1
