#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_rasterizer_gpu.py tests/test_fused_step_gpu.py tests/test_reference_pipeline_gpu.py -q -p no:cacheprovider -k "features or frozen or syn4 or fused" < /dev/null > gpurun_out/n_pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/n_pytest.txt
timeout 500 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --relight-frames 0 > gpurun_out/n_bench.json 2> gpurun_out/n_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/n_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['spread_iters_per_s'])
for k,v in d['other_configs'].items(): print(k[:60], {a:b for a,b in v.items() if a in ('iters_per_s','relight_fps','stage_ms')})
PY
