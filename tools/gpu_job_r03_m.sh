#!/bin/bash
# Round 3, GPU call M: shading tests + the default bench (K = 384 entries after the table-word change for K > 128).
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_shading_gpu.py tests/test_fused_step_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/m_pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/m_pytest.txt
timeout 500 python bench.py --steps 40 --warmup 8 --no-cpu-baseline > gpurun_out/m_bench.json 2> gpurun_out/m_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/m_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['spread_iters_per_s'])
for k,v in d['kernels'].items(): print("%-26s %.4f"%(k, v['ms_per_iteration']))
for k,v in d['other_configs'].items(): print(k[:60], {a:b for a,b in v.items() if a in ('iters_per_s','relight_fps','exposed_comm_ms')})
print(d['relight']['relight_fps'])
PY
