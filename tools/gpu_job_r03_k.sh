#!/bin/bash
# Round 3, GPU call K: coefficient rotation with the row in registers (no LDS staging): parity + kernel times.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_shading_gpu.py tests/test_fused_step_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/k_pytest.txt 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/k_pytest.txt
cd /tmp; rm -rf /tmp/pk
ONLY64=1 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pk -o p -- python /root/repo/tools/kbench_shade.py < /dev/null > /tmp/pk.log 2>&1
tail -2 /tmp/pk.log | cut -c1-900
f=$(find /tmp/pk -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_summary.py "$f" gpurun_out/k_shade_kernel_stats.md "rocprofv3 --kernel-trace --stats -- ONLY64=1 python tools/kbench_shade.py" < /dev/null; grep "frs_" gpurun_out/k_shade_kernel_stats.md | cut -c1-150
timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-other-configs --relight-frames 0 > gpurun_out/k_bench.json 2> gpurun_out/k_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/k_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['spread_iters_per_s'], 'aux', d['kernels']['shade_frs_aux']['ms_per_iteration'])
PY
