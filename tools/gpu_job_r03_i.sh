#!/bin/bash
# Round 3, GPU call I: atomic-free direct tile binning: rasterizer / fused / DP tests, stage times at 300k and 2M, a bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_rasterizer_gpu.py tests/test_fused_step_gpu.py tests/test_shading_gpu.py tests/test_relight_gpu.py tests/test_densify_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/i_pytest_first.txt 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/i_pytest_first.txt
ITERS=10 timeout 120 python tools/kbench_raster.py > gpurun_out/i_kbench_raster.txt 2>&1; tail -1 gpurun_out/i_kbench_raster.txt
timeout 200 python tools/kbench_sort_long.py > gpurun_out/i_sort_long.json 2>/dev/null; cat gpurun_out/i_sort_long.json
timeout 500 python bench.py --steps 24 --warmup 6 --no-cpu-baseline > gpurun_out/i_bench.json 2> gpurun_out/i_bench.err
echo "bench rc=$?"; tail -1 gpurun_out/i_bench.err; cut -c1-260 gpurun_out/i_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/i_bench.json').read().strip().splitlines()[-1])
print(d['spread_iters_per_s'])
for k,v in d['kernels'].items(): print("%-28s %.4f x%s"%(k, v['ms_per_iteration'], v['launches_per_iteration']))
for k,v in d['other_configs'].items(): print(k[:50], v.get('iters_per_s'), v.get('relight_fps'))
PY
