#!/bin/bash
# Round 3, GPU call J: the tile forward with decoupled waves (R3DG_OPT_FWD_DECOUPLED): parity, stand-alone A/B at 300k / 800x800 and
# at 2M / 1800x700, the iteration with it.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python -m pytest tests/test_rasterizer_gpu.py -q -p no:cacheprovider -k "decoupled" < /dev/null > gpurun_out/j_pytest.txt 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/j_pytest.txt
for d in 0 1; do
  echo "DECOUPLED=$d"
  R3DG_OPT_FWD_DECOUPLED=$d R3DG_OPT_BWD_DECOUPLED=$d ITERS=10 timeout 120 python tools/kbench_raster.py 2>/dev/null | tail -1
  R3DG_OPT_FWD_DECOUPLED=$d R3DG_OPT_BWD_DECOUPLED=$d S=0 ITERS=10 timeout 120 python tools/kbench_raster.py 2>/dev/null | tail -1
done
for d in 0 1; do
  R3DG_OPT_BWD_DECOUPLED=$d R3DG_OPT_FWD_DECOUPLED=$d timeout 300 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --no-other-configs --relight-frames 8 > gpurun_out/j_bench_$d.json 2> gpurun_out/j_bench_$d.err
  echo "bench decoupled=$d rc=$?"; python - <<PY
import json
d=json.loads(open('gpurun_out/j_bench_$d.json').read().strip().splitlines()[-1])
print(d['value'], d['spread_iters_per_s']['median'], 'render_forward', d['kernels']['render_forward']['ms_per_iteration'], 'render_backward', d['kernels']['render_backward']['ms_per_iteration'], 'shade_forward', d['kernels']['shade_forward']['ms_per_iteration'], 'relight', d.get('relight',{}).get('relight_fps'))
PY
done
