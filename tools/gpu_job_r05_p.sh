#!/bin/bash
# 2M configuration: geometry backward on the early stream, binning workgroup size
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
show() { python - "$1" <<'P'
import json, sys
d=json.load(open("gpurun_out/bench_full.json"))
print(sys.argv[1], d["value"], d["spread_iters_per_s"]["median"], {k:v["ms_per_iteration"] for k,v in d["kernels"].items() if k in ("duplicate_with_keys","sort_pairs","preprocess","shade_forward","preprocess_backward","shade_frs_listed")})
P
}
A="--points 2000000 --width 1800 --height 700 --steps 24 --warmup 6 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 2"
for k in 2 3 4; do
  timeout 300 python tools/bench_with_options.py BINNING_BLOCK_K=$k -- $A < /dev/null > /dev/null 2> gpurun_out/p_err.txt; show "2M K=$k"
done
R3DG_SPLIT_GEOMETRY=0 timeout 300 python bench.py $A < /dev/null > /dev/null 2> gpurun_out/p_err.txt; show "2M auto, geometry on main"
timeout 300 python bench.py $A < /dev/null > /dev/null 2> gpurun_out/p_err.txt; show "2M auto"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 3 < /dev/null > /dev/null 2> gpurun_out/p_err.txt; show headline
timeout 600 python -m pytest tests/test_rasterizer_gpu.py tests/test_fused_step_gpu.py -q -x -p no:cacheprovider < /dev/null > gpurun_out/p_pytest.txt 2>&1; tail -2 gpurun_out/p_pytest.txt
