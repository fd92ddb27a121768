#!/bin/bash
# binning kernels with preloaded rectangles: parity tests, headline bench, 2M config, DTU
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
timeout 900 python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_gpu.py -q -x -p no:cacheprovider < /dev/null > gpurun_out/n_pytest.txt 2>&1; tail -3 gpurun_out/n_pytest.txt
show() { python - "$1" <<'P'
import json, sys
d=json.load(open("gpurun_out/bench_full.json"))
print(sys.argv[1], d["value"], d["spread_iters_per_s"]["median"], d["device_clock"]["shader_clock_ghz_under_valu_load"], {k:v["ms_per_iteration"] for k,v in d["kernels"].items() if k in ("duplicate_with_keys","sort_pairs","preprocess","shade_forward","render_forward")})
P
}
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 3 < /dev/null > /dev/null 2> gpurun_out/n_bench.err; show headline
timeout 300 python bench.py --points 2000000 --width 1800 --height 700 --steps 24 --warmup 6 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 2 < /dev/null > /dev/null 2> gpurun_out/n_bench2m.err; show 2M
timeout 300 python bench.py --width 1600 --height 1200 --sample-num 32 --objective syn4 --steps 60 --warmup 10 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 2 < /dev/null > /dev/null 2> gpurun_out/n_benchdtu.err; show DTU
