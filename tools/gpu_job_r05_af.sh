#!/bin/bash
# binning kernels: emit's flush atomics batched; per-kernel times at 300k and 2M (BINNING_BLOCK_K 2 / 4)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
R=$PWD; O=$R/gpurun_out
timeout 600 python -m pytest tests/test_rasterizer_gpu.py -q -x -p no:cacheprovider < /dev/null > $O/af_pytest.txt 2>&1; tail -2 $O/af_pytest.txt | cut -c1-150
prof() { # label, env..., args
  label=$1; shift
  cd /tmp; rm -rf /tmp/prof
  env "$@" R3DG_BENCH_NO_ALONE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python $R/bench.py $ARGS --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 < /dev/null > $O/af_prof.log 2>&1
  f=$(find /tmp/prof -name "*.db" | head -1)
  cd $R
  python tools/rocpd_timeline.py "$f" 8 < /dev/null 2>&1 | grep -E "wall|tile_count|tile_emit|tile_scan|tile_sort|preprocess_kernel" | sed "s/^/$label: /" | cut -c1-150
}
ARGS="--steps 20 --warmup 5"; prof "300k K=2" X=1
ARGS="--points 2000000 --width 1800 --height 700 --steps 12 --warmup 4"; prof "2M K=2" R3DG_OPT_BINNING_BLOCK_K=2
ARGS="--points 2000000 --width 1800 --height 700 --steps 12 --warmup 4"; prof "2M K=4" R3DG_OPT_BINNING_BLOCK_K=4
show() { python - "$1" <<'P'
import json, sys
d=json.load(open("gpurun_out/bench_full.json"))
print(sys.argv[1], d["value"], d["spread_iters_per_s"]["median"], {k:v["ms_per_iteration"] for k,v in d["kernels"].items() if k in ("duplicate_with_keys","sort_pairs")})
P
}
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 3 < /dev/null > /dev/null 2> $O/af_err.txt; show headline
timeout 300 python bench.py --points 2000000 --width 1800 --height 700 --steps 24 --warmup 6 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 2 < /dev/null > /dev/null 2> $O/af_err.txt; show 2M
