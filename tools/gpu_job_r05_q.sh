#!/bin/bash
# shading forward on the chain's stream: A/B + parity tests + sequence
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
R=$PWD; O=$R/gpurun_out
show() { python - "$1" <<'P'
import json, sys
d=json.load(open("gpurun_out/bench_full.json"))
print(sys.argv[1], d["value"], d["spread_iters_per_s"]["median"], d["device_clock"]["shader_clock_ghz_under_valu_load"], {k:v["ms_per_iteration"] for k,v in d["kernels"].items() if k in ("duplicate_with_keys","sort_pairs","preprocess","shade_forward","shade_frs_listed","render_forward")})
P
}
A="--steps 200 --warmup 20 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 3"
for v in 0 1 0 1; do
  R3DG_SHADE_FWD_ON_EARLY=$v timeout 300 python bench.py $A < /dev/null > /dev/null 2> $O/q_err.txt; show "fwd_on_early=$v"
done
timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_reference_pipeline_gpu.py tests/test_psnr_vs_reference_gpu.py -q -x -p no:cacheprovider < /dev/null > $O/q_pytest.txt 2>&1; tail -2 $O/q_pytest.txt
cd /tmp; rm -rf /tmp/prof
R3DG_BENCH_NO_ALONE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 < /dev/null > $O/q_prof.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd $R
python tools/rocpd_timeline.py "$f" 15 < /dev/null > $O/q_timeline.txt 2>&1
python tools/rocpd_timeline.py "$f" seq < /dev/null > $O/q_sequence.txt 2>&1
head -8 $O/q_timeline.txt | cut -c1-160
