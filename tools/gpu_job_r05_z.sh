#!/bin/bash
# chain late = 0 / 1 / 2 with the chain kernel, capped / uncapped shading forward; parity; sequence
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
R=$PWD; O=$R/gpurun_out
timeout 600 python -m pytest tests/test_shading_gpu.py -q -x -p no:cacheprovider -k "incident_chain" < /dev/null > $O/z_pytest1.txt 2>&1; tail -2 $O/z_pytest1.txt | cut -c1-200
show() { python - "$1" <<'P'
import json, sys
d=json.load(open("gpurun_out/bench_full.json"))
print(sys.argv[1], d["value"], d["spread_iters_per_s"]["median"], d["device_clock"]["shader_clock_ghz_under_valu_load"], {k:v["ms_per_iteration"] for k,v in d["kernels"].items() if k in ("adam_step","shade_frs_aux","shade_forward","duplicate_with_keys","sort_pairs","stage2_activate","preprocess")})
P
}
A="--steps 200 --warmup 20 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 3"
for l in 1 2; do for lr in auto 1; do for k in 1 0; do
  R3DG_INCIDENT_CHAIN_KERNEL=$k R3DG_INCIDENT_CHAIN_LATE=$l R3DG_SHADE_LEAVE_ROOM=$lr timeout 300 python bench.py $A < /dev/null > /dev/null 2> $O/z_err.txt; show "kernel=$k late=$l leave_room=$lr"
done; done; done
timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_reference_pipeline_gpu.py tests/test_psnr_vs_reference_gpu.py tests/test_fused_dp_gpu.py tests/test_train_loop_gpu.py -q -x -p no:cacheprovider < /dev/null > $O/z_pytest2.txt 2>&1; tail -3 $O/z_pytest2.txt | cut -c1-200
cd /tmp; rm -rf /tmp/prof
R3DG_BENCH_NO_ALONE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 < /dev/null > $O/z_prof.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd $R
python tools/rocpd_timeline.py "$f" 15 < /dev/null > $O/z_timeline.txt 2>&1
python tools/rocpd_timeline.py "$f" seq < /dev/null > $O/z_sequence.txt 2>&1
head -5 $O/z_timeline.txt | cut -c1-160
cut -c1-130 $O/z_sequence.txt | sed -n 8,26p
