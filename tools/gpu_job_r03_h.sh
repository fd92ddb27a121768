#!/bin/bash
# Round 3, GPU call H: fixed-ray-set kernels with the device-wide group queue: parity tests, A/B against the committed kernels,
# SQ counters, two bench lines (the timed block of call G was an outlier: 340 vs 596 it/s in the five blocks after it).
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python -m pytest tests/test_shading_gpu.py tests/test_fused_step_gpu.py tests/test_reference_pipeline_gpu.py tests/test_relight_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/h_pytest_first.txt 2>&1
echo "pytest(shading, fused, pipeline, relight) rc=$?"; tail -4 gpurun_out/h_pytest_first.txt
timeout 400 python tools/variants_frs.py run gpurun_out/h_variants_frs.json 2>&1 | tail -4
cd /tmp
GA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
rm -rf /tmp/pv_1
ONLY64=1 ITERS=4 timeout 250 rocprofv3 --pmc $GA --kernel-trace -d /tmp/pv_1 -o p -- python /root/repo/tools/kbench_shade.py < /dev/null > /tmp/pv_1.log 2>&1
cd /root/repo
python tools/pmc_valu.py gpurun_out/h_pmc_valu_shade.json "rocprofv3 --pmc (one SQ group) --kernel-trace on tools/kbench_shade.py (K=64, P=300000); mean per dispatch" $(find /tmp/pv_1 -name '*.db' | head -1) < /dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/h_pmc_valu_shade.json'))['kernels']
for k in ('shade_forward_frs_kernel','shade_backward_frs_kernel'):
    r=d.get(k,{})
    print(k, {a:r.get(a) for a in ('duration_us_under_pmc','valu_busy_frac','waves_per_simd','wait_frac','issue_stall_frac')})
PY
for i in 1 2; do
timeout 500 python bench.py --steps 24 --warmup 6 --no-cpu-baseline > gpurun_out/h_bench_$i.json 2> gpurun_out/h_bench_$i.err
echo "bench $i rc=$?"; tail -1 gpurun_out/h_bench_$i.err; cut -c1-260 gpurun_out/h_bench_$i.json
done
