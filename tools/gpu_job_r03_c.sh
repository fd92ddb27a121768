#!/bin/bash
# Round 3, GPU call C: shading + pipeline + relight + DP tests (incl. the 8-rank rehearsal), A/B of the fixed-ray-set variants,
# the new bench line with every BASELINE configuration.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_shading_gpu.py tests/test_reference_pipeline_gpu.py tests/test_relight_gpu.py tests/test_fused_step_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/c_pytest_first.txt 2>&1
echo "pytest(first) rc=$?"; tail -15 gpurun_out/c_pytest_first.txt
timeout 400 python -m pytest tests/test_fused_dp_gpu.py tests/test_bvh_gpu.py tests/test_rasterizer_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/c_pytest_dp.txt 2>&1
echo "pytest(dp,bvh,raster) rc=$?"; tail -12 gpurun_out/c_pytest_dp.txt
timeout 300 python tools/variants_frs.py run gpurun_out/c_variants_frs.json 2>&1 | tail -8
timeout 400 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --repeats 2 > gpurun_out/c_bench.json 2> gpurun_out/c_bench.err
echo "bench rc=$?"; tail -2 gpurun_out/c_bench.err; cut -c1-300 gpurun_out/c_bench.json
