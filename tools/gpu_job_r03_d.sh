#!/bin/bash
# Round 3, GPU call D: shading / rasterizer tests after the fixes (fixed-ray-set area, long-tile radix sort with equal depths),
# the 8-rank rehearsal, the long-tile sort A/B at the composition scale, the reference's unmodified train.py on hardware.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python -m pytest tests/test_shading_gpu.py tests/test_rasterizer_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/d_pytest_first.txt 2>&1
echo "pytest(shading, rasterizer) rc=$?"; tail -12 gpurun_out/d_pytest_first.txt
timeout 300 python -m pytest tests/test_fused_dp_gpu.py -q -p no:cacheprovider -k "eight or does_not_fit" < /dev/null > gpurun_out/d_pytest_dp.txt 2>&1
echo "pytest(dp) rc=$?"; grep -v "amdgpu.ids\|socket.cpp" gpurun_out/d_pytest_dp.txt | tail -12
timeout 300 python tools/kbench_sort_long.py > gpurun_out/d_sort_long.json 2> gpurun_out/d_sort_long.err
echo "sort_long rc=$?"; cat gpurun_out/d_sort_long.json; tail -3 gpurun_out/d_sort_long.err
timeout 900 python tools/reference_train_py_gpu_run.py --reference reference_scratch > gpurun_out/d_reference_train_py_gpu.txt 2>&1
echo "reference train.py rc=$?"; tail -40 gpurun_out/d_reference_train_py_gpu.txt
timeout 500 python bench.py --steps 24 --warmup 6 --no-cpu-baseline > gpurun_out/d_bench.json 2> gpurun_out/d_bench.err
echo "bench rc=$?"; tail -2 gpurun_out/d_bench.err; cut -c1-300 gpurun_out/d_bench.json
