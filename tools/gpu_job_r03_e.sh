#!/bin/bash
# Round 3, GPU call E: the tightened comparison with the real reference build (every differing pixel explained), non-finite
# upstream gradients through the -ffast-math shading kernels, the long-tile sort A/B again (no agent-scope fences) with a
# kernel trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_shading_gpu.py -q -p no:cacheprovider -k "non_finite or fixed_ray" < /dev/null > gpurun_out/e_pytest_shading.txt 2>&1
echo "pytest(shading) rc=$?"; tail -25 gpurun_out/e_pytest_shading.txt
timeout 300 python -m pytest tests/test_rasterizer_gpu.py -q -p no:cacheprovider -k "long_tile or long_tiles" < /dev/null > gpurun_out/e_pytest_sort.txt 2>&1
echo "pytest(sort) rc=$?"; tail -5 gpurun_out/e_pytest_sort.txt
timeout 300 python tools/kbench_sort_long.py > gpurun_out/e_sort_long.json 2> gpurun_out/e_sort_long.err
echo "sort_long rc=$?"; cat gpurun_out/e_sort_long.json; tail -3 gpurun_out/e_sort_long.err
cd /tmp; rm -rf /tmp/prof_sort
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_sort -o sort -- python /root/repo/tools/kbench_sort_long.py < /dev/null > /tmp/prof_sort.log 2>&1
f=$(find /tmp/prof_sort -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_summary.py "$f" gpurun_out/e_sort_long_kernels.md "rocprofv3 --kernel-trace --stats -- python tools/kbench_sort_long.py (2 M Gaussians, 1800x700: 11 forwards per LONG_TILE_SORT setting, 1 then 0)" < /dev/null
head -30 gpurun_out/e_sort_long_kernels.md
timeout 900 python -m pytest tests/test_reference_gpu.py -q -p no:cacheprovider -s -k "rasterizer" < /dev/null > gpurun_out/e_pytest_reference.txt 2>&1
echo "pytest(reference) rc=$?"; grep -v "amdgpu.ids" gpurun_out/e_pytest_reference.txt | tail -150
