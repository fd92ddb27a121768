#!/bin/bash
# Round 3, GPU call G: the fixed-ray-set kernels with the next group's data staged by LDS-DMA and the prefetch issued after the
# table words: parity tests, A/B against the committed kernels (variants), kernel trace, a bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 400 python -m pytest tests/test_shading_gpu.py tests/test_fused_step_gpu.py tests/test_reference_pipeline_gpu.py tests/test_relight_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/g_pytest_first.txt 2>&1
echo "pytest(shading, fused, pipeline, relight) rc=$?"; tail -12 gpurun_out/g_pytest_first.txt
timeout 400 python tools/variants_frs.py run gpurun_out/g_variants_frs.json 2>&1 | tail -8
cd /tmp; rm -rf /tmp/pk
ONLY64=1 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pk -o p -- python /root/repo/tools/kbench_shade.py < /dev/null > /tmp/pk.log 2>&1
f=$(find /tmp/pk -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_summary.py "$f" gpurun_out/g_shade_kernel_stats.md "rocprofv3 --kernel-trace --stats -- ONLY64=1 python tools/kbench_shade.py" < /dev/null; head -22 gpurun_out/g_shade_kernel_stats.md | cut -c1-160
timeout 300 python -m pytest tests/test_reference_gpu.py -q -p no:cacheprovider -s -k "dtu" < /dev/null > gpurun_out/g_pytest_reference.txt 2>&1
echo "pytest(reference dtu) rc=$?"; grep -n "unexplained\|^FAILED\|passed\|failed" gpurun_out/g_pytest_reference.txt | cut -c1-200 | tail -12
timeout 500 python bench.py --steps 24 --warmup 6 --no-cpu-baseline > gpurun_out/g_bench.json 2> gpurun_out/g_bench.err
echo "bench rc=$?"; tail -2 gpurun_out/g_bench.err; cut -c1-300 gpurun_out/g_bench.json
