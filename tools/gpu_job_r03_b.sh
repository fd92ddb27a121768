#!/bin/bash
# Round 3, GPU call B: the suite on the fixed-ray-set shading kernels, list mode, option API, sRGB fix of the smoothness terms;
# stand-alone shading times (general vs fixed-ray-set) + rocprofv3 kernel stats of that run; the default bench.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 200 python -m pytest tests/test_shading_gpu.py tests/test_reference_pipeline_gpu.py -q -x -p no:cacheprovider < /dev/null > gpurun_out/b_pytest_first.txt 2>&1
echo "pytest(first) rc=$?"; tail -15 gpurun_out/b_pytest_first.txt
timeout 500 python -m pytest tests -m gpu -q -p no:cacheprovider --deselect tests/test_shading_gpu.py --deselect tests/test_reference_pipeline_gpu.py < /dev/null > gpurun_out/b_pytest_rest.txt 2>&1
echo "pytest(rest) rc=$?"; tail -12 gpurun_out/b_pytest_rest.txt
timeout 120 python tools/kbench_shade.py > gpurun_out/b_kbench_shade.txt 2>&1; cat gpurun_out/b_kbench_shade.txt | tail -4
cd /tmp; rm -rf /tmp/pk
ONLY64=1 timeout 120 rocprofv3 --kernel-trace --stats -d /tmp/pk -o p -- python /root/repo/tools/kbench_shade.py < /dev/null > /tmp/pk.log 2>&1
f=$(find /tmp/pk -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_summary.py "$f" gpurun_out/b_shade_kernel_stats.md "rocprofv3 --kernel-trace --stats -- ONLY64=1 python tools/kbench_shade.py" < /dev/null; head -30 gpurun_out/b_shade_kernel_stats.md
timeout 250 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --repeats 2 > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err
echo "bench rc=$?"; tail -2 gpurun_out/b_bench.err; cut -c1-300 gpurun_out/b_bench.json
R3DG_SHADE_FRS=0 timeout 200 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --repeats 2 --no-other-configs --relight-frames 0 > gpurun_out/b_bench_nofrs.json 2> gpurun_out/b_bench_nofrs.err
cut -c1-200 gpurun_out/b_bench_nofrs.json
