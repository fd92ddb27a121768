#!/bin/bash
# Round 4, round-end evidence on one box: smoke, the full GPU suite, job A (counters of the iteration's launches, kernel stats and
# timeline of the default bench command), kernel stats of the DTU configuration, the default bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
R=$PWD
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > gpurun_out/r04_smoke.txt 2>&1; tail -1 gpurun_out/r04_smoke.txt
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider < /dev/null > gpurun_out/r04_pytest_gpu.txt 2>&1; tail -2 gpurun_out/r04_pytest_gpu.txt
bash tools/gpu_job_r04_a.sh > gpurun_out/r04_job_a.log 2>&1; tail -12 gpurun_out/r04_job_a.log | cut -c1-220
rm -rf /tmp/pd; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pd -o p -- python $R/tools/kbench_dtu.py 20 > /tmp/pd.log 2>&1)
python tools/rocpd_summary.py "$(find /tmp/pd -name '*.db' | head -1)" gpurun_out/r04_dtu_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python tools/kbench_dtu.py 20  (configs[3]: 1600x1200, sample_num 32, run_dtu.sh objective, frozen geometry; 23 iterations + setup: BVH build and visibility trace)" < /dev/null > /dev/null 2>&1
timeout 900 python bench.py < /dev/null > gpurun_out/r04_bench_default.log 2>&1; tail -1 gpurun_out/r04_bench_default.log > gpurun_out/r04_bench_default.json
cut -c1-400 gpurun_out/r04_bench_default.json
