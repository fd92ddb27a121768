# round 5, first GPU call: the new parity tests (relight kernels vs oracle + reference fixture, per-Gaussian FRS check, counting
# trace), the pipeline test with its outlier fractions printed, smoke, and the default bench (compact last line).
set -x
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > gpurun_out/r05_smoke.txt 2>&1; tail -1 gpurun_out/r05_smoke.txt
timeout 900 python -m pytest tests/test_shading_gpu.py tests/test_relight_gpu.py -q -p no:cacheprovider -s -k "relight or fixed_ray_set_kernels_match_oracle or turns or transport" < /dev/null > gpurun_out/r05_a_relight_tests.log 2>&1; tail -5 gpurun_out/r05_a_relight_tests.log
timeout 300 python -m pytest tests/test_bvh_gpu.py -q -p no:cacheprovider -k "formulations" < /dev/null > gpurun_out/r05_a_bvh.log 2>&1; tail -3 gpurun_out/r05_a_bvh.log
timeout 600 python -m pytest tests/test_reference_pipeline_gpu.py -q -s -p no:cacheprovider < /dev/null > gpurun_out/r05_a_pipeline.log 2>&1; tail -3 gpurun_out/r05_a_pipeline.log
timeout 900 python bench.py < /dev/null > gpurun_out/r05_a_bench.out 2> gpurun_out/r05_a_bench.err; tail -c 2500 gpurun_out/r05_a_bench.out
cp gpurun_out/bench_full.json gpurun_out/r05_a_bench_full.json
python tools/bench_summary.py gpurun_out/r05_a_bench.out | head -60
