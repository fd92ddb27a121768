set -x
cd /root/repo
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider < /dev/null > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
export TMPDIR=/tmp
cd /tmp
CMD="python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs"
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- $CMD < /dev/null > /root/repo/gpurun_out/prof_bench.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_summary.py "$f" gpurun_out/stage2_fused_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs" < /dev/null
python tools/rocpd_timeline.py "$f" 15 < /dev/null > gpurun_out/timeline.txt 2>&1
cd /tmp
dbs=""
for c in FETCH_SIZE WRITE_SIZE; do
  for w in raster shade; do
    rm -rf /tmp/pmc_${c}_${w}
    ONLY64=1 timeout 250 rocprofv3 --pmc $c -d /tmp/pmc_${c}_${w} -o p -- python /root/repo/tools/kbench_${w}.py < /dev/null > /tmp/pmc.log 2>&1
    dbs="$dbs $(find /tmp/pmc_${c}_${w} -name '*.db' | head -1)"
  done
done
cd /root/repo
python tools/pmc_traffic.py gpurun_out/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE collected in SEPARATE passes (tools/kbench_raster.py S=16, tools/kbench_shade.py K=64; P=300000, 800x800, R~1.77M), mean per launch" $dbs < /dev/null
timeout 240 python bench.py < /dev/null > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log > gpurun_out/bench_default.json
cut -c1-200 gpurun_out/bench_default.json
