# Round-end evidence job (MI355X, via gpurun): full GPU suite, parity logs against the real reference build, rocprofv3 kernel stats
# (training iteration and relight frame), PMC passes (SQ counters and HBM traffic: ONE counter group per pass, --kernel-trace only),
# the default bench line.  Outputs land in gpurun_out/; the round's summaries are copied to profiles/ (see profiles/README.md).
set -x
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8      # (threads: the container's CPU quota, see bench.py)
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider < /dev/null > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 600 python -m pytest tests/test_reference_gpu.py -q -s -p no:cacheprovider -k "baseline_sizes or bvh" < /dev/null > gpurun_out/parity_reference.log 2>&1; tail -2 gpurun_out/parity_reference.log
timeout 600 python -m pytest tests/test_reference_pipeline_gpu.py tests/test_psnr_vs_reference_gpu.py -q -s -p no:cacheprovider < /dev/null > gpurun_out/parity_pipeline.log 2>&1; tail -2 gpurun_out/parity_pipeline.log
cd /tmp
CMD="python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0"
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- $CMD < /dev/null > /root/repo/gpurun_out/prof_bench.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_summary.py "$f" gpurun_out/stage2_fused_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0" < /dev/null
python tools/rocpd_timeline.py "$f" 15 < /dev/null > gpurun_out/timeline.txt 2>&1
python tools/rocpd_timeline.py "$f" seq < /dev/null > gpurun_out/sequence.txt 2>&1
# the same for the data-parallel path over a one-rank RCCL group
cd /tmp
rm -rf /tmp/prof_dp
R3DG_DP_SINGLE_RANK=1 R3DG_DIST_BACKEND=nccl timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_dp -o bench -- $CMD < /dev/null > /root/repo/gpurun_out/prof_bench_dp.log 2>&1
fd=$(find /tmp/prof_dp -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_timeline.py "$fd" 15 < /dev/null > gpurun_out/timeline_dp.txt 2>&1
python tools/rocpd_timeline.py "$fd" seq < /dev/null > gpurun_out/sequence_dp.txt 2>&1
cd /tmp
CMD2="python /root/repo/bench.py --steps 2 --warmup 1 --no-cpu-baseline --relight-frames 40 --no-other-configs --repeats 0"
rm -rf /tmp/prof2
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o bench -- $CMD2 < /dev/null > /root/repo/gpurun_out/prof_relight.log 2>&1
f2=$(find /tmp/prof2 -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_summary.py "$f2" gpurun_out/relight_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --relight-frames 40 --no-other-configs --repeats 0  (43 relight frames at K=384, S=28 + the visibility trace + 3 training steps)" < /dev/null
cd /tmp
# HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes
dbs=""
for c in FETCH_SIZE WRITE_SIZE; do
  for w in raster shade; do
    rm -rf /tmp/pmc_${c}_${w}
    ONLY64=1 ITERS=4 timeout 250 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${c}_${w} -o p -- python /root/repo/tools/kbench_${w}.py < /dev/null > /tmp/pmc.log 2>&1
    dbs="$dbs $(find /tmp/pmc_${c}_${w} -name '*.db' | head -1)"
  done
done
cd /root/repo
python tools/pmc_traffic.py gpurun_out/pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE collected in SEPARATE passes (tools/kbench_raster.py S=16, tools/kbench_shade.py K=64; P=300000, 800x800, R~1.77M), mean per launch" $dbs < /dev/null
cd /tmp
# SQ counters: one group per pass
GA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
GB="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
GC="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
dbs=""
i=0
for grp in "$GA" "$GB" "$GC"; do
  i=$((i+1))
  for w in raster shade; do
    rm -rf /tmp/pv_${i}_${w}
    ONLY64=1 ITERS=4 timeout 250 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pv_${i}_${w} -o p -- python /root/repo/tools/kbench_${w}.py < /dev/null > /tmp/pv.log 2>&1
    dbs="$dbs $(find /tmp/pv_${i}_${w} -name '*.db' | head -1)"
  done
done
cd /root/repo
python tools/kernel_resources.py gpurun_out/kernel_resources.json < /dev/null
python tools/pmc_valu.py gpurun_out/pmc_valu.json "rocprofv3 --pmc <one SQ counter group per pass> --kernel-trace on tools/kbench_raster.py (S=16) and tools/kbench_shade.py (K=64); P=300000, 800x800, R~1.77M; mean per dispatch" --resources gpurun_out/kernel_resources.json $dbs < /dev/null
cp gpurun_out/pmc_valu.json profiles/r03_pmc_valu.json; cp gpurun_out/pmc_traffic.json profiles/r03_pmc_traffic.json   # (the bench line below quotes them)
timeout 700 python bench.py < /dev/null > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log > gpurun_out/bench_default.json
cut -c1-300 gpurun_out/bench_default.json
cp profiles/r03_pmc_valu.json gpurun_out/r03_pmc_valu.json
