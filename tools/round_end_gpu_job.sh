#!/bin/bash
# Round-end evidence job (MI355X, via gpurun): smoke, the full GPU suite, parity logs against the real reference build, rocprofv3
# kernel stats + timeline (training iteration, relight frame), PMC passes -- HBM traffic (FETCH_SIZE / WRITE_SIZE, ONE counter per
# pass) and SQ counters (ONE group per pass), every pass with --kernel-trace only -- on the kernels as the ITERATION launches them
# (tools/kbench_raster.py ACTIVE=2,3,4 NODEPTH=1; tools/kbench_shade.py) and on the relight frame, the tile backward's ablation
# builds, the default bench line.  Outputs: gpurun_out/<round>_*; the summaries to be judged are copied to profiles/ afterwards.
#   ROUND=r06 bash tools/round_end_gpu_job.sh [quick]      ("quick": no full pytest run)
RD=${ROUND:-r06}
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8      # (threads: the container's CPU quota, see bench.py)
R=$PWD
O=$R/gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $O/${RD}_smoke.txt 2>&1; tail -1 $O/${RD}_smoke.txt
if [ "$1" != quick ]; then
  timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider < /dev/null > $O/${RD}_pytest_gpu.txt 2>&1; tail -3 $O/${RD}_pytest_gpu.txt
  timeout 600 python -m pytest tests/test_reference_gpu.py -q -s -p no:cacheprovider -k "baseline_sizes or bvh" < /dev/null > $O/${RD}_parity_vs_real_reference_baseline_sizes.txt 2>&1; tail -2 $O/${RD}_parity_vs_real_reference_baseline_sizes.txt
  timeout 600 python -m pytest tests/test_reference_pipeline_gpu.py tests/test_psnr_vs_reference_gpu.py tests/test_relight_gpu.py tests/test_shading_gpu.py -q -s -p no:cacheprovider -k "pipeline or psnr or reference_python or relight_kernels or fixed_ray_set_kernels_match_oracle" < /dev/null > $O/${RD}_pipeline_relight_and_psnr_vs_reference.txt 2>&1; tail -2 $O/${RD}_pipeline_relight_and_psnr_vs_reference.txt
fi
# ---- kernel stats + timeline of the training iteration (the one-stream "alone" pass of bench.py is left out of the trace)
cd /tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0"
rm -rf /tmp/prof
R3DG_BENCH_NO_ALONE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- $CMD < /dev/null > $O/prof_bench.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd $R
python tools/rocpd_summary.py "$f" $O/${RD}_stage2_fused_bench_kernel_stats.md "R3DG_BENCH_NO_ALONE=1 rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0" < /dev/null
python tools/rocpd_timeline.py "$f" 15 < /dev/null > $O/${RD}_stage2_fused_step_timeline.txt 2>&1
python tools/rocpd_timeline.py "$f" seq < /dev/null > $O/${RD}_stage2_fused_step_sequence.txt 2>&1
# ---- the relight frame
cd /tmp
CMD2="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --relight-frames 40 --no-other-configs --repeats 0"
rm -rf /tmp/prof2
R3DG_BENCH_NO_ALONE=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof2 -o bench -- $CMD2 < /dev/null > $O/prof_relight.log 2>&1
f2=$(find /tmp/prof2 -name "*.db" | head -1)
cd $R
python tools/rocpd_summary.py "$f2" $O/${RD}_relight_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --relight-frames 40 --no-other-configs --repeats 0  (relight frames at K=384, S=28 + the visibility trace + 3 training steps)" < /dev/null
# ---- HBM traffic + SQ counters: training kernels
cd /tmp
GA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
GB="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
GC="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
dbs=""
for c in FETCH_SIZE WRITE_SIZE; do
  for w in raster shade; do
    rm -rf /tmp/pmc_${c}_${w}
    ACTIVE=2,3,4 NODEPTH=1 ONLY64=1 ITERS=4 timeout 250 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${c}_${w} -o p -- python $R/tools/kbench_${w}.py < /dev/null > /tmp/pmc.log 2>&1
    dbs="$dbs $(find /tmp/pmc_${c}_${w} -name '*.db' | head -1)"
  done
done
cd $R
python tools/pmc_traffic.py $O/${RD}_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE collected in SEPARATE passes (tools/kbench_raster.py S=16 with ACTIVE=2,3,4 NODEPTH=1 -- the backward as the training iteration launches it: three live feature channels, no depth gradient; tools/kbench_shade.py K=64; P=300000, 800x800, R~1.77M), mean per launch" $dbs < /dev/null
cd /tmp
dbs=""
i=0
for grp in "$GA" "$GB" "$GC"; do
  i=$((i+1))
  for w in raster shade; do
    rm -rf /tmp/pv_${i}_${w}
    ACTIVE=2,3,4 NODEPTH=1 ONLY64=1 ITERS=4 timeout 250 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pv_${i}_${w} -o p -- python $R/tools/kbench_${w}.py < /dev/null > /tmp/pv.log 2>&1
    dbs="$dbs $(find /tmp/pv_${i}_${w} -name '*.db' | head -1)"
  done
done
cd $R
python tools/kernel_resources.py $O/${RD}_kernel_resources.json < /dev/null
python tools/pmc_valu.py $O/${RD}_pmc_valu.json "rocprofv3 --pmc <one SQ counter group per pass> --kernel-trace on tools/kbench_raster.py (S=16, ACTIVE=2,3,4 NODEPTH=1: the backward as the training iteration launches it) and tools/kbench_shade.py (K=64); P=300000, 800x800, R~1.77M; mean per dispatch" --resources $O/${RD}_kernel_resources.json $dbs < /dev/null
# ---- the same for the relight frame (bench.py's relight part)
cd /tmp
CMD3="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --repeats 0 --relight-frames 8"
dbs=""; tdbs=""
i=0
for grp in "$GA" "$GB"; do
  i=$((i+1))
  rm -rf /tmp/pr_$i
  R3DG_BENCH_NO_ALONE=1 timeout 280 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pr_$i -o p -- $CMD3 < /dev/null > /tmp/pr.log 2>&1
  dbs="$dbs $(find /tmp/pr_$i -name '*.db' | head -1)"
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$c
  R3DG_BENCH_NO_ALONE=1 timeout 280 rocprofv3 --pmc $c --kernel-trace -d /tmp/pt_$c -o p -- $CMD3 < /dev/null > /tmp/pr.log 2>&1
  tdbs="$tdbs $(find /tmp/pt_$c -name '*.db' | head -1)"
done
cd $R
python tools/pmc_valu.py $O/${RD}_pmc_valu_relight.json "rocprofv3 --pmc <one SQ group per pass> --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --repeats 0 --relight-frames 8 (K=384, 300k Gaussians, 256x512 HDR map)" $dbs < /dev/null
python tools/pmc_traffic.py $O/${RD}_pmc_traffic_relight.json "the same command, FETCH_SIZE / WRITE_SIZE in separate passes" $tdbs < /dev/null
# ---- attribution of the tile backward's time (ablation builds; only if they were built: python tools/variants_bwd.py build)
if [ -d relightable3dgaussian_amd/lib/variants/bwd_no_atomics ]; then
  timeout 300 python tools/variants_bwd.py run > $O/${RD}_bwd_ablation_product.txt 2>&1; cat $O/${RD}_bwd_ablation_product.txt | cut -c1-200
fi
# ---- Adam cold (the last-level cache evicted between launches) and the L2 / fabric request counters of the tile kernels
timeout 200 python tools/kbench_adam.py < /dev/null > $O/${RD}_adam_cold.txt 2>&1; tail -2 $O/${RD}_adam_cold.txt
ROUND=$RD bash tools/gpu_job.sh pmc l2_raster "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITE_sum TCC_WRITEBACK_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" -- env ACTIVE=2,3,4 NODEPTH=1 ITERS=4 python $R/tools/kbench_raster.py > /dev/null 2>&1
# ---- the default bench line (the counter files the line quotes must be the ones just collected: they are read from profiles/)
cp $O/${RD}_pmc_valu.json $O/${RD}_pmc_traffic.json $O/${RD}_pmc_valu_relight.json $O/${RD}_pmc_traffic_relight.json profiles/
timeout 900 python bench.py < /dev/null > $O/${RD}_bench_default.out 2> $O/${RD}_bench_default.err; tail -1 $O/${RD}_bench_default.out > $O/${RD}_bench_default_compact.json
cp $O/bench_full.json $O/${RD}_bench_default.json
cut -c1-600 $O/${RD}_bench_default_compact.json
python tools/bench_summary.py $O/${RD}_bench_default.json | head -40
