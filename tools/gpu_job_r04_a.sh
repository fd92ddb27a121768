#!/bin/bash
# Round 4, GPU job A: counter evidence for the kernels as the ITERATION launches them: rasterizer backward with
# the three feature channels the run_nerf.sh objective carries (tools/kbench_raster.py ACTIVE=2,3,4), the direction-free
# fixed-ray-set shading kernels (tools/kbench_shade.py).  HBM traffic (FETCH_SIZE / WRITE_SIZE, one counter per pass), SQ groups
# (one group per pass), each pass with --kernel-trace only; then kernel stats + timeline of the default bench command.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
R=$PWD
cd /tmp
dbs=""
for c in FETCH_SIZE WRITE_SIZE; do
  for w in raster shade; do
    rm -rf /tmp/pmc_${c}_${w}
    ACTIVE=2,3,4 ONLY64=1 ITERS=4 timeout 250 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_${c}_${w} -o p -- python $R/tools/kbench_${w}.py < /dev/null > /tmp/pmc.log 2>&1
    dbs="$dbs $(find /tmp/pmc_${c}_${w} -name '*.db' | head -1)"
  done
done
cd $R
python tools/pmc_traffic.py gpurun_out/r04_pmc_traffic.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE collected in SEPARATE passes (tools/kbench_raster.py S=16 with ACTIVE=2,3,4 -- the backward carries the three feature channels of the run_nerf.sh objective, as the training iteration launches it; tools/kbench_shade.py K=64; P=300000, 800x800, R~1.77M), mean per launch" $dbs < /dev/null
cd /tmp
GA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
GB="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
GC="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
dbs=""
i=0
for grp in "$GA" "$GB" "$GC"; do
  i=$((i+1))
  for w in raster shade; do
    rm -rf /tmp/pv_${i}_${w}
    ACTIVE=2,3,4 ONLY64=1 ITERS=4 timeout 250 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pv_${i}_${w} -o p -- python $R/tools/kbench_${w}.py < /dev/null > /tmp/pv.log 2>&1
    dbs="$dbs $(find /tmp/pv_${i}_${w} -name '*.db' | head -1)"
  done
done
cd $R
python tools/kernel_resources.py gpurun_out/r04_kernel_resources.json < /dev/null
python tools/pmc_valu.py gpurun_out/r04_pmc_valu.json "rocprofv3 --pmc <one SQ counter group per pass> --kernel-trace on tools/kbench_raster.py (S=16, ACTIVE=2,3,4: the backward as the training iteration launches it) and tools/kbench_shade.py (K=64); P=300000, 800x800, R~1.77M; mean per dispatch" --resources gpurun_out/r04_kernel_resources.json $dbs < /dev/null
cd /tmp
CMD="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0"
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- $CMD < /dev/null > $R/gpurun_out/prof_bench.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd $R
python tools/rocpd_summary.py "$f" gpurun_out/r04_stage2_fused_bench_kernel_stats.md "rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0" < /dev/null
python tools/rocpd_timeline.py "$f" 15 < /dev/null > gpurun_out/r04_stage2_fused_step_timeline.txt 2>&1
python tools/rocpd_timeline.py "$f" seq < /dev/null > gpurun_out/r04_stage2_fused_step_sequence.txt 2>&1
python - <<'PY'
import json
t = json.load(open("gpurun_out/r04_pmc_traffic.json"))["kernels"]
v = json.load(open("gpurun_out/r04_pmc_valu.json"))["kernels"]
for k in ("shade_forward_frs_kernel", "shade_backward_frs_kernel", "render_backward_wave_kernel", "render_forward_wave_kernel",
          "tile_emit_kernel", "tile_sort_small_kernel", "tile_count_kernel", "frs_rotate_kernel"):
    a, b = t.get(k, {}), v.get(k, {})
    print(k, "raw MB %.1f corrected MB %.1f" % (a.get("hbm_bytes_raw", 0) / 1e6, a.get("hbm_bytes_corrected", 0) / 1e6),
          {x: b.get(x) for x in ("duration_us_under_pmc", "valu_busy_frac", "waves_per_simd", "wait_frac", "issue_stall_frac", "clock_ghz")},
          "SQ_INSTS_VALU", b.get("counters_mean_per_dispatch", {}).get("SQ_INSTS_VALU"))
PY
head -30 gpurun_out/r04_stage2_fused_bench_kernel_stats.md | cut -c1-200
