#!/bin/bash
# Round 3, GPU call A: (1) the whole GPU suite on the new code (transport default, packed BVH records, feature-only backward,
# smoothness kernels, frozen geometry), (2) a quick bench line, (3) shade_backward ablations + LDS counters for the attribution
# of its bank conflicts, (4) rasterizer time vs feature count.      gpurun --timeout 900 -- 'bash tools/gpu_job_r03_a.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 420 python -m pytest tests -m gpu -q -x -p no:cacheprovider < /dev/null > gpurun_out/a_pytest.txt 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/a_pytest.txt
timeout 200 python bench.py --steps 24 --warmup 6 --no-cpu-baseline --repeats 2 > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err
echo "bench rc=$?"; tail -2 gpurun_out/a_bench.err; cut -c1-400 gpurun_out/a_bench.json
timeout 240 python tools/ablate_shade_backward.py run gpurun_out/a_ablate_shade_backward.json 2>&1 | tail -12
# LDS bank conflicts: product vs the variant without the LDS atomics / without the whole texture scatter
cd /tmp
GB="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
for v in product abl_no_lds_atomic abl_no_env_grad; do
  rm -rf /tmp/pl_$v
  LIBV=""
  [ "$v" != product ] && LIBV="/root/repo/relightable3dgaussian_amd/lib/variants/$v/libr3dg_hip.so"
  R3DG_LIB_PATH="$LIBV" ONLY64=1 timeout 120 rocprofv3 --pmc $GB --kernel-trace -d /tmp/pl_$v -o p -- python /root/repo/tools/kbench_shade.py < /dev/null > /tmp/pl_$v.log 2>&1
  f=$(find /tmp/pl_$v -name '*.db' | head -1)
  python /root/repo/tools/pmc_valu.py /root/repo/gpurun_out/a_pmc_lds_$v.json "tools/kbench_shade.py ONLY64=1, variant $v" $f < /dev/null | tail -1
done
cd /root/repo
for S in 0 4 16 28; do
  echo "S=$S $(S=$S ITERS=6 timeout 100 python tools/kbench_raster.py 2>&1 | tail -1)"
done > gpurun_out/a_raster_vs_S.txt 2>&1
cat gpurun_out/a_raster_vs_S.txt
