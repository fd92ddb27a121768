#!/bin/bash
# Round 3, GPU call F: the two fixed tests (non-finite upstream gradients, real-reference comparison), occupancy variants of the
# fixed-ray-set forward, SQ counters of the shading kernels, the learning-rate rules under data parallelism, a bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 300 python -m pytest tests/test_shading_gpu.py tests/test_rasterizer_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/f_pytest_first.txt 2>&1
echo "pytest(shading, rasterizer) rc=$?"; tail -8 gpurun_out/f_pytest_first.txt
timeout 900 python -m pytest tests/test_reference_gpu.py -q -p no:cacheprovider -s -k "rasterizer" < /dev/null > gpurun_out/f_pytest_reference.txt 2>&1
echo "pytest(reference) rc=$?"; grep -n "unexplained\|bad \|^FAILED\|passed\|failed" gpurun_out/f_pytest_reference.txt | cut -c1-220 | tail -150
VARIANTS_ONLY=frs_fwd_occ4,frs_fwd_occ5 timeout 300 python tools/variants_frs.py run gpurun_out/f_variants_frs.json 2>&1 | tail -8
cd /tmp
GA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
GB="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
GC="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_DEP_WAIT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
dbs=""
i=0
for grp in "$GA" "$GB" "$GC"; do
  i=$((i+1))
  rm -rf /tmp/pv_${i}
  ONLY64=1 ITERS=4 timeout 250 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pv_${i} -o p -- python /root/repo/tools/kbench_shade.py < /dev/null > /tmp/pv_${i}.log 2>&1
  tail -2 /tmp/pv_${i}.log | cut -c1-300
  dbs="$dbs $(find /tmp/pv_${i} -name '*.db' | head -1)"
done
cd /root/repo
python tools/pmc_valu.py gpurun_out/f_pmc_valu_shade.json "rocprofv3 --pmc <one SQ counter group per pass> --kernel-trace on tools/kbench_shade.py (K=64, P=300000); mean per dispatch" $dbs < /dev/null
timeout 900 python tools/dp_psnr_equal_views.py --ranks 2,4 --iters 240 > gpurun_out/f_dp_psnr.txt 2> gpurun_out/f_dp_psnr.err
echo "dp_psnr rc=$?"; cat gpurun_out/f_dp_psnr.txt; grep -v "amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/f_dp_psnr.err | tail -5
timeout 500 python bench.py --steps 24 --warmup 6 --no-cpu-baseline > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err
echo "bench rc=$?"; tail -2 gpurun_out/f_bench.err; cut -c1-300 gpurun_out/f_bench.json
