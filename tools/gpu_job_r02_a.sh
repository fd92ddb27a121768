# round-2 GPU job A: new parity tests (real reference at BASELINE sizes, render_equation.cu pin, bench --gpus 2), PMC VALU/LDS
# evidence for the four tile/shading kernels, FETCH/WRITE calibration, default bench line.
set -x
cd /root/repo
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_reference_gpu.py tests/test_render_equation_gpu.py tests/test_fused_dp_gpu.py -q -s -p no:cacheprovider < /dev/null > gpurun_out/a_pytest_new.log 2>&1; tail -5 gpurun_out/a_pytest_new.log
export TMPDIR=/tmp
cd /tmp
GA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
GB="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
dbs=""
i=0
for grp in "$GA" "$GB"; do
  i=$((i+1))
  for w in raster shade; do
    rm -rf /tmp/pv_${i}_${w}
    ONLY64=1 ITERS=4 timeout 250 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pv_${i}_${w} -o p -- python /root/repo/tools/kbench_${w}.py < /dev/null > /tmp/pv_${i}_${w}.log 2>&1
    tail -2 /tmp/pv_${i}_${w}.log
    dbs="$dbs $(find /tmp/pv_${i}_${w} -name '*.db' | head -1)"
  done
done
cd /root/repo
python tools/kernel_resources.py gpurun_out/a_kernel_resources.json < /dev/null
python tools/pmc_valu.py gpurun_out/a_pmc_valu.json "rocprofv3 --pmc <one SQ counter group per pass> --kernel-trace on tools/kbench_raster.py (S=16) and tools/kbench_shade.py (K=64); P=300000, 800x800, R~1.77M; mean per dispatch" --resources gpurun_out/a_kernel_resources.json $dbs < /dev/null
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/cal_$c
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/cal_$c -o p -- /root/repo/tools/pmc_calibration < /dev/null > /tmp/cal_$c.log 2>&1
  tail -1 /tmp/cal_$c.log
done
cd /root/repo
python tools/pmc_calibration.py gpurun_out/a_pmc_calibration.json $(find /tmp/cal_FETCH_SIZE -name '*.db' | head -1) $(find /tmp/cal_WRITE_SIZE -name '*.db' | head -1) < /dev/null > gpurun_out/a_pmc_calibration.log 2>&1
timeout 420 python bench.py < /dev/null > gpurun_out/a_bench_default.log 2>&1; tail -1 gpurun_out/a_bench_default.log > gpurun_out/a_bench_default.json
cut -c1-300 gpurun_out/a_bench_default.json
