set -x
cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
GA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
GB="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
dbs=""
i=0
for grp in "$GA" "$GB"; do
  i=$((i+1))
  rm -rf /tmp/pv_${i}
  ONLY64=1 timeout 250 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pv_${i} -o p -- python /root/repo/tools/kbench_shade.py < /dev/null > /tmp/pv_${i}.log 2>&1
  tail -2 /tmp/pv_${i}.log
  dbs="$dbs $(find /tmp/pv_${i} -name '*.db' | head -1)"
done
cd /root/repo
python tools/pmc_valu.py gpurun_out/pmc_shade.json "shade kernels, tools/kbench_shade.py ONLY64" $dbs < /dev/null
