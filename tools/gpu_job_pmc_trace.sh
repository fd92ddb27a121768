set -x
cd /root/repo; mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp; export MODES=4,3
GA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
GB="SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT GRBM_GUI_ACTIVE"
GC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
GD="TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE"
GE="SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE"
dbs=""
i=0
for grp in "$GA" "$GB" "$GC" "$GD" "$GE"; do
  i=$((i+1))
  rm -rf /tmp/pt_${i}
  timeout 250 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pt_${i} -o p -- python /root/repo/tools/kbench_trace.py < /dev/null > /tmp/pt_${i}.log 2>&1
  tail -3 /tmp/pt_${i}.log
  dbs="$dbs $(find /tmp/pt_${i} -name '*.db' | head -1)"
done
cd /root/repo
python tools/pmc_valu.py gpurun_out/pmc_trace.json "trace kernels, tools/kbench_trace.py K=64" $dbs < /dev/null
