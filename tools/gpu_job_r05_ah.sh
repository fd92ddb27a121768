#!/bin/bash
# where the count kernel's 0.25 ms at 2M Gaussians go: ablation builds under rocprofv3 kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
R=$PWD; O=$R/gpurun_out
for v in product count_no_flush count_no_hist; do
  cd /tmp; rm -rf /tmp/prof
  if [ $v = product ]; then LIB=""; else LIB="$R/relightable3dgaussian_amd/lib/variants/$v/libr3dg_hip.so"; fi
  R3DG_LIB_PATH=$LIB P=2000000 W=1800 H=700 ITERS=6 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python $R/tools/kbench_binning.py < /dev/null > $O/ah_log.txt 2>&1
  f=$(find /tmp/prof -name "*.db" | head -1)
  cd $R
  python tools/rocpd_summary.py "$f" $O/ah_$v.md "kbench_binning 2M $v" < /dev/null > /dev/null 2>&1
  echo "== $v"; grep -E "tile_count|tile_emit|tile_scan|preprocess_kernel" $O/ah_$v.md | cut -c1-140
done
