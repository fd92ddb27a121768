# streaming smoothness kernel: the DTU iteration under both forms / rows per wave, per-kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for v in "R3DG_SMOOTH_FORM=0" "R3DG_SMOOTH_ROWS=8" "R3DG_SMOOTH_ROWS=16" "R3DG_SMOOTH_ROWS=32"; do
  i=$((i+1))
  rm -rf /tmp/pd; (cd /tmp && env $v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pd -o p -- python $GRAFT_REPO_ROOT/tools/kbench_dtu.py 20 > /tmp/pd.log 2>&1)
  f=$(find /tmp/pd -name "*.db" | head -1)
  python tools/rocpd_summary.py "$f" gpurun_out/dtu_var$i.md "dtu $v" < /dev/null > /dev/null 2>&1
done
