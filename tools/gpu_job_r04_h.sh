#!/bin/bash
# streamed smoothness kernel after a change: parity with the three passes, then its time inside the DTU configuration
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp; R=$PWD
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_fused_step_gpu.py -m gpu -q -k "smooth" 2>&1 | tail -1
rm -rf /tmp/pd; (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pd -o p -- python $R/tools/kbench_dtu.py 20 > /tmp/pd.log 2>&1) || true
f=$(find /tmp/pd -name "*.db" | head -1)
python tools/rocpd_summary.py "$f" gpurun_out/dtu_now.md "dtu" < /dev/null > /dev/null 2>&1
grep -E "smooth|s2_loss" gpurun_out/dtu_now.md | sed -E 's/\(int[^|]*//'
