#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
timeout 200 python tools/tile_rect_stats.py 300000 800 800 2>&1 | tail -12
timeout 300 python tools/tile_rect_stats.py 2000000 1800 700 2>&1 | tail -12
