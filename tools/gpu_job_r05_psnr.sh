#!/bin/bash
# training quality at the headline size against the real reference kernels, final tree
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
R3DG_PSNR_HEADLINE=1 timeout 900 python -m pytest tests/test_psnr_vs_reference_gpu.py -q -s -k headline -p no:cacheprovider < /dev/null > gpurun_out/r05_psnr_headline_size.txt 2>&1
tail -8 gpurun_out/r05_psnr_headline_size.txt | cut -c1-220
