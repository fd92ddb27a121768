#!/bin/bash
# Round 4: SQ counters of the DTU configuration's image-space kernels (streamed smoothness kernel, loss, SSIM), groups A and B.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
R=$PWD
cd /tmp
GA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
GB="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
dbs=""
i=0
for grp in "$GA" "$GB"; do
  i=$((i+1))
  rm -rf /tmp/pd_$i
  timeout 280 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pd_$i -o p -- python $R/tools/kbench_dtu.py 6 < /dev/null > /tmp/pd.log 2>&1
  dbs="$dbs $(find /tmp/pd_$i -name '*.db' | head -1)"
done
cd $R
python tools/pmc_valu.py gpurun_out/r04_pmc_valu_dtu.json "rocprofv3 --pmc <one SQ group per pass> --kernel-trace -- python tools/kbench_dtu.py 6  (configs[3]: 1600x1200, sample_num 32, run_dtu.sh objective, frozen geometry)" --resources profiles/r04_kernel_resources.json $dbs < /dev/null
python - <<'PY'
import json
v = json.load(open("gpurun_out/r04_pmc_valu_dtu.json"))["kernels"]
for k in ("s2_smooth_stream_kernel", "s2_loss_kernel", "ssim_forward_kernel", "ssim_backward_kernel", "render_backward_features_kernel", "render_forward_wave_kernel"):
    b = v.get(k, {})
    print(k, {x: b.get(x) for x in ("duration_us_under_pmc", "valu_busy_frac", "valu_issue_frac", "waves_per_simd", "wait_frac", "issue_stall_frac", "clock_ghz")},
          "VALU", b.get("counters_mean_per_dispatch", {}).get("SQ_INSTS_VALU"))
PY
