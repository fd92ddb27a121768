#!/bin/bash
# Round 4, GPU job B: SQ counters of the relight kernels (fixed light: shade_forward_transport_kernel; turning light:
# shade_forward_split_kernel) from the relight part of bench.py.
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
  rm -rf /tmp/pr_$i
  timeout 280 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pr_$i -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --repeats 0 --relight-frames 8 < /dev/null > /tmp/pr.log 2>&1
  dbs="$dbs $(find /tmp/pr_$i -name '*.db' | head -1)"
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pt_$c
  timeout 280 rocprofv3 --pmc $c --kernel-trace -d /tmp/pt_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --repeats 0 --relight-frames 8 < /dev/null > /tmp/pr.log 2>&1
  tdbs="$tdbs $(find /tmp/pt_$c -name '*.db' | head -1)"
done
cd $R
python tools/pmc_valu.py gpurun_out/r04_pmc_valu_relight.json "rocprofv3 --pmc <one SQ group per pass> --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs --repeats 0 --relight-frames 8 (K=384, 300k Gaussians, 256x512 HDR map)" $dbs < /dev/null
python tools/pmc_traffic.py gpurun_out/r04_pmc_traffic_relight.json "the same command, FETCH_SIZE / WRITE_SIZE in separate passes" $tdbs < /dev/null
python - <<'PY'
import json
v = json.load(open("gpurun_out/r04_pmc_valu_relight.json"))["kernels"]
t = json.load(open("gpurun_out/r04_pmc_traffic_relight.json"))["kernels"]
for k in ("shade_forward_split_kernel", "shade_forward_transport_kernel", "shade_forward_row_kernel", "shade_build_split_kernel"):
    b = v.get(k, {})
    a = t.get(k, {})
    print(k, {x: b.get(x) for x in ("duration_us_under_pmc", "valu_busy_frac", "waves_per_simd", "wait_frac", "issue_stall_frac", "clock_ghz", "trans_frac", "salu_issue_frac")},
          "VALU", b.get("counters_mean_per_dispatch", {}).get("SQ_INSTS_VALU"), "VMEM", b.get("counters_mean_per_dispatch", {}).get("SQ_INSTS_VMEM"),
          "raw MB %.0f corrected %.0f" % (a.get("hbm_bytes_raw", 0) / 1e6, a.get("hbm_bytes_corrected", 0) / 1e6))
PY
