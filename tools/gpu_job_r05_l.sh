#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
for i in 1 2; do
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 3 < /dev/null 2> gpurun_out/l_bench$i.err | tail -1 | cut -c1-200
python - <<'P'
import json
d=json.load(open("gpurun_out/bench_full.json"))
print(d["value"], d["spread_iters_per_s"]["median"], d["device_clock"]["shader_clock_ghz_under_valu_load"], {k:v["ms_per_iteration"] for k,v in d["kernels"].items() if k in ("ssim","render_backward","preprocess_backward","shade_frs_listed","stage2_loss","pseudo_normal","stage2_activate","stage2_activate_backward")})
P
done
