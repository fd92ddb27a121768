#!/bin/bash
# loss kernel loads-first (bench + tests) and the host profile of the unmodified train.py at the headline size
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8
timeout 600 python -m pytest tests/test_fused_step_gpu.py tests/test_reference_pipeline_gpu.py -q -x -p no:cacheprovider < /dev/null > gpurun_out/m_pytest.txt 2>&1; tail -3 gpurun_out/m_pytest.txt
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 3 < /dev/null 2> gpurun_out/m_bench.err | tail -1 | cut -c1-200
python - <<'P'
import json
d=json.load(open("gpurun_out/bench_full.json"))
print(d["value"], d["spread_iters_per_s"]["median"], d["device_clock"]["shader_clock_ghz_under_valu_load"], {k:v["ms_per_iteration"] for k,v in d["kernels"].items() if k in ("ssim","render_backward","stage2_loss","pseudo_normal","stage2_activate","stage2_activate_backward")})
P
timeout 900 python tools/reference_train_py_gpu_run.py --reference reference_scratch --headline --views 16 --res 800 --sample-num 64 --stage2-iterations 200 --profile > gpurun_out/r05_reference_train_py_headline.txt 2>&1
tail -60 gpurun_out/r05_reference_train_py_headline.txt | cut -c1-220
