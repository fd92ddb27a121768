# round 5, call B: the incident-light chain A/B (R3DG_EARLY_INCIDENTS), the fused-step / DP tests that exercise it, the re-tuned
# parity tests of call A
set -x
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
timeout 900 python -m pytest tests/test_fused_step_gpu.py tests/test_fused_dp_gpu.py tests/test_train_loop_gpu.py tests/test_psnr_vs_reference_gpu.py -q -p no:cacheprovider -x < /dev/null > gpurun_out/r05_b_fused_tests.log 2>&1; tail -5 gpurun_out/r05_b_fused_tests.log
timeout 600 python -m pytest tests/test_shading_gpu.py tests/test_relight_gpu.py tests/test_reference_pipeline_gpu.py -q -p no:cacheprovider -s -k "relight or fixed_ray_set_kernels_match_oracle or pipeline or stage" < /dev/null > gpurun_out/r05_b_parity.log 2>&1; tail -5 gpurun_out/r05_b_parity.log
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 3"
for v in 0 1 0 1; do
  R3DG_EARLY_INCIDENTS=$v $B 2>/dev/null | tail -1 > gpurun_out/r05_b_inc$v.json
  python - <<EOF
import json
d=json.load(open("gpurun_out/r05_b_inc$v.json"))
print("EARLY_INCIDENTS=$v", d["value"], d["ms_per_step"], d.get("spread_iters_per_s"))
EOF
done
R3DG_EARLY_INCIDENTS=1 R3DG_FWD_STAGGER=0 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('inc1 stagger0', d['value'], d.get('spread_iters_per_s'))"
cd /tmp
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof -o bench -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 < /dev/null > /root/repo/gpurun_out/r05_b_prof.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_timeline.py "$f" seq < /dev/null > gpurun_out/r05_b_sequence.txt 2>&1
tail -45 gpurun_out/r05_b_sequence.txt | cut -c1-150
