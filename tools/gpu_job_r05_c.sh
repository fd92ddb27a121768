# round 5, call C: the rewritten tile backward (moments, one LDS record, lean instances): rasterizer / reference / fused tests, A/B
# against the library built before the rewrite (lib/variants/r05_base), the re-tuned parity tests, a timeline of the pipelined step
set -x
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
timeout 900 python -m pytest tests/test_rasterizer_gpu.py -q -p no:cacheprovider -x -k "backward or gradient or lean" < /dev/null > gpurun_out/r05_c_raster.log 2>&1; tail -4 gpurun_out/r05_c_raster.log
timeout 900 python -m pytest tests/test_reference_gpu.py tests/test_fused_step_gpu.py tests/test_reference_pipeline_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/r05_c_ref.log 2>&1; tail -4 gpurun_out/r05_c_ref.log
timeout 600 python -m pytest tests/test_shading_gpu.py tests/test_relight_gpu.py -q -p no:cacheprovider -s -k "relight or fixed_ray_set_kernels_match_oracle" < /dev/null > gpurun_out/r05_c_parity.log 2>&1; tail -4 gpurun_out/r05_c_parity.log
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 3"
for lib in base new base new; do
  if [ $lib = base ]; then export R3DG_LIB_PATH=/root/repo/relightable3dgaussian_amd/lib/variants/r05_base/libr3dg_hip.so; else unset R3DG_LIB_PATH; fi
  $B 2>/dev/null | tail -1 > /tmp/ab.json
  python - <<EOF
import json
d=json.load(open("/tmp/ab.json")); f=json.load(open("gpurun_out/bench_full.json")); k=f["kernels"]
print("lib=$lib", d["value"], d.get("spread_iters_per_s"), {n: (k[n]["ms_per_iteration"], k[n].get("alone_ms_per_iteration")) for n in ("render_backward","preprocess_backward","render_forward")})
EOF
done
unset R3DG_LIB_PATH
R3DG_OPT_BWD_LEAN=0 $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new, lean off', d['value'], d.get('spread_iters_per_s'), d['roofline']['avg_kernel_ms'])"
cd /tmp
rm -rf /tmp/prof
R3DG_BENCH_NO_ALONE=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof -o bench -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 < /dev/null > /root/repo/gpurun_out/r05_c_prof.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_timeline.py "$f" seq < /dev/null > gpurun_out/r05_c_sequence.txt 2>&1
python tools/rocpd_timeline.py "$f" 12 < /dev/null > gpurun_out/r05_c_timeline.txt 2>&1
cat gpurun_out/r05_c_sequence.txt | cut -c1-150
head -12 gpurun_out/r05_c_timeline.txt
