cd /root/repo
timeout 600 python -m pytest tests/test_fused_step_gpu.py -q -x -p no:cacheprovider < /dev/null > gpurun_out/trial_pytest.log 2>&1
tail -3 gpurun_out/trial_pytest.log
python bench.py --no-cpu-baseline --relight-frames 0 --no-other-configs > gpurun_out/b3.log 2>&1; python - <<EOF
import json
l=[x for x in open("gpurun_out/b3.log") if x.startswith("{")][-1]
d=json.loads(l); print(d["value"], d["ms_per_step"]); print({k:v["ms_per_iteration"] for k,v in d["kernels"].items()})
EOF
