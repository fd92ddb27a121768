cd /root/repo
timeout 300 python -m pytest tests/test_relight_gpu.py -q -p no:cacheprovider < /dev/null > gpurun_out/trial_pytest.log 2>&1
tail -25 gpurun_out/trial_pytest.log
timeout 150 python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-other-configs --relight-frames 12 < /dev/null > gpurun_out/b4.log 2>&1
python - <<EOF
import json
l=[x for x in open("gpurun_out/b4.log") if x.startswith("{")]
print(json.loads(l[-1])["relight"] if l else open("gpurun_out/b4.log").read()[-1500:])
EOF
