cd /root/repo
timeout 900 python -m pytest tests/test_fused_dp_gpu.py tests/test_fused_step_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -8
