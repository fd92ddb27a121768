cd /root/repo
timeout 600 python -m pytest tests/test_bvh_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5
timeout 600 python tools/kbench_trace.py 2>&1 | tail -8
