cd /root/repo
timeout 600 python -m pytest tests/test_bvh_gpu.py -m gpu -x -q -p no:cacheprovider -k "formulations" 2>&1 | tail -3
MODES=4,3,0 timeout 600 python tools/kbench_trace.py 2>&1 | grep "packet\|bitwise\|mismatch"
