cd /root/repo
timeout 900 python -m pytest tests/test_bvh_gpu.py tests/test_reference_gpu.py -m gpu -x -q -p no:cacheprovider -k "bvh or BVH or trace" 2>&1 | tail -4
MODES=4 python tools/kbench_trace.py 2>&1 | grep packet
