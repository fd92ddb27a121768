cd /root/repo
timeout 900 python -m pytest tests/test_rasterizer_gpu.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
ITERS=10 python tools/kbench_raster.py 2>&1 | tail -1
S=28 ITERS=6 python tools/kbench_raster.py 2>&1 | tail -1
