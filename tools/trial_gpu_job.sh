cd /root/repo
STAGE=0 python tools/kbench_raster.py 2>&1 | tail -1
STAGE=1 python tools/kbench_raster.py 2>&1 | tail -1
timeout 900 python -m pytest tests/test_rasterizer_gpu.py tests/test_reference_gpu.py -q -x -p no:cacheprovider < /dev/null > gpurun_out/trial_pytest.log 2>&1
tail -5 gpurun_out/trial_pytest.log
