cd /root/repo
timeout 600 python -m pytest tests/test_fused_dp_gpu.py tests/test_fused_step_gpu.py -q --maxfail=10 -p no:cacheprovider < /dev/null > gpurun_out/trial_pytest.log 2>&1
tail -15 gpurun_out/trial_pytest.log
R3DG_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 8 --warmup 4 --points 100000 < /dev/null > gpurun_out/trial_bench2.log 2>&1
tail -c 700 gpurun_out/trial_bench2.log
