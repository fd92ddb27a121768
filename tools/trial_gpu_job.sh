cd /root/repo
timeout 900 python -m pytest tests/test_densify_gpu.py tests/test_train_loop_gpu.py "tests/test_fused_step_gpu.py::test_fused_stage1_training_with_densification" "tests/test_fused_step_gpu.py::test_fused_stage1_matches_autograd" tests/test_fused_dp_gpu.py -q --maxfail=10 -p no:cacheprovider < /dev/null > gpurun_out/trial_pytest.log 2>&1
tail -40 gpurun_out/trial_pytest.log
timeout 400 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --relight-frames 0 < /dev/null > gpurun_out/trial_bench.log 2>&1
tail -c 1500 gpurun_out/trial_bench.log
