# full GPU suite + default bench (name prefix = $1)
set -x
cd /root/repo; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider < /dev/null > gpurun_out/$1_pytest_gpu.log 2>&1; tail -5 gpurun_out/$1_pytest_gpu.log
timeout 600 python bench.py < /dev/null > gpurun_out/$1_bench.log 2>&1; tail -1 gpurun_out/$1_bench.log > gpurun_out/$1_bench.json; cut -c1-400 gpurun_out/$1_bench.json
