# final check of a round: smoke(), the whole GPU suite, the default bench line
cd /root/repo; mkdir -p gpurun_out
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider < /dev/null > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
timeout 600 python bench.py < /dev/null > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log > gpurun_out/bench_default.json
python -c "
import json
d=json.load(open('gpurun_out/bench_default.json'))
print(d['value'], d['ms_per_step'], d['spread_iters_per_s'])
print({k:v for k,v in d['relight'].items() if k!='kernels'})
print(d['other_configs'])"
