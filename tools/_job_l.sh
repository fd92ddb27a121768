timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
B="python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --relight-frames 0"
for i in 1 2; do
  $B 2>/dev/null | grep '^{"metric' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K20W5', d['value'], d['spread_iters_per_s']['min'], d['spread_iters_per_s']['median'], d['spread_iters_per_s']['max'])"
done
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof -o bench -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 > /dev/null 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd /root/repo; mkdir -p gpurun_out
python tools/rocpd_timeline.py "$f" seq > gpurun_out/seq.txt 2>&1
python tools/rocpd_timeline.py "$f" 15 > gpurun_out/seq_timeline.txt 2>&1
