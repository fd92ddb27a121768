# round 5, call F: timeline of the pipelined step after the backward rewrite, leave-room A/B, relight reference test
set -x
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8
timeout 300 python -m pytest tests/test_relight_gpu.py -q -p no:cacheprovider -s -k "reference_python" < /dev/null > gpurun_out/r05_f_relight.log 2>&1; tail -3 gpurun_out/r05_f_relight.log; grep -n "visibility classes\|^E  \|bad [1-9]" gpurun_out/r05_f_relight.log | head -20
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 3"
for v in 1 0 1 0; do R3DG_SHADE_LEAVE_ROOM=$v $B 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('leave_room=$v', d['value'], d.get('spread_iters_per_s'))"; done
cd /tmp
rm -rf /tmp/prof
R3DG_BENCH_NO_ALONE=1 timeout 300 rocprofv3 --kernel-trace -d /tmp/prof -o bench -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 < /dev/null > /root/repo/gpurun_out/r05_f_prof.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_timeline.py "$f" seq < /dev/null > gpurun_out/r05_f_sequence.txt 2>&1
python tools/rocpd_timeline.py "$f" 12 < /dev/null > gpurun_out/r05_f_timeline.txt 2>&1
cat gpurun_out/r05_f_sequence.txt | cut -c1-150
head -12 gpurun_out/r05_f_timeline.txt
