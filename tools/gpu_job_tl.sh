cd /tmp; export TMPDIR=/tmp
CMD="python /root/repo/bench.py --steps 30 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0"
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o bench -- $CMD < /dev/null > /root/repo/gpurun_out/prof_bench.log 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd /root/repo
python tools/rocpd_timeline.py "$f" 25 < /dev/null > gpurun_out/timeline.txt 2>&1
head -16 gpurun_out/timeline.txt
python tools/rocpd_timeline.py "$f" seq > gpurun_out/timeline_seq.txt 2>&1
