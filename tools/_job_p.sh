cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof -o bench -- python /root/repo/bench.py --steps 20 --warmup 5 --no-cpu-baseline --relight-frames 0 --no-other-configs --repeats 0 > /dev/null 2>&1
f=$(find /tmp/prof -name "*.db" | head -1)
cd /root/repo; mkdir -p gpurun_out
python tools/rocpd_timeline.py "$f" seq > gpurun_out/seq.txt 2>&1
python tools/rocpd_timeline.py "$f" 15 > gpurun_out/seq_timeline.txt 2>&1
python tools/host_time_probe.py 2>&1 | tail -5
