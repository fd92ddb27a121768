# the 2M-Gaussian configuration (configs[4]) launch by launch
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_relight_gpu.py -m gpu -x -q 2>&1 | tail -1
rm -rf /tmp/p2m; (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/p2m -o p -- python $GRAFT_REPO_ROOT/bench.py --points 2000000 --width 1800 --height 700 --steps 8 --warmup 3 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 0 > /tmp/p2m.log 2>&1)
tail -1 /tmp/p2m.log | cut -c1-200
f=$(find /tmp/p2m -name "*.db" | head -1)
python tools/rocpd_summary.py "$f" gpurun_out/r04_2m_stats.md "bench.py --points 2000000 --width 1800 --height 700 --steps 8 --warmup 3" < /dev/null > /dev/null 2>&1
python tools/rocpd_timeline.py "$f" 6 < /dev/null > gpurun_out/r04_2m_timeline.txt 2>&1
python tools/rocpd_timeline.py "$f" seq < /dev/null > gpurun_out/r04_2m_sequence.txt 2>&1
