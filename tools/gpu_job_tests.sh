# usage: bash tools/gpu_job_tests.sh <log name> <pytest args...>
set -x
cd /root/repo
mkdir -p gpurun_out
name=$1; shift
timeout 1200 python -m pytest "$@" -q -s -p no:cacheprovider < /dev/null > gpurun_out/$name.log 2>&1
grep -v "^W2026\|amdgpu.ids\|socket.cpp\|Gloo" gpurun_out/$name.log | tail -60
