set -x
cd /root/repo
python tools/kbench_shade.py 2>&1 | grep "K="
