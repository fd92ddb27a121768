cd /root/repo
run() { # name, args...
  name=$1; shift
  timeout 120 python -c "
import faulthandler, sys, runpy
faulthandler.dump_traceback_later(75, exit=True)
sys.argv = ['bench.py'] + '$*'.split()
runpy.run_path('bench.py', run_name='__main__')
" > gpurun_out/diag_$name.log 2>&1
  echo "== $name rc=$?"; tail -c 1200 gpurun_out/diag_$name.log | tail -25
}
run relight --steps 8 --warmup 4 --no-other-configs --no-cpu-baseline --relight-frames 5
run cpu --steps 8 --warmup 4 --no-other-configs --relight-frames 0 --cpu-baseline-seconds 5
