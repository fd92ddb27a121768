#!/bin/bash
# ONE parametrised GPU job (replaces the per-call tools/gpu_job_rNN_*.sh scripts of rounds 2-5).  Runs on the MI355X box:
#     gpurun --timeout 900 -- 'bash tools/gpu_job.sh <task> [args] [+ <task> [args]] ...'
# Every task writes under gpurun_out/ (merged back by gpurun); the summaries to be judged are copied to profiles/ by hand.
# ROUND=r06 names the files.  Tasks ("+" chains several in one call):
#   smoke                         __graft_entry__.smoke()
#   suite                         smoke + the whole GPU suite                       -> <RD>_smoke.txt, <RD>_pytest_gpu.txt
#   tests <pytest args>           python -m pytest <args> -m gpu                    -> <RD>_tests_<n>.txt
#   bench [bench.py args]         the bench line (default arguments when none)      -> <RD>_bench_<n>.json (+ bench_full.json)
#   short [bench.py args]         the headline iteration only (no side measurements, 40 steps, 2 extra blocks)
#   ab <steps> "<label>=<ENV=val ...>" ...   A/B of the headline iteration under environment settings, two runs each
#   prof <tag> [bench.py args]    rocprofv3 --kernel-trace --stats of a bench command: kernel stats, per-step timeline, one step
#                                 launch by launch                                   -> <RD>_<tag>_kernel_stats.md, _timeline.txt, _sequence.txt
#   pmc <tag> <counters|@A|@B|@C|@L2|@HBM> -- <command>    one rocprofv3 --pmc pass per counter GROUP (--kernel-trace only), then
#                                 tools/pmc_valu.py / pmc_traffic.py                 -> <RD>_pmc_<tag>.json
#   evidence [quick]              the round-end evidence run (tools/round_end_gpu_job.sh)
#   run <shell command>           anything else, output to <RD>_run_<n>.txt
RD=${ROUND:-r06}
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=8 MKL_NUM_THREADS=8   # (threads: the container's CPU quota, see bench.py)
R=$PWD
O=$R/gpurun_out
n=0
GA="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
GB="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
GC="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"
GL2="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE"
SHORT="--no-cpu-baseline --no-other-configs --relight-frames 0"

show_bench() {      # one line per bench document: value, median, the hot stages
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
except Exception as e:
    print("%-28s no bench line (%r)" % (sys.argv[1], e)); sys.exit(0)
k = d.get("kernels") or {}
row = {n: k[n].get("ms_per_iteration") for n in ("shade_forward", "duplicate_with_keys", "sort_pairs", "render_forward", "render_backward",
                                                  "shade_backward", "shade_frs_aux", "adam_step") if n in k}
print("%-28s %8.1f it/s  median %s  relight %s  %s" % (sys.argv[1], d.get("value") or 0, (d.get("spread_iters_per_s") or {}).get("median"),
                                                       d.get("relight_fps") or (d.get("relight") or {}).get("relight_fps"), row))
PY
}

run_task() {
  task=$1; shift
  n=$((n+1))
  case $task in
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $O/${RD}_smoke.txt 2>&1; tail -1 $O/${RD}_smoke.txt ;;
    suite)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $O/${RD}_smoke.txt 2>&1; tail -1 $O/${RD}_smoke.txt
      timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 < /dev/null > $O/${RD}_pytest_gpu.txt 2>&1; tail -25 $O/${RD}_pytest_gpu.txt ;;
    tests)
      timeout 2400 python -m pytest "$@" -m gpu -q -p no:cacheprovider < /dev/null > $O/${RD}_tests_$n.txt 2>&1; tail -30 $O/${RD}_tests_$n.txt ;;
    bench)
      timeout 1200 python bench.py "$@" < /dev/null > $O/${RD}_bench_$n.out 2> $O/${RD}_bench_$n.err
      tail -1 $O/${RD}_bench_$n.out > $O/${RD}_bench_${n}_compact.json; cp $O/bench_full.json $O/${RD}_bench_$n.json 2>/dev/null
      cut -c1-1500 $O/${RD}_bench_${n}_compact.json; tail -3 $O/${RD}_bench_$n.err | cut -c1-300 ;;
    short)
      timeout 600 python bench.py --steps 40 --warmup 8 --repeats 2 $SHORT "$@" < /dev/null 2> $O/${RD}_short_$n.err | tail -1 > /dev/null
      cp $O/bench_full.json $O/${RD}_short_$n.json; show_bench "short $*" $O/${RD}_short_$n.json ;;
    ab)
      steps=$1; shift
      for spec in "$@"; do
        label="${spec%%=*}"; envs="${spec#*=}"
        for rep in 1 2; do
          env $envs timeout 600 python bench.py --steps $steps --warmup 8 --repeats 2 $SHORT $AB_ARGS < /dev/null > /dev/null 2> $O/${RD}_ab.err
          show_bench "$label" $O/bench_full.json | tee -a $O/${RD}_ab_$n.txt
        done
      done ;;
    prof)
      tag=$1; shift
      args="$*"; [ -z "$args" ] && args="--steps 20 --warmup 5 --repeats 0 $SHORT"
      rm -rf /tmp/prof_$n
      (cd /tmp && R3DG_BENCH_NO_ALONE=1 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$n -o bench -- python $R/bench.py $args < /dev/null > $O/${RD}_${tag}_prof.log 2>&1)
      f=$(find /tmp/prof_$n -name "*.db" | head -1)
      python tools/rocpd_summary.py "$f" $O/${RD}_${tag}_kernel_stats.md "R3DG_BENCH_NO_ALONE=1 rocprofv3 --kernel-trace --stats -- python bench.py $args" < /dev/null
      python tools/rocpd_timeline.py "$f" 15 < /dev/null > $O/${RD}_${tag}_timeline.txt 2>&1
      python tools/rocpd_timeline.py "$f" seq < /dev/null > $O/${RD}_${tag}_sequence.txt 2>&1
      head -40 $O/${RD}_${tag}_kernel_stats.md | cut -c1-200 ;;
    pmc)
      tag=$1; shift
      groups=()
      while [ $# -gt 0 ] && [ "$1" != "--" ]; do
        case $1 in @A) groups+=("$GA");; @B) groups+=("$GB");; @C) groups+=("$GC");; @L2) groups+=("$GL2");;
                   @HBM) groups+=("FETCH_SIZE" "WRITE_SIZE");; *) groups+=("$1");; esac
        shift
      done
      shift
      dbs=""; i=0; traffic=0
      for grp in "${groups[@]}"; do
        i=$((i+1)); rm -rf /tmp/pmc_${n}_$i
        case "$grp" in FETCH_SIZE|WRITE_SIZE) traffic=1;; esac
        (cd /tmp && timeout 400 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_${n}_$i -o p -- "$@" < /dev/null > /tmp/pmc_${n}_$i.log 2>&1)
        dbs="$dbs $(find /tmp/pmc_${n}_$i -name '*.db' | head -1)"
      done
      if [ $traffic = 1 ]; then
        python tools/pmc_traffic.py $O/${RD}_pmc_${tag}.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes -- $*" $dbs < /dev/null
      else
        python tools/kernel_resources.py $O/${RD}_kernel_resources.json < /dev/null > /dev/null 2>&1
        python tools/pmc_valu.py $O/${RD}_pmc_${tag}.json "rocprofv3 --pmc <one counter group per pass> --kernel-trace -- $*" --resources $O/${RD}_kernel_resources.json $dbs < /dev/null
      fi ;;
    evidence)
      ROUND=$RD bash tools/round_end_gpu_job.sh "$@" ;;
    run)
      timeout 2400 bash -c "$*" < /dev/null > $O/${RD}_run_$n.txt 2>&1; tail -40 $O/${RD}_run_$n.txt | cut -c1-400 ;;
    *)
      echo "gpu_job.sh: unknown task '$task'"; return 1 ;;
  esac
}

args=()
for a in "$@"; do
  if [ "$a" = "+" ]; then
    [ ${#args[@]} -gt 0 ] && run_task "${args[@]}"
    args=()
  else
    args+=("$a")
  fi
done
[ ${#args[@]} -gt 0 ] && run_task "${args[@]}"
exit 0
