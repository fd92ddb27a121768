#!/bin/bash
# A/B of the stage-2 iteration at sample_num 384 under environment settings
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for spec in "$@"; do
  label="${spec%%=*}"; envs="${spec#*=}"
  env $envs python bench.py --sample-num 384 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 1 2>/dev/null | grep '^{' | tail -1 > /tmp/ab.json
  python - "$label" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read())
k = d["kernels"]
print("%-14s %7.1f it/s  " % (sys.argv[1], d["value"]),
      {n: k[n]["avg_ms"] for n in ("shade_forward", "sort_pairs", "duplicate_with_keys", "render_forward", "shade_backward", "render_backward") if n in k})
PY
done
