#!/bin/bash
# A/B of the headline iteration under environment settings: tools/ab_bench.sh "<label>=<env assignments>" ...   (GPU box)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for spec in "$@"; do
  label="${spec%%=*}"; envs="${spec#*=}"
  for rep in 1 2; do
    env $envs python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 2 2>/dev/null | grep '^{' | tail -1 > /tmp/ab.json
    python - "$label" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab.json").read())
k = d["kernels"]
print("%-14s %7.1f it/s  median %s  " % (sys.argv[1], d["value"], d.get("spread_iters_per_s", {}).get("median")),
      {n: k[n]["avg_ms"] for n in ("shade_forward", "sort_pairs", "duplicate_with_keys", "render_forward", "shade_backward", "render_backward") if n in k})
PY
  done
done
