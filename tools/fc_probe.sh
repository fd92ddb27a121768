#!/bin/bash
# one-rank data-parallel bench lines under a priced all-reduce: tools/fc_probe.sh "<GB/s list>" [extra env assignments]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for g in $1; do
  env R3DG_DP_SINGLE_RANK=1 R3DG_DIST_BACKEND=nccl ${g:+R3DG_DP_FAKE_COMM_GBS=$g} $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --relight-frames 0 --repeats 0 2>/dev/null | grep '^{' | tail -1 > /tmp/fc.json
  python - "$g" <<'PY'
import json, sys
d = json.loads(open("/tmp/fc.json").read())
k = d["kernels"]
print(sys.argv[1], "GB/s:", d["value"], "it/s", d["ms_per_step"], "ms; exposed", d.get("exposed_comm_ms"), "reserved", d.get("reserved_cus_for_comm"),
      {n: k[n]["avg_ms"] for n in ("render_backward", "shade_backward", "shade_forward", "adam_step") if n in k})
PY
done
