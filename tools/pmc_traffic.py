"""Merge rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (rocpd sqlite databases, one counter per pass as the gpurun policy
and MI355X_MICROARCH.md prescribe) into profiles/<round>_pmc_traffic.json: mean per launch, per kernel.

    python tools/pmc_traffic.py out.json "<note>" db1 [db2 ...]

hbm_bytes_corrected = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: gfx950's FETCH_SIZE counts a 128-byte request as 64 bytes
(MI355X_MICROARCH.md, HBM section); the factor is calibrated on wide streaming reads, so for gather-heavy kernels the
truth lies between hbm_bytes_raw and hbm_bytes_corrected."""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"r3dg::(\w+)", name)
    return m.group(1) if m else name[:60]


def main():
    out, note, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        namecol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
        cntcol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
        for k, c, v in cur.execute("select %s, %s, value from counters_collection" % (namecol, cntcol)):
            if "r3dg::" not in k or c not in ("FETCH_SIZE", "WRITE_SIZE"):
                continue
            a = acc[short(k)][c]
            a[0] += float(v)
            a[1] += 1
    kernels = {}
    for k, d in sorted(acc.items()):
        f = d["FETCH_SIZE"][0] / max(d["FETCH_SIZE"][1], 1)
        w = d["WRITE_SIZE"][0] / max(d["WRITE_SIZE"][1], 1)
        kernels[k] = {"FETCH_SIZE_KB": round(f, 2), "WRITE_SIZE_KB": round(w, 2),
                      "hbm_bytes_raw": (f + w) * 1024, "hbm_bytes_corrected": (2 * f + w) * 1024}
    # which SOURCE these counters describe: sha256 of each kernel's defining file (+ its local headers) at collection time;
    # bench.py compares it with the tree it runs from and marks the figures stale when they differ
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from relightable3dgaussian_amd import kernel_sources
    json.dump({"note": note, "sources": kernel_sources.stamp(sorted(kernels)), "kernels": kernels}, open(out, "w"), indent=1)
    print("wrote", out, len(kernels), "kernels")


if __name__ == "__main__":
    main()
