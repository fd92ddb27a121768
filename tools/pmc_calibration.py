"""Merge the FETCH_SIZE and WRITE_SIZE passes of tools/pmc_calibration (rocpd sqlite) into profiles/<round>_pmc_calibration.json:
per access pattern the bytes the kernel provably moves, what the counters report, and the factor between them.

    python tools/pmc_calibration.py out.json fetch.db write.db"""
import json
import re
import sqlite3
import sys
from collections import defaultdict

GIB = 1 << 30
M = 16 << 20
# kernel -> (bytes read, bytes written) requested by the program; for the gathers the bytes in distinct 64 B / 128 B lines
KNOWN = {
    "stream_read16": (GIB, 0), "stream_read4": (GIB, 0), "stream_copy16": (GIB, GIB), "stream_write4": (0, GIB),
    "stream_write16": (0, GIB), "gather_records<4>": (4 * M, 0), "gather_records<8>": (8 * M, 0),
    "gather_records<16>": (16 * M, 0), "gather_records<64>": (64 * M, 0), "scatter_atomic4": (4 * M, 4 * M),
}


def main():
    out, dbs = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
        namecol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
        cntcol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
        for k, c, v in cur.execute("select %s, %s, value from counters_collection" % (namecol, cntcol)):
            m = re.search(r"calib::(\w+(?:<\d+>)?)", k)
            if not m:
                continue
            a = acc[m.group(1)][c]
            a[0] += float(v)
            a[1] += 1
    rows = {}
    for k, (rd, wr) in KNOWN.items():
        d = acc.get(k, {})
        f = d["FETCH_SIZE"][0] / d["FETCH_SIZE"][1] * 1024 if d.get("FETCH_SIZE", [0, 0])[1] else None
        w = d["WRITE_SIZE"][0] / d["WRITE_SIZE"][1] * 1024 if d.get("WRITE_SIZE", [0, 0])[1] else None
        row = {"bytes_read_requested": rd, "bytes_written_requested": wr, "FETCH_SIZE_bytes": f, "WRITE_SIZE_bytes": w}
        if f is not None and rd:
            row["read_bytes_per_FETCH_SIZE_byte"] = round(rd / f, 3)
            if k.startswith("gather"):
                rec = int(re.search(r"<(\d+)>", k).group(1))
                row["line64_bytes_touched"] = M * 64
                row["line64_bytes_per_FETCH_SIZE_byte"] = round(M * 64 / f, 3)
                row["line128_bytes_touched"] = M * 128
                row["record_bytes"] = rec
        if w is not None and wr:
            row["written_bytes_per_WRITE_SIZE_byte"] = round(wr / w, 3)
        rows[k] = row
    doc = {"note": "tools/pmc_calibration on MI355X: 1 GiB arrays (4x the Infinity Cache), 16 Mi random records per gather, "
                   "FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes, mean per launch (KB -> bytes x1024). "
                   "read_bytes_per_FETCH_SIZE_byte is the factor to multiply FETCH_SIZE with for that access pattern.",
           "patterns": rows}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(rows, indent=1))


if __name__ == "__main__":
    main()
