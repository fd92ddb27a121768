"""Summarise rocprofv3 --pmc output (rocpd sqlite): per kernel, per counter: mean value over dispatches.
    python tools/rocpd_counters.py <db> [substring filter]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = db.cursor()
views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print("# columns:", cols)
namecol = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
cntcol = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
valcol = "value" if "value" in cols else [c for c in cols if "value" in c][0]
acc = defaultdict(lambda: [0.0, 0])
for k, c, v in cur.execute("select %s, %s, %s from counters_collection" % (namecol, cntcol, valcol)):
    if flt and flt not in k:
        continue
    a = acc[(k[:70], c)]
    a[0] += float(v)
    a[1] += 1
for (k, c), (tot, n) in sorted(acc.items()):
    print("%-72s %-28s mean %.4g  (n=%d)" % (k, c, tot / n, n))
