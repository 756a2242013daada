#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (trace_results.db) into the per-kernel summary CSV committed under profiles/.
usage: db_to_csv.py <trace_results.db> <out.csv> [header comment]"""
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
c = sqlite3.connect(db)
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
with open(out, "w") as f:
    if note:
        f.write("# " + note + "\n")
    f.write("kernel,calls,total_us,avg_us,pct\n")
    for name, calls, tot, avg, pct in rows:
        name = name.split("(")[0]
        f.write(f"{name},{calls},{tot:.1f},{avg:.1f},{pct:.2f}\n")
    try:
        pm = list(c.execute("select * from counters_collection limit 1"))
    except Exception:
        pm = []
print(f"{len(rows)} kernels -> {out}")
