#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc rocpd databases: per kernel, the sum of every counter over its dispatches (+ dispatch count).
usage: pmc_to_csv.py <out.csv> <db> [<db> ...]"""
import sqlite3
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
names = []
for db in sys.argv[2:]:
    c = sqlite3.connect(db)
    cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
    rows = c.execute("select * from counters_collection")
    ik = cols.index("kernel_name") if "kernel_name" in cols else cols.index("name")
    ic = cols.index("counter_name")
    iv = cols.index("value")
    idd = cols.index("dispatch_id") if "dispatch_id" in cols else None
    seen = set()
    for r in rows:
        k = r[ik].split("(")[0].replace(", ", ";").replace(",", ";")
        acc[k][r[ic]] += float(r[iv])
        if r[ic] not in names:
            names.append(r[ic])
        if idd is not None and (db, r[idd]) not in seen and r[ic] == names[0]:
            seen.add((db, r[idd]))
            calls[k] += 1
with open(out, "w") as f:
    f.write("kernel,dispatches," + ",".join(names) + "\n")
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", acc[k].get(names[0], 0))):
        f.write(k + "," + str(calls.get(k, 0)) + "," + ",".join("%.0f" % acc[k].get(n, 0) for n in names) + "\n")
print(out)
