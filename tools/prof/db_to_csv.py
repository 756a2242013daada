#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (trace_results.db) into the per-kernel summary CSV committed under profiles/.
usage: db_to_csv.py <trace_results.db> <out.csv> [header comment] [skip_steps] [n_steps]
skip_steps > 0: leave out the dispatches of the first `skip_steps` steps of a bench.py run (a step starts at k_frag_list; k_adapt_fixed before round 6) - the
warm-up steps, whose first launches run cold (the first k_bqsr_count of a process takes 3x its steady time) and are not part of
what bench.py times.  With skip_steps = 0 the numbers are rocprofv3's own `top_kernels` summary."""
import sqlite3
import sys
from collections import defaultdict

db, out = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
n_steps = int(sys.argv[5]) if len(sys.argv) > 5 else 0  # > 0: only the dispatches of that many steps from `skip` on (bench.py's timed steps: what
                                                        # follows them - the serial-order steps of round 6 - is not part of them)
c = sqlite3.connect(db)
if skip == 0:
    rows = [(n.split("(")[0], calls, tot, avg, pct) for n, calls, tot, avg, pct in
            c.execute("select name, total_calls, total_duration, average, percentage from top_kernels")]
else:
    ks = list(c.execute("select name, start, end from kernels order by start"))
    starts = [s for n, s, e in ks if "k_frag_list" in n] or [s for n, s, e in ks if "k_adapt_fixed" in n]
    if len(starts) <= skip:
        sys.exit(f"only {len(starts)} steps in the trace")
    t0 = starts[skip]
    t1 = starts[skip + n_steps] if n_steps and len(starts) > skip + n_steps else float("inf")
    acc = defaultdict(lambda: [0, 0.0])
    for n, s, e in ks:
        if t0 <= s < t1:
            a = acc[n.split("(")[0]]
            a[0] += 1
            a[1] += (e - s) / 1e3
    total = sum(v[1] for v in acc.values())
    rows = sorted(((n, v[0], v[1], v[1] / v[0], 100.0 * v[1] / total) for n, v in acc.items()), key=lambda r: -r[2])
with open(out, "w") as f:
    if note:
        f.write("# " + note + "\n")
    f.write("kernel,calls,total_us,avg_us,pct\n")
    for name, calls, tot, avg, pct in rows:
        f.write(f"{name},{calls},{tot:.1f},{avg:.1f},{pct:.2f}\n")
print(f"{len(rows)} kernels -> {out}")
