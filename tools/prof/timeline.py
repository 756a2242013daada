#!/usr/bin/env python3
"""Timeline of the LAST step of a bench.py run from a rocprofv3 rocpd database (kernel trace): every dispatch in order with its
duration and the idle gap in front of it, folded into runs of the same kernel.  A step starts at k_frag_list (k_adapt_fixed before round 6).
usage: timeline.py <trace_results.db> [out.txt]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
rows = [(n.split("(")[0].replace("void ", "").replace("elp::", ""), s, e) for n, s, e in rows]
# (round 6: mark duplicates' front pass does the adapt stage's fixed-field part; a step then starts at k_frag_list)
starts = [i for i, r in enumerate(rows) if r[0].startswith("k_frag_list")] or [i for i, r in enumerate(rows) if r[0].startswith("k_adapt_fixed")]
if not starts:
    sys.exit("no k_frag_list / k_adapt_fixed dispatch in the trace")
step = rows[starts[-1]:]
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
t0 = step[0][1]
busy = 0
folded = []  # name, calls, busy_us, gap_us, first_start_us
prev_end = t0
for n, s, e in step:
    gap = max(0, s - prev_end) / 1e3
    d = (e - s) / 1e3
    busy += d
    if folded and folded[-1][0] == n:
        folded[-1][1] += 1; folded[-1][2] += d; folded[-1][3] += gap
    else:
        folded.append([n, 1, d, gap, (s - t0) / 1e3])
    prev_end = max(prev_end, e)
span = (prev_end - t0) / 1e3
out.write(f"# last step: {len(step)} dispatches, span {span:.0f} us, kernels busy {busy:.0f} us, idle {span - busy:.0f} us\n")
out.write("start_us,kernel,calls,busy_us,gap_us\n")
for n, k, d, g, s in folded:
    out.write(f"{s:.0f},{n},{k},{d:.1f},{g:.1f}\n")
