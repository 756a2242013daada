#!/usr/bin/env python3
"""Timeline of the LAST step of a bench.py run from a rocprofv3 rocpd database (kernel trace): every dispatch in order with its
duration and the idle gap in front of it, folded into runs of the same kernel.  A step starts at k_frag_list (k_adapt_fixed before round 6).
usage: timeline.py <trace_results.db> [out.txt] [step]     step: 0-based index of the step (default: the last one of the trace)
Round 6: a step's stages may run at once on three streams (bench.py's default order): `busy` is then the union of the dispatches'
intervals, a dispatch's gap the time since the latest end of anything in front of it, and a run of equal kernels is only folded
while they follow each other in start order."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
rows = [(n.split("(")[0].replace("void ", "").replace("elp::", ""), s, e) for n, s, e in rows]
# (round 6: mark duplicates' front pass does the adapt stage's fixed-field part; a step then starts at k_frag_list)
starts = [i for i, r in enumerate(rows) if r[0].startswith("k_frag_list")] or [i for i, r in enumerate(rows) if r[0].startswith("k_adapt_fixed")]
if not starts:
    sys.exit("no k_frag_list / k_adapt_fixed dispatch in the trace")
which = int(sys.argv[3]) if len(sys.argv) > 3 else len(starts) - 1
step = rows[starts[which]:(starts[which + 1] if which + 1 < len(starts) else len(rows))]
# (bench.py restores FLAG and QUAL between two steps - two device-to-device copies of 0.1 / 7.5 GB that are not part of a step, and
# clears nothing else: the long copies at the step's end are dropped)
for k in range(len(step) - 1, max(len(step) - 8, 0), -1):  # (the next step's first fill may stand behind them)
    if step[k][0].startswith("__amd_rocclr_copyBuffer") and step[k][2] - step[k][1] > 1000000:
        step = step[:k - 1] if k >= 1 and step[k - 1][0].startswith("__amd_rocclr_copyBuffer") else step[:k]
        break
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
# (union of the intervals: dispatches of different streams overlap)
iv = sorted((s, e) for _, s, e in step)
union, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s_, e_ in iv[1:]:
    if s_ > cur_e:
        union += cur_e - cur_s
        cur_s, cur_e = s_, e_
    else:
        cur_e = max(cur_e, e_)
union = (union + cur_e - cur_s) / 1e3
out.write(f"# step {which} of {len(starts)}: {len(step)} dispatches, span {span:.0f} us, sum of kernel times {busy:.0f} us, some kernel running {union:.0f} us, idle {span - union:.0f} us\n")
out.write("start_us,kernel,calls,busy_us,gap_us\n")
for n, k, d, g, s in folded:
    out.write(f"{s:.0f},{n},{k},{d:.1f},{g:.1f}\n")
