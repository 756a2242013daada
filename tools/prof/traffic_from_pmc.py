#!/usr/bin/env python3
"""profiles/traffic.json from a PMC summary (tools/prof/pmc_to_csv.py): HBM bytes per read and kernel label.
bytes = 2 * FETCH_SIZE + WRITE_SIZE in KB (MI355X_MICROARCH.md: FETCH_SIZE tallies 64 B per 128-B request on gfx950), summed over the
kernel's dispatches of one pass of the path, divided by the reads of that pass.  Kernel functions are mapped to the labels bench.py
prints (the first argument of ELP_LAUNCH in elprep_amd/csrc/*.hip).
usage: traffic_from_pmc.py <pmc.csv> <reads> <out.json> [source note]"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
label_of = {}
for f in glob.glob(os.path.join(ROOT, "elprep_amd", "csrc", "*.hip")):
    for m in re.finditer(r'ELP_LAUNCH\(c,\s*"([a-z0-9_]+)",\s*\(?([A-Za-z_0-9]+)', open(f).read()):
        label_of.setdefault(m.group(2), m.group(1))
src, reads, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
acc = {}
for row in csv.DictReader(open(src)):
    fn = re.sub(r"<.*", "", row["kernel"].replace("void ", "").replace("elp::", "")).strip()
    lab = label_of.get(fn)
    if lab is None:
        continue
    b = (2.0 * float(row["FETCH_SIZE"]) + float(row["WRITE_SIZE"])) * 1024.0
    acc[lab] = acc.get(lab, 0.0) + b
tj = {"source": (sys.argv[4] if len(sys.argv) > 4 else src) + ": rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
      "tools/prof/run_path.py; FETCH_SIZE doubled (MI355X_MICROARCH.md: it tallies 64 B per 128-B request on gfx950); summed over the kernel's "
      "dispatches of one pass of the path",
      "reads": reads,
      "bytes_per_read": {k: round(v / reads, 2) for k, v in sorted(acc.items(), key=lambda kv: -kv[1]) if v / reads >= 1.0}}
json.dump(tj, open(out, "w"), indent=1)
print(json.dumps(tj["bytes_per_read"]))
