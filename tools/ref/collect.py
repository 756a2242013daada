#!/usr/bin/env python3
"""What the real `elprep filter` wrote, as a fixture the oracle is compared with (tools/ref/make_fixtures.sh).
usage: collect.py <workdir with in.sam out.sam metrics.txt recal.txt case.json> <fixture.json>
Records are identified by (QNAME, first/last/secondary/supplementary bits, RNAME, POS): the input's staging index of every output line."""
import hashlib
import json
import os
import sys


def key(f):
    return (f[0], int(f[1]) & 0x9C0, f[2], f[3], f[5])  # QNAME, FIRST|LAST|SECONDARY|SUPPLEMENTARY, RNAME, POS, CIGAR


def records(path):
    for line in open(path):
        if not line.startswith("@"):
            yield line.rstrip("\n").split("\t")


def main():
    w, out = sys.argv[1], sys.argv[2]
    index = {}
    for i, f in enumerate(records(os.path.join(w, "in.sam"))):
        k = key(f)
        if k in index:
            sys.exit(f"collect: input records {index[k]} and {i} share the identifying fields {k}")
        index[k] = i
    n = len(index)
    order, flags, quals = [], [0] * n, [""] * n
    for f in records(os.path.join(w, "out.sam")):
        i = index[key(f)]
        order.append(i)
        flags[i] = int(f[1])
        quals[i] = f[10]
    fix = json.load(open(os.path.join(w, "case.json")))
    fix.update({
        "order": order,                      # output line -> staging index (records dropped by the pipeline are absent)
        "flags": flags,                      # by staging index (0 for dropped records)
        "qual_sha256": hashlib.sha256("\n".join(quals).encode()).hexdigest(),
        "qual_head": quals[:8],
        "metrics_txt": open(os.path.join(w, "metrics.txt")).read(),
        "recal_txt": open(os.path.join(w, "recal.txt")).read(),
    })
    json.dump(fix, open(out, "w"))
    print(f"collect: {len(order)} of {n} records in the output, {sum(1 for x in flags if x & 0x400)} duplicates -> {out}")


if __name__ == "__main__":
    main()
