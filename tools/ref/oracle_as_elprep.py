#!/usr/bin/env python3
"""A STAND-IN for the elprep binary in a dry run of tools/ref/make_fixtures.sh: writes out.sam (and metrics.txt / recal.txt where the
case asks for them) for a work directory of write_inputs.py - from the CPU oracle, in the formats elprep writes them - so that the
recipe's plumbing (SAM writer, record identification in collect.py, the comparison of tests/test_oracle_golden.py) can be exercised where
no Go toolchain exists: `filter`, `sfm` (split / per-split filter / merge), the hand-derived edge cases and CleanSam (round 6).  What it
produces is the oracle compared with itself: it is NEVER stored under tests/golden/ref/ and pins nothing
(tests/test_oracle_golden.py::test_fixture_recipe_dry_run runs it in a temporary directory).
usage: oracle_as_elprep.py <workdir>"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from elprep_amd.batch import parse_cigar  # noqa: E402
from elprep_amd.engine import dup_metrics_report  # noqa: E402  (host text code; no device involved)
from tools.ref import cases  # noqa: E402


def main():
    w = sys.argv[1]
    case = json.load(open(os.path.join(w, "case.json")))
    B = cases.build(case)
    o = cases.oracle_outputs(case, B)
    b = B.b
    b.flag[:] = np.asarray(o["flags"], dtype=b.flag.dtype)
    b.qual[:] = o["qual"]
    b.mapq[:] = np.asarray(o["mapq"], dtype=b.mapq.dtype)
    if case["kind"] == "cleansam":  # CleanSam rewrites CIGARs: rebuild the column
        ops = [parse_cigar(c) for c in o["cigar"]]
        b.cigar = np.concatenate(ops).astype(np.uint32) if ops else b.cigar
        b.cigar_off = np.concatenate([[0], np.cumsum([len(x) for x in ops])]).astype(np.uint64)
    header = [ln for ln in open(os.path.join(w, "in.sam")) if ln.startswith("@")]
    with open(os.path.join(w, "out.sam"), "w") as f:
        f.writelines(header)
        for i in o["order"]:
            f.write(cases.sam_line(b, int(i), B.rg_ids, B.ref_names) + "\n")
    if o["metrics_ctr"] is not None:
        open(os.path.join(w, "metrics.txt"), "w").write(dup_metrics_report(o["metrics_ctr"], B.h.lib_names, "dry run"))
    if o["recal_txt"] is not None:
        open(os.path.join(w, "recal.txt"), "w").write(o["recal_txt"])


if __name__ == "__main__":
    main()
