#!/usr/bin/env python3
"""A STAND-IN for the elprep binary in a dry run of tools/ref/make_fixtures.sh: writes out.sam, metrics.txt and recal.txt for a work
directory of write_inputs.py - from the CPU oracle, in the formats `elprep filter` writes them - so that the recipe's plumbing (SAM
writer, record identification in collect.py, the comparison of tests/test_oracle_golden.py) can be exercised where no Go toolchain
exists.  What it produces is the oracle compared with itself: it is NEVER stored under tests/golden/ref/ and pins nothing
(tests/test_oracle_golden.py::test_fixture_recipe_dry_run runs it in a temporary directory).
usage: oracle_as_elprep.py <workdir>"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402
from elprep_amd.engine import dup_metrics_report  # noqa: E402  (host text code; no device involved)
from tools import synth  # noqa: E402
from tools.ref.write_inputs import sam_line  # noqa: E402


def main():
    w = sys.argv[1]
    case = json.load(open(os.path.join(w, "case.json")))
    cfg = synth.config(case["genome"], case["seed_index"])
    cfg.p_frag = case["p_frag"]
    b = synth.generate(cfg, 0, case["pairs"])
    h = cfg.header()
    refs = [synth.reference(cfg, r) for r in range(h.n_ref)]
    sites = [orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))) for r in range(h.n_ref)]
    flags = orc.mark_duplicates(b, h)
    perm = orc.sort_coordinate(b, flags)
    _, ctr, _ = orc.dup_metrics(b, h, perm, 100)
    qt, ct, xt = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), flags, 500)
    fin = orc.BqsrFinal(qt, ct, xt, 500)
    qual = fin.apply(b, h, 0)
    b.flag[:] = flags
    b.qual[:] = qual
    header = [ln for ln in open(os.path.join(w, "in.sam")) if ln.startswith("@")]
    with open(os.path.join(w, "out.sam"), "w") as f:
        f.writelines(header)
        for i in perm[:orc.num_sorted(b)]:
            f.write(sam_line(b, int(i), h, cfg.ref_names) + "\n")
    open(os.path.join(w, "metrics.txt"), "w").write(dup_metrics_report(ctr, h.lib_names, "dry run"))
    open(os.path.join(w, "recal.txt"), "w").write(fin.report(h.cov_names, "GATK"))


if __name__ == "__main__":
    main()
