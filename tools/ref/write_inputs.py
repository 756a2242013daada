#!/usr/bin/env python3
"""Inputs for a run of the REAL elPrep (tools/ref/make_fixtures.sh, tools/ref/bundle.sh): per case a directory with in.sam, for the BQSR
cases ref.fasta + sites.bed, and case.json (how to rebuild the same reads: everything is a function of tools/synth presets or of the
hand-derived cases under tests/).  The cases are listed in tools/ref/cases.py.
usage: write_inputs.py <outdir> [pairs] [seed_index]           the round-5 form: ONE `filter` case on synthetic reads, written into <outdir>
       write_inputs.py --all <outdir> [pairs]                   every case of tools/ref/cases.py, one sub-directory each"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools.ref import cases  # noqa: E402
from tools.ref.cases import sam_line as _sam_line  # noqa: E402


def sam_line(b, i, h, names):  # (the round-5 signature, kept for its callers)
    return _sam_line(b, i, h.rg_ids, names)


def main():
    if sys.argv[1] == "--all":
        out = sys.argv[2]
        pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
        for c in cases.all_cases(pairs):
            B = cases.write_case(c, os.path.join(out, c["name"]))
            print(f"{c['name']}: {B.b.n} records")
        return
    out = sys.argv[1]
    pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    seed_index = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    c = {"name": f"filter_tiny_seed{seed_index}", "kind": "filter", "synth": {"genome": "tiny", "seed_index": seed_index, "pairs": pairs, "p_frag": 0.02}}
    B = cases.write_case(c, out)
    print(f"wrote {B.b.n} records to {out}")


if __name__ == "__main__":
    main()
