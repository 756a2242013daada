"""Writes tests/golden/oracle_regression.json: digests of every output of the CPU oracle (oracle/) on a few small seeded read sets.
What it is for: parity of the device path is anchored on the oracle (the reference itself cannot be built or run
here), so the oracle must not drift from round to round unnoticed - any change to oracle/ that changes a result changes a digest, and
tests/test_oracle_golden.py fails until the change is looked at and this file is regenerated on purpose.
These are NOT outputs of the reference (none can be made in this environment): they pin the restatement, the restatement is pinned on
the reference's own vectors (intervals_test_go.json) and the hand-derived cases of tests/test_oracle_kat.py / tests/kat_cases.py.
usage: python tests/golden/make_oracle_regression.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle as orc  # noqa: E402
from tools import synth  # noqa: E402


def digest(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256(a.tobytes()).hexdigest()[:32]


def case(seed, pairs, quals, p_dup, p_frag):
    cfg = synth.config("tiny", seed)
    cfg.qual_mode = quals
    cfg.p_dup = p_dup
    cfg.p_frag = p_frag
    b = synth.generate(cfg, 0, pairs)
    h = cfg.header()
    refs = [synth.reference(cfg, r) for r in range(h.n_ref)]
    sites = [orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))) for r in range(h.n_ref)]
    flags, upos, score = orc.mark_duplicates(b, h, with_adapted=True)
    perm = orc.sort_coordinate(b, flags)
    _, ctr, _ = orc.dup_metrics(b, h, perm, 100)
    qt, ct, xt = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), flags, 500)
    qual = orc.BqsrFinal(qt, ct, xt, 500).apply(b, h, 0)
    return {"seed": seed, "pairs": pairs, "qual_mode": quals, "p_dup": p_dup, "p_frag": p_frag, "records": int(b.n),
            "duplicates": int(((flags & 0x400) != 0).sum()), "observations": int(qt[..., 0].sum()), "mismatches": int(qt[..., 1].sum()),
            "input": digest(np.concatenate([b.pos.view(np.uint8), b.flag.view(np.uint8), b.qual, b.seq4])),
            "upos": digest(upos), "score": digest(score), "flags": digest(flags), "perm": digest(perm), "counters": digest(ctr),
            "qual_table": digest(qt), "cycle_table": digest(ct), "context_table": digest(xt), "qual_out": digest(qual)}


CASES = [(1, 3000, 0, 0.10, 0.0), (2, 3000, 1, 0.10, 0.02), (3, 1500, 0, 0.50, 0.3), (4, 800, 1, 0.05, 1.0)]

if __name__ == "__main__":
    out = {"what": __doc__.split("\n")[0], "cases": [case(*c) for c in CASES]}
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.json"), "w"), indent=1)
    for c in out["cases"]:
        print(c["seed"], c["records"], c["duplicates"], c["observations"])
