"""Randomised parity sweep (GPU box): the whole path through the C ABI against the oracle on many small seeded read sets with varied
generator settings - more seeds than the test suite holds, for the rare interleavings a fixed seed never meets.
usage: python tools/fuzz_parity.py [first_seed] [n_seeds]   (exit code 1 on the first mismatch)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as orc  # the checker  # noqa: E402
from elprep_amd.engine import BqsrTables, Engine  # noqa: E402
from tools import synth  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    cfg = synth.config("tiny", seed)
    cfg.p_frag = float(rng.choice([0.0, 0.02, 0.3, 1.0]))
    cfg.p_dup = float(rng.choice([0.05, 0.1, 0.5]))
    cfg.p_mate_unmapped = float(rng.choice([0.0, 0.01, 0.1]))
    cfg.qual_mode = int(rng.integers(0, 2))
    cfg.n_lanes = int(rng.choice([4, 4, 4, 1, 9, 20, 40]))  # read groups = BQSR covariates (round 5: any number of them)
    pairs = int(rng.choice([50, 700, 5000, 30000]))
    b = synth.generate(cfg, 0, pairs)
    h = cfg.header()
    refs = [synth.reference(cfg, r) for r in range(h.n_ref)]
    sites = [orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))) for r in range(h.n_ref)]
    # kernel choices at random too (elp_set_tuning): every choice must give the oracle's bytes
    tuning = {"radix_tile": int(rng.integers(0, 4)), "sort_pairs": int(rng.integers(0, 2)), "tie_rounds": int(rng.integers(0, 2)),
              "mate_path": int(rng.choice([0, 0, 1, 2])), "pair_table_slots": int(rng.choice([0, 0, 16, 1024])),
              "count_kernel": int(rng.choice([0, 0, 1, 2, 3])), "apply_kernel": int(rng.choice([0, 0, 1, 3])), "score_kernel": int(rng.choice([0, 0, 1])),
              "md_fused": int(rng.choice([0, 0, 1]))}
    e = Engine(h, 0, tuning=tuning)
    cuts = np.linspace(0, b.n, int(rng.integers(1, 5)) + 1).astype(int)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        e.stage(b.take(np.arange(lo, hi)))
    for r in range(h.n_ref):
        e.set_reference(r, refs[r])
        e.set_known_sites(r, sites[r])
    at_once = bool(rng.integers(0, 2))  # round 6: the sort, the metrics pass and the BQSR chain driven at once from three host threads
    e.sort_ahead(bool(rng.integers(0, 2)))  # elp_sort_ahead: the sort's key passes queued from inside mark duplicates
    if at_once:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(2) as pool:
            e.mark_duplicates(True, fetch=False)
            st, mx = pool.submit(e.sort_coordinate), pool.submit(e.dup_metrics, 100)
            qt, ct, xt = e.recalibrate(500)
            lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
            qual = e.apply_bqsr(lut, present, 500)
            perm, ctr = st.result(), mx.result()
        flags = e.flags()
    else:
        flags = e.mark_duplicates(True)
        perm = e.sort_coordinate()
        ctr = e.dup_metrics(100)
        qt, ct, xt = e.recalibrate(500)
        lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
        qual = e.apply_bqsr(lut, present, 500)
    oflags = orc.mark_duplicates(b, h)
    operm = orc.sort_coordinate(b, oflags)
    _, octr, _ = orc.dup_metrics(b, h, operm, 100)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), oflags, 500)
    oqual = orc.BqsrFinal(oq, oc, ox, 500).apply(b, h, 0)
    ok = (np.array_equal(flags, oflags), np.array_equal(perm, operm), np.array_equal(ctr, octr),
          np.array_equal(qt, oq) and np.array_equal(ct, oc) and np.array_equal(xt, ox), np.array_equal(qual, oqual))
    e.close()
    print(f"seed {seed}: {b.n} records, p_frag {cfg.p_frag}, p_dup {cfg.p_dup}, quals {cfg.qual_mode}, {tuning}, at once {at_once}: "
          f"flags {ok[0]} perm {ok[1]} metrics {ok[2]} tables {ok[3]} qual {ok[4]}", flush=True)
    if not all(ok):
        bad += 1
print("mismatching seeds:", bad)
sys.exit(1 if bad else 0)
