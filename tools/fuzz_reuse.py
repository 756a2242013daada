"""ONE context, many read sets (GPU box): a context's buffers only grow and it caches facts about the staged columns (one length,
quality hint, adapted values, tables, LUT, sort) - every other parity test and fuzz tool makes a fresh context per read set.  A session
keeps one Engine and runs the whole path on read sets of very different sizes and shapes (ragged / one length, few / many qualities,
1-30 000 records), elp_reset between them, kernel choices redrawn per read set; every output against the oracle.
usage: python tools/fuzz_reuse.py [first_seed] [n_sessions] [read sets per session] [sfm]   (exit code 1 on a mismatch)
"sfm": every read set is the group-split context of an `sfm` rank - the reads of two contig groups of the synthetic genome in aligner
order, split ids in the split column, the spread reads' copies tagged sr among them (mark duplicates in aligner order WITH records on
the table path: the listed inserts of round 5) - 100 to 60 000 records, the oracle run split by split."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle as orc  # the checker  # noqa: E402
from elprep_amd.engine import BqsrTables, Engine  # noqa: E402
from tests.test_gpu_ragged import _random_case  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
sessions = int(sys.argv[2]) if len(sys.argv) > 2 else 6
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 8
bad = 0
if len(sys.argv) > 4 and sys.argv[4] == "sfm":
    from elprep_amd import sfm
    from elprep_amd.batch import Batch
    from tools import synth
    for s in range(first, first + sessions):
        srng = np.random.default_rng(s)
        base = synth.config("tiny", s)
        base.n_lanes = int(srng.choice([1, 4, 9]))
        h = base.header()
        gof, G = sfm.contig_groups(base.ref_len, 80000)
        ranges = sfm.group_ranges(gof, G)
        refs = [synth.reference(base, r) for r in range(h.n_ref)]
        sites = [orc.flatten(orc.sort_by_start(synth.known_sites_raw(base, r))) for r in range(h.n_ref)]
        e = Engine(h, 0)
        for r in range(h.n_ref):
            e.set_reference(r, refs[r])
            e.set_known_sites(r, sites[r])
        for k in range(rounds):
            seed = 1000 * s + k
            rng = np.random.default_rng(seed)
            parts = []
            for g in range(1, G + 1):
                c = synth.config("tiny", s)
                c.n_lanes = base.n_lanes
                c.seed = base.seed + 7919 * g + 13 * k + 1
                c.ref_seed = base.seed
                c.home_lo, c.home_hi = ranges[g - 1]
                c.p_frag = float(rng.choice([0.0, 0.02, 0.3]))
                c.p_dup = float(rng.choice([0.05, 0.1, 0.5]))
                c.p_mate_unmapped = float(rng.choice([0.0, 0.01, 0.1]))
                c.qual_mode = base.qual_mode
                parts.append(synth.generate(c, 0, int(rng.choice([25, 400, 3000, 15000]))))
            b = Batch.concat(parts)
            g_of, sp = sfm.split_records(b, gof)
            p = sfm.with_sr(b, sp, g_of)
            tuning = {"radix_tile": int(rng.integers(0, 4)), "mate_path": int(rng.choice([0, 0, 0, 1, 2])), "pair_table_slots": int(rng.choice([0, 0, 16, 1024])),
                      "count_kernel": int(rng.choice([0, 0, 1, 3])), "apply_kernel": int(rng.choice([0, 0, 1, 3])), "md_fused": int(rng.choice([0, 0, 1]))}
            for key, v in tuning.items():
                e.set_tuning(key, v)
            e.reset()
            cuts = np.linspace(0, p.n, int(rng.integers(1, 4)) + 1).astype(int)
            for lo, hi in zip(cuts[:-1], cuts[1:]):
                if hi > lo:
                    e.stage(p.take(np.arange(lo, hi)))
            flags = e.mark_duplicates(True)
            e.sort_coordinate(fetch=False)
            ctr = e.dup_metrics(100)
            qt, ct, xt = e.recalibrate(500)
            lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
            qual = e.apply_bqsr(lut, present, 500)
            oflags = np.zeros(p.n, np.uint16)
            oq = oc = ox = octr = None
            for sid in np.unique(p.split):  # the reference runs one `filter` process per split file
                sel = np.nonzero(p.split == sid)[0]
                sub = p.take(sel)
                perm = orc.sort_coordinate(sub, orc.mark_duplicates(sub, h))
                fl, c7, _ = orc.dup_metrics(sub, h, perm, 100)
                q, c2, x = orc.bqsr_gather(sub, h, orc.BqsrRef(refs, sites), fl, 500)
                oflags[sel] = fl
                oq = q if oq is None else oq + q
                oc = c2 if oc is None else oc + c2
                ox = x if ox is None else ox + x
                octr = c7 if octr is None else octr + c7
            oqual = orc.BqsrFinal(oq, oc, ox, 500).apply(p, h, 0)
            ok = (np.array_equal(flags, oflags), np.array_equal(ctr, octr), np.array_equal(qt, oq) and np.array_equal(ct, oc) and np.array_equal(xt, ox),
                  np.array_equal(qual, oqual))
            print(f"session {s} ({base.n_lanes} read groups) sfm read set {k}: {p.n} records, {int(p.has_sr.sum())} tagged copies, {tuning}: "
                  f"flags {ok[0]} metrics {ok[1]} tables {ok[2]} qual {ok[3]}", flush=True)
            if not all(ok):
                bad += 1
        e.close()
    print("mismatching read sets:", bad)
    sys.exit(1 if bad else 0)
for s in range(first, first + sessions):
    srng = np.random.default_rng(s)
    n_cov = int(srng.choice([2, 3, 4, 17, 33]))
    e = None
    for k in range(rounds):
        seed = 1000 * s + k
        rng = np.random.default_rng(seed)
        n = int(rng.choice([3, 40, 600, 2500, 2900, 6000, 30000]))
        quals = [2, 5, 6, 12, 23, 37, 41] if rng.random() < 0.6 else list(range(2, 45))
        if rng.random() < 0.5:
            length = int(rng.integers(17, 261))
            b, h, refs, sites = _random_case(seed, n, quals=quals, n_cov=n_cov, len_mix=((length, length, 1.0),))
            if b.n > 4 and rng.random() < 0.7:
                b = b.take(np.arange(b.n - 2))  # without the generator's two odd records: a read set of ONE length
        else:
            b, h, refs, sites = _random_case(seed, n, quals=quals, n_cov=n_cov)
        tuning = {"radix_tile": int(rng.integers(0, 4)), "sort_pairs": int(rng.integers(0, 2)), "tie_rounds": int(rng.integers(0, 2)),
                  "mate_path": int(rng.choice([0, 0, 1, 2])), "pair_table_slots": int(rng.choice([0, 0, 16, 1024])),
                  "count_kernel": int(rng.choice([0, 0, 1, 2, 3])), "apply_kernel": int(rng.choice([0, 0, 1, 3])), "score_kernel": int(rng.choice([0, 0, 1])),
                  "md_fused": int(rng.choice([0, 0, 1]))}
        if e is None:
            e = Engine(h, 0)
        for key, v in tuning.items():
            e.set_tuning(key, v)
        e.reset()
        cuts = np.linspace(0, b.n, int(rng.integers(1, 4)) + 1).astype(int)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            if hi > lo:
                e.stage(b.take(np.arange(lo, hi)))
        for r in range(h.n_ref):
            e.set_reference(r, refs[r])
            e.set_known_sites(r, sites[r])
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
        lens = set(np.diff(b.qual_off).tolist())
        print(f"session {s} ({n_cov} covariates) read set {k}: {b.n} records, {'one length' if len(lens) == 1 else 'ragged'}, {len(quals)} qualities, {tuning}: "
              f"flags {ok[0]} perm {ok[1]} metrics {ok[2]} tables {ok[3]} qual {ok[4]}", flush=True)
        if not all(ok):
            bad += 1
    e.close()
print("mismatching read sets:", bad)
sys.exit(1 if bad else 0)
