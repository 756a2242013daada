"""GPU parity tests (-m gpu): the HIP path through the C ABI against the CPU oracle, bit-exact, on the same seeded inputs."""
import numpy as np
import pytest

import oracle as orc
from elprep_amd.batch import Batch, Header, batch_from_records, NIL16
from elprep_amd.engine import BqsrTables, Engine, ElpError
from tests.common import dataset

pytestmark = pytest.mark.gpu

CASES = [("tiny", 300, 0, 0.0), ("tiny", 4000, 1, 0.05), ("tiny", 20000, 2, 0.02), ("c1", 60000, 0, 0.01)]


def _engine(b, h, chunks=1):
    e = Engine(h)
    if chunks == 1:
        e.stage(b)
    else:  # ragged batches, as the reference's variable batch sizes (sam/filter-pipeline.go:84-85)
        cuts = np.linspace(0, b.n, chunks + 1).astype(int)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            e.stage(b.take(np.arange(lo, hi)))
    return e


@pytest.mark.parametrize("name,pairs,seed,pfrag", CASES)
def test_adapt_sort_markdup_metrics(name, pairs, seed, pfrag):
    cfg, b, h, refs, sites = dataset(name, pairs, seed, pfrag)
    e = _engine(b, h, chunks=3)
    assert e.n == b.n
    # adapted values (filters/mark-duplicates.go:57-110)
    oflags, oupos, oscore = orc.mark_duplicates(b, h, with_adapted=True)
    up, sc = e.adapted()
    assert np.array_equal(up, oupos) and np.array_equal(sc, oscore)
    # coordinate sort: identical permutation (total order + stable ties)
    perm = e.sort_coordinate()
    operm = orc.sort_coordinate(b)
    assert np.array_equal(perm, operm)
    # duplicate flags
    flags = e.mark_duplicates(also_opticals=True)
    assert np.array_equal(flags, oflags)
    assert ((flags & 0x400) != 0).sum() > 0
    # DuplicationMetrics counters incl. optical duplicates
    _, octr, _ = orc.dup_metrics(b, h, operm, 100)
    ctr = e.dup_metrics(100)
    assert np.array_equal(ctr, octr)
    if pairs >= 4000:
        assert ctr[:, 6].sum() > 0  # optical duplicates present
    # the three set-size histograms per library, with a short last bin (3) and a long one
    for hl in (3, 24):
        _, octr2, ohist = orc.dup_metrics(b, h, operm, 100, hist_len=hl)
        ctr2, hist = e.dup_metrics(100, hist_len=hl)
        assert np.array_equal(ctr2, octr2) and np.array_equal(hist, ohist)
        assert hist[:, 0, 1].sum() > 0 and hist[:, 0, 2:].sum() > 0
    e.close()


@pytest.mark.parametrize("mode", ["shuffled", "blocks", "triple_run"])
def test_mates_in_any_staging_order(mode):
    """Mates next to each other pair up without the hash table, all others through it: same flags and metrics either way."""
    cfg, b, h, refs, sites = dataset("tiny", 6000, 3, 0.03)
    rng = np.random.default_rng(11)
    if mode == "shuffled":      # (almost) no mate keeps its neighbour
        idx = rng.permutation(b.n)
    elif mode == "blocks":      # odd-sized blocks in random order: most pairs stay neighbours, those cut by a block border do not
        cuts = np.concatenate([[0], np.sort(rng.choice(np.arange(1, b.n), 400, replace=False)), [b.n]])
        blocks = [np.arange(lo, hi) for lo, hi in zip(cuts[:-1], cuts[1:])]
        idx = np.concatenate([blocks[k] for k in rng.permutation(len(blocks))])
    else:                       # every record twice in a row with cleared pairing flags on the copy: runs of neighbours with one key
        idx = np.repeat(np.arange(b.n), 2)
    pb = b.take(idx)
    if mode == "triple_run":
        pb.flag[1::2] &= np.uint16(0xFFFF ^ 0x1 ^ 0x2 ^ 0x8 ^ 0x20 ^ 0x40 ^ 0x80)  # the copies are single-end fragments
    e = _engine(pb, h, chunks=2)
    oflags = orc.mark_duplicates(pb, h)
    flags = e.mark_duplicates(also_opticals=True)
    assert np.array_equal(flags, oflags)
    operm = orc.sort_coordinate(pb)
    _, octr, _ = orc.dup_metrics(pb, h, operm, 100)
    assert np.array_equal(e.dup_metrics(100), octr)
    e.close()


@pytest.mark.parametrize("name,pairs,seed,pfrag", CASES[:3])
def test_bqsr_gather_apply(name, pairs, seed, pfrag):
    cfg, b, h, refs, sites = dataset(name, pairs, seed, pfrag)
    e = _engine(b, h, chunks=2)
    flags = e.mark_duplicates()
    for r in range(h.n_ref):
        e.set_reference(r, refs[r])
        e.set_known_sites(r, sites[r])
    qt, ct, xt = e.recalibrate(500)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), flags, 500)
    assert qt[..., 0].sum() > 0
    assert np.array_equal(qt, oq) and np.array_equal(ct, oc) and np.array_equal(xt, ox)
    # finalize on the host, apply on the device, compare every recalibrated quality byte
    tb = BqsrTables(qt, ct, xt, 500).finalize()
    for levels, sqq in ((0, ()), (4, (10, 20, 30))):
        lut, present = tb.build_lut(levels, sqq)
        e2 = _engine(b, h)
        got = e2.apply_bqsr(lut, present, 500)
        want = orc.BqsrFinal(oq, oc, ox, 500).apply(b, h, levels, sqq)
        assert np.array_equal(got, want)
        assert (got != b.qual).mean() > 0.5
        e2.close()
    e.close()


def test_sort_large_tie_runs_and_unmapped_block():
    """Runs longer than the all-pairs limit go through the radix tie-break (QNAME bytes, flags, MAPQ, mate fields, TLEN)."""
    rng = np.random.default_rng(5)
    recs = []
    for k in range(700):  # one pile-up at (0, 100, +) with many QNAME prefixes/lengths and exact QNAME ties
        name = "r%d" % rng.integers(0, 150) + ("x" * int(rng.integers(0, 3)))
        paired = bool(rng.integers(0, 2))
        recs.append(dict(qname=name, flag=(0x1 | (0x40 if rng.integers(0, 2) else 0x80)) if paired else 0, refid=0, pos=100, cigar="10M",
                         mapq=int(rng.integers(0, 4)), next_refid=int(rng.integers(-1, 2)), pnext=int(rng.integers(0, 5)),
                         tlen=int(rng.integers(-3, 4)), seq="A" * 10, qual=[30] * 10, rgid=0))
    for k in range(300):  # unmapped block: refid -1, pos 0
        recs.append(dict(qname="u%05d" % rng.integers(0, 200), flag=0x4 | 0x1 | (0x40 if k % 2 else 0x80) | 0x8, refid=-1, pos=0, cigar="*",
                         seq="A" * 10, qual=[30] * 10, rgid=0))
    for k in range(50):
        recs.append(dict(qname="s%d" % k, flag=16 if k % 2 else 0, refid=1, pos=int(rng.integers(1, 30)), cigar="10M", mapq=60, seq="A" * 10,
                         qual=[30] * 10, rgid=0))
    order = rng.permutation(len(recs))
    b = batch_from_records([recs[i] for i in order])
    h = Header(ref_len=np.array([1000, 1000], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))
    e = Engine(h)
    e.stage(b)
    assert np.array_equal(e.sort_coordinate(), orc.sort_coordinate(b))
    e.close()


def test_markdup_exact_ties_and_fragments():
    """Exact (score, QNAME) ties resolve as in the single-threaded reference run: the later arrival wins."""
    q = [30] * 10
    recs = [dict(qname="same", flag=0, refid=0, pos=100, cigar="10M", mapq=60, seq="A" * 10, qual=q, rgid=0) for _ in range(4)]
    recs += [dict(qname="pp", flag=99, refid=0, pos=200, cigar="10M", mapq=60, next_refid=0, pnext=300, tlen=110, seq="A" * 10, qual=q, rgid=0),
             dict(qname="pp", flag=147, refid=0, pos=300, cigar="10M", mapq=60, next_refid=0, pnext=200, tlen=-110, seq="A" * 10, qual=q, rgid=0),
             dict(qname="pq", flag=99, refid=0, pos=200, cigar="10M", mapq=60, next_refid=0, pnext=300, tlen=110, seq="A" * 10, qual=q, rgid=0),
             dict(qname="pq", flag=147, refid=0, pos=300, cigar="10M", mapq=60, next_refid=0, pnext=200, tlen=-110, seq="A" * 10, qual=q, rgid=0),
             dict(qname="lone", flag=99, refid=0, pos=200, cigar="10M", mapq=60, next_refid=0, pnext=300, tlen=110, seq="A" * 10, qual=q, rgid=0),
             dict(qname="frag", flag=0, refid=0, pos=200, cigar="10M", mapq=60, seq="A" * 10, qual=[40] * 10, rgid=NIL16)]
    b = batch_from_records(recs)
    h = Header(ref_len=np.array([1000], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))
    e = Engine(h)
    e.stage(b)
    flags = e.mark_duplicates()
    assert np.array_equal(flags, orc.mark_duplicates(b, h))
    assert [(int(f) & 0x400) != 0 for f in flags[:4]] == [True, True, True, False]
    e.close()


def test_scores_of_very_long_reads():
    """Reads far longer than a step of the score kernel (its per-block LDS words do not cover the group: the atomics path), mixed with
    short ones; scores are compared with the oracle, the low-quality bounds through a second adapt after nothing changed."""
    rng = np.random.default_rng(5)
    recs = []
    for k in range(40):
        L = int(rng.choice([90_000, 130_000, 7, 150, 31]))
        q = rng.choice([2, 2, 14, 15, 30, 40, 93], size=L).tolist()
        recs.append(dict(qname="L%d" % k, flag=0, refid=0, pos=100 + k, cigar="%dM" % L, mapq=60, seq="A" * L, qual=q, rgid=0))
    b = batch_from_records(recs)
    h = Header(ref_len=np.array([400000], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))
    e = Engine(h)
    e.stage(b)
    _, oupos, oscore = orc.mark_duplicates(b, h, with_adapted=True)
    up, sc = e.adapted()
    assert np.array_equal(up, oupos) and np.array_equal(sc, oscore)
    assert np.array_equal(e.mark_duplicates(), orc.mark_duplicates(b, h))
    e.close()


def test_empty_and_single_record():
    h = Header(ref_len=np.array([1000], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))
    e = Engine(h)
    assert e.sort_coordinate().size == 0 and e.mark_duplicates().size == 0
    assert e.dup_metrics().sum() == 0
    b = batch_from_records([dict(qname="a", flag=0, refid=0, pos=5, cigar="4M", mapq=60, seq="ACGT", qual=[30, 30, 2, 2], rgid=0)])
    e.stage(b)
    assert e.sort_coordinate().tolist() == [0]
    assert e.mark_duplicates().tolist() == [0]
    assert e.dup_metrics()[0].tolist() == [1, 0, 0, 0, 0, 0, 0]
    e.close()


def test_errors_surface_like_reference_panics():
    h = Header(ref_len=np.array([1000], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))
    e = Engine(h)
    e.stage(batch_from_records([dict(qname="a", flag=0, refid=0, pos=5, cigar="2M", mapq=60, seq="AC", qual=[30, 99], rgid=0)]))
    with pytest.raises(ElpError, match="Invalid QUAL"):
        e.mark_duplicates()
    e.close()
    e = Engine(h)
    e.stage(batch_from_records([dict(qname="a", flag=0, refid=0, pos=5, cigar="2M", mapq=60, seq="AC", qual=[30, 30], rgid=NIL16)]))
    lut = np.zeros((1, 94, 1001, 17), np.uint8)
    with pytest.raises(ElpError, match="read groups"):
        e.apply_bqsr(lut, np.ones(1, np.uint8), 500)
    e.close()


def test_full_size_properties():
    """Size-independent properties at a larger size than the oracle comparisons: sortedness under CoordinateLess on sampled
    neighbours, idempotence of duplicate marking, permutation validity, table sums."""
    cfg, b, h, refs, sites = dataset("c1", 250000, 3, 0.01)
    e = _engine(b, h, chunks=4)
    perm = e.sort_coordinate()
    assert np.array_equal(np.sort(perm), np.arange(b.n, dtype=np.uint32))
    rng = np.random.default_rng(0)
    for k in rng.integers(0, b.n - 1, 4000):
        assert not orc.coordinate_less(b, int(perm[k + 1]), int(perm[k]))
    f1 = e.mark_duplicates()
    f2 = e.mark_duplicates()  # idempotent: flags already set do not change the outcome
    assert np.array_equal(f1, f2)
    ctr = e.dup_metrics()
    n_primary_mapped = int(((b.flag & 0x904) == 0).sum())
    assert ctr[:, 0].sum() + 2 * ctr[:, 1].sum() <= n_primary_mapped
    assert ctr[:, 2].sum() == int(((b.flag & 0x4) == 0).astype(bool)[(b.flag & 0x900) != 0].sum())
    for r in range(h.n_ref):
        e.set_reference(r, refs[r]); e.set_known_sites(r, sites[r])
    qt, ct, xt = e.recalibrate(500)
    # every counted base is counted once per table (cycle always defined; context missing only next to masked/non-ACGT bases)
    assert np.array_equal(qt[..., 0], ct[..., 0].sum(axis=2)) and np.array_equal(qt[..., 1], ct[..., 1].sum(axis=2))
    assert (xt[..., 0].sum(axis=2) <= qt[..., 0]).all()
    e.close()


def test_c3_scale_properties():
    """Size-independent properties on the bench workload itself (genome c3, 24 contigs) at 12 M reads — far beyond what the oracle
    finishes in seconds: permutation validity and sortedness of sampled neighbours under CoordinateLess, idempotence of duplicate
    marking, duplicate-count plausibility, table identities (every counted base is in the quality table and in exactly one cycle
    cell), apply touches only qualities >= 6 and is a function of the input (two contexts give the same bytes)."""
    from concurrent.futures import ThreadPoolExecutor
    from tools import synth
    cfg = synth.config("c3")
    h = cfg.header()
    with ThreadPoolExecutor(8) as pool:
        parts = list(pool.map(lambda lo: synth.generate(cfg, lo, lo + 500_000), range(0, 6_000_000, 500_000)))
    engines = [Engine(h), Engine(h)]
    for e in engines:
        for p in parts:
            e.stage(p)
    e = engines[0]
    n = e.n
    assert n > 12_000_000
    perm = e.sort_coordinate()
    seen = np.zeros(n, dtype=bool)
    seen[perm] = True
    assert seen.all()
    b = Batch.concat(parts)
    rng = np.random.default_rng(1)
    for k in rng.integers(0, n - 1, 3000):
        assert not orc.coordinate_less(b, int(perm[k + 1]), int(perm[k]))
    f1 = e.mark_duplicates(True)
    f2 = e.mark_duplicates(True)
    assert np.array_equal(f1, f2)
    dup_frac = ((f1 & 0x400) != 0).mean()
    assert 0.05 < dup_frac < 0.2  # the generator makes 10 % of the pairs copies of an earlier pair
    assert (((f1 ^ b.flag) & ~np.uint16(0x400)) == 0).all()  # only the duplicate bit ever changes
    ctr = e.dup_metrics(100)
    assert ctr[:, 6].sum() > 0 and ctr[:, 5].sum() >= ctr[:, 6].sum()
    refs_sites = [(r, synth.reference(cfg, r), orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r)))) for r in range(h.n_ref)]
    for eng in engines:
        for r, ref, st in refs_sites:
            eng.set_reference(r, ref)
            eng.set_known_sites(r, st)
    qt, ct, xt = e.recalibrate(500)
    assert qt[..., 0].sum() > 1_000_000_000
    assert np.array_equal(qt[..., 0], ct[..., 0].sum(axis=2)) and np.array_equal(qt[..., 1], ct[..., 1].sum(axis=2))
    assert (xt[..., 0].sum(axis=2) <= qt[..., 0]).all() and (qt[..., 1] <= qt[..., 0]).all()
    assert qt[:, :6].sum() == 0 and ct[:, :, 500].sum() == 0  # no quality < 6, no cycle 0
    tb = BqsrTables(qt, ct, xt, 500).finalize()
    lut, present = tb.build_lut(0)
    q1 = e.apply_bqsr(lut, present, 500)
    engines[1].mark_duplicates(True)  # flags do not influence apply; same staged input
    q2 = engines[1].apply_bqsr(lut, present, 500)
    assert np.array_equal(q1, q2)
    low = b.qual < 6
    assert np.array_equal(q1[low], b.qual[low])
    assert (q1[~low] >= 1).all() and (q1 != b.qual).mean() > 0.3
    for eng in engines:
        eng.close()
