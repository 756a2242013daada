"""GPU parity tests (-m gpu) added in round 2: sr-tagged copies, DeleteOrStore toggling for more than two primary records per
QNAME, the optical-duplicate list cap and large duplicate sets, the bench workload itself against the oracle."""
import numpy as np
import pytest

import oracle as orc
from elprep_amd.batch import Batch, Header, batch_from_records
from elprep_amd.engine import BqsrTables, Engine, ElpError
from tests import kat_cases
from tests.common import dataset

pytestmark = pytest.mark.gpu


def test_sr_tagged_copies_hand_derived():
    """tests/kat_cases.sr_case through the C ABI: `filter` on everything, then `sfm` with the two group splits in ONE context
    (split ids 1 and 2) and the spread split in another; counters are the hand-derived ones, flags those of the oracle run split by
    split (the tagged copies of a spread pair must not pair up in the shared context)."""
    whole, splits, expected, dups = kat_cases.sr_case()
    h = kat_cases.header2()
    e = Engine(h)
    e.stage(whole)
    assert np.array_equal(e.sort_coordinate(), orc.sort_coordinate(whole))
    flags = e.mark_duplicates(True)
    assert kat_cases.flagged_names(whole, flags) == dups["filter"]
    assert e.dup_metrics(100)[0].tolist() == expected
    e.close()
    a, b = splits["A"], splits["B"]
    a.split[:] = 1
    b.split[:] = 2
    both = Batch.concat([a, b])
    eg, es = Engine(h), Engine(h)
    eg.stage(both)
    es.stage(splits["spread"])
    perm = eg.sort_coordinate()
    n_out = eg.n_sorted
    assert n_out == both.n - 4 and not both.has_sr[perm[:n_out]].any() and both.has_sr[perm[n_out:]].all()
    # group A's output, then group B's (refid order): what the two filter runs write, concatenated
    want = np.concatenate([orc.sort_coordinate(a)[:orc.num_sorted(a)], a.n + orc.sort_coordinate(b)[:orc.num_sorted(b)]])
    assert np.array_equal(perm[:n_out], want)
    fg = eg.mark_duplicates(True)
    assert np.array_equal(fg, np.concatenate([orc.mark_duplicates(a, h), orc.mark_duplicates(b, h)]))
    fs = es.mark_duplicates(True)
    assert kat_cases.flagged_names(splits["spread"], fs) == dups["spread"]
    cg, cs = eg.dup_metrics(100), es.dup_metrics(100)
    assert cg[0].tolist() == [1, 2, 0, 1, 1, 0, 0] and cs[0].tolist() == [0, 2, 0, 0, 0, 1, 0]
    assert (cg + cs)[0].tolist() == expected
    eg.close()
    es.close()


def test_sort_runs_behind_mark_duplicates():
    b, h, order, dup = kat_cases.sort_sees_duplicate_bits_case()
    e = Engine(h)
    e.stage(b)
    assert e.sort_coordinate().tolist() == [4, 0, 2, 5, 1, 3]
    flags = e.mark_duplicates(True)
    assert np.nonzero(flags & 0x400)[0].tolist() == dup
    assert e.sort_coordinate().tolist() == order
    e.close()


def test_delete_or_store_toggling_hand_derived():
    h = kat_cases.header2()
    for k, (b, want) in enumerate(kat_cases.toggling_cases()):
        e = Engine(h)
        e.stage(b)
        flags = e.mark_duplicates(True)
        assert np.nonzero(flags & 0x400)[0].tolist() == want, k
        assert np.array_equal(flags, orc.mark_duplicates(b, h)), k
        e.close()


@pytest.mark.parametrize("where", ["adjacent", "anywhere"])
def test_third_primary_record_per_qname_against_oracle(where):
    """A synthetic batch in which 300 pairs get a third primary mapped record with the same QNAME (a copy of one mate, moved), next
    to the pair or anywhere in staging order, in front of it or behind it: flags and counters equal the oracle's sequential run."""
    cfg, b, h, refs, sites = dataset("tiny", 5000, 5, 0.02)
    rng = np.random.default_rng(3)
    cand = np.nonzero(((b.flag & 0x904) == 0) & ((b.flag & 0x9) == 0x1))[0]
    picks = rng.choice(cand, 300, replace=False)
    order = list(range(b.n))
    extra_src = []
    for k, i in enumerate(picks):
        extra_src.append(int(i))
        new_id = b.n + k
        if where == "adjacent":
            at = order.index(int(i)) + int(rng.integers(0, 2))
        else:
            at = int(rng.integers(0, len(order) + 1))
        order.insert(at, new_id)
    ext = b.take(np.asarray(extra_src))
    ext.pos[:] = np.maximum(1, ext.pos + rng.integers(-40, 40, ext.n)).astype(np.int32)  # some land on the old key, most elsewhere
    pb = Batch.concat([b, ext]).take(np.asarray(order))
    e = Engine(h)
    e.stage(pb)
    oflags = orc.mark_duplicates(pb, h)
    flags = e.mark_duplicates(True)
    assert np.array_equal(flags, oflags)
    operm = orc.sort_coordinate(pb, oflags)  # the sort runs behind the mark-duplicates filter: modFlag sees the duplicate bits
    assert np.array_equal(e.sort_coordinate(), operm)
    _, octr, ohist = orc.dup_metrics(pb, h, operm, 100, hist_len=16)
    ctr, hist = e.dup_metrics(100, hist_len=16)
    assert np.array_equal(ctr, octr) and np.array_equal(hist, ohist)
    e.close()


def _pileup(n_fwd_first, n_rev_first, rng):
    """one duplicate set: pairs on one pair key (0:1000 forward, 0:1200 reverse); in n_fwd_first of them the forward read is the
    first of the pair (so it is the read that is listed), in n_rev_first the reverse read; QNAMEs carry tile/x/y with optical
    clusters (few tiles, small coordinate range)."""
    n = n_fwd_first + n_rev_first
    first_fwd = np.zeros(n, dtype=bool)
    first_fwd[:n_fwd_first] = True
    rng.shuffle(first_fwd)
    tile = rng.integers(1101, 1109, n)
    x, y = rng.integers(1000, 3000, n), rng.integers(1000, 3000, n)
    names = [b"P%d:1:FC:1:%d:%d:%d" % (k, tile[k], x[k], y[k]) for k in range(n)]
    lens = np.fromiter((len(s) for s in names), dtype=np.int64, count=n)
    qname = np.frombuffer(b"".join(s + s for s in names), dtype=np.uint8)
    N = 2 * n
    qoff = np.zeros(N + 1, np.uint64)
    np.cumsum(np.repeat(lens, 2), out=qoff[1:])
    L = 8
    flag = np.empty(N, np.uint16)
    flag[0::2] = np.where(first_fwd, 99, 163)   # forward mate: first (99) or last (163) of the pair
    flag[1::2] = np.where(first_fwd, 147, 83)   # reverse mate
    pos = np.empty(N, np.int32); pos[0::2] = 1000; pos[1::2] = 1200
    pnext = np.empty(N, np.int32); pnext[0::2] = 1200; pnext[1::2] = 1000
    tlen = np.empty(N, np.int32); tlen[0::2] = 200 + L; tlen[1::2] = -(200 + L)
    qual = np.full(N * L, 30, np.uint8)
    qual[:2 * L] = 40  # pair 0 is the best pair: the origin
    off = np.arange(N + 1, dtype=np.uint64)
    return Batch(refid=np.zeros(N, np.int32), pos=pos, next_refid=np.zeros(N, np.int32), pnext=pnext, tlen=tlen, flag=flag,
                 mapq=np.full(N, 60, np.uint8), rgid=np.asarray(rng.integers(0, 2, n).repeat(2), np.uint16), has_sr=np.zeros(N, np.uint8),
                 l_seq=np.full(N, L, np.uint32), qname_off=qoff, qname=qname, cigar_off=off, cigar=np.full(N, (L << 4) | 0, np.uint32),
                 seq_off=off * np.uint64(L // 2), seq4=np.full(N * L // 2, 0x12, np.uint8), qual_off=off * np.uint64(L), qual=qual)


@pytest.mark.parametrize("n_fwd,n_rev", [(3000, 2500), (300_040, 700)])
def test_large_duplicate_sets_and_the_list_cap(n_fwd, n_rev):
    """Duplicate sets far beyond what one thread evaluates: 5500 listed reads (both strand lists counted by the cooperative
    union-find), and a forward list of more than 300000 reads, which the reference cuts to 300001 entries and counts as 0 optical
    duplicates (filters/mark-optical-duplicates.go:289-299, 328-330) while the reverse list is counted as usual."""
    rng = np.random.default_rng(n_fwd)
    b = _pileup(n_fwd, n_rev, rng)
    h = Header(ref_len=np.array([5000], np.int32), rg_lib=np.array([0, 0], np.uint16), rg_cov=np.array([0, 1], np.uint16))
    e = Engine(h)
    e.stage(b)
    flags = e.mark_duplicates(True)
    oflags, octr, ohist = orc.dup_metrics(b, h, None, 100, hist_len=8)
    assert np.array_equal(flags, oflags)
    assert int(((flags & 0x400) != 0).sum()) == b.n - 2
    ctr, hist = e.dup_metrics(100, hist_len=8)
    assert np.array_equal(ctr, octr) and np.array_equal(hist, ohist)
    assert ctr[0, 5] == b.n // 2 - 1 and ctr[0, 6] > 100
    assert np.array_equal(e.dup_metrics(100), octr)
    e.close()


def test_qname_length_limit():
    h = kat_cases.header2()
    e = Engine(h)
    rec = dict(qname="q" * 1001, flag=0, refid=0, pos=5, cigar="2M", mapq=60, seq="AC", qual=[30, 30], rgid=0)
    with pytest.raises(ElpError, match="QNAME"):
        e.stage(batch_from_records([rec]))
    # a pile-up of long names (1000 bytes, differing in the last bytes) goes through the live-byte tie-break
    recs = [dict(rec, qname="q" * 990 + "%010d" % ((k * 7919) % 100)) for k in range(100)]
    b = batch_from_records(recs)
    e.stage(b)
    assert np.array_equal(e.sort_coordinate(), orc.sort_coordinate(b))
    e.close()


def test_bench_workload_against_the_oracle():
    """The bench configuration itself (genome c3: 24 contigs, hg38 / 12; the generator's read mix) at 2 M reads, every output of
    the path compared with the oracle: adapted values, permutation, flags, counters, the three BQSR tables, every QUAL byte."""
    from bench import flatten_sites
    from tools import synth
    cfg = synth.config("c3")
    h = cfg.header()
    b = synth.generate(cfg, 0, 1_000_000)
    refs = [synth.reference(cfg, r) for r in range(h.n_ref)]
    sites = [flatten_sites(synth.known_sites_raw(cfg, r)) for r in range(h.n_ref)]
    for r in (0, 7, 23):
        assert np.array_equal(sites[r], orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))))
    e = Engine(h)
    cuts = np.linspace(0, b.n, 4).astype(int)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        e.stage(b.take(np.arange(lo, hi)))
    # the reference's order of events: filters (MarkDuplicates) while the records stream in, the sort as the pipeline's
    # Finalize (sam/filter-pipeline.go:116), then the metrics pass over the sorted records
    oflags = orc.mark_duplicates(b, h)
    assert np.array_equal(e.mark_duplicates(True), oflags)
    operm = orc.sort_coordinate(b, oflags)
    assert np.array_equal(e.sort_coordinate(), operm)
    _, octr, _ = orc.dup_metrics(b, h, operm, 100)
    assert np.array_equal(e.dup_metrics(100), octr)
    for r in range(h.n_ref):
        e.set_reference(r, refs[r])
        e.set_known_sites(r, sites[r])
    qt, ct, xt = e.recalibrate(500)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), oflags, 500)
    assert np.array_equal(qt, oq) and np.array_equal(ct, oc) and np.array_equal(xt, ox)
    tb = BqsrTables(qt, ct, xt, 500).finalize()
    lut, present = tb.build_lut(0)
    got = e.apply_bqsr(lut, present, 500)
    assert np.array_equal(got, orc.BqsrFinal(oq, oc, ox, 500).apply(b, h, 0))
    e.close()


def test_device_tables_add_allreduce_fetch():
    """The device-resident side of the C ABI's collective: tables of two contexts of one rank summed in HBM, the all-reduce of a
    group of one (a no-op that still returns the counters), fetch - equal to the host-side sum of the per-context tables."""
    cfg, b, h, refs, sites = dataset("tiny", 4000, 1, 0.05)
    half = b.n // 2 & ~1
    parts = [b.take(np.arange(0, half)), b.take(np.arange(half, b.n))]
    engines, host = [], None
    for p in parts:
        e = Engine(h)
        e.stage(p)
        e.mark_duplicates(True)
        for r in range(h.n_ref):
            e.set_reference(r, refs[r])
            e.set_known_sites(r, sites[r])
        t = e.recalibrate(500)
        host = [x.copy() for x in t] if host is None else [a + x for a, x in zip(host, t)]
        e.recalibrate_device(500)
        engines.append(e)
    e0 = engines[0]
    e0.group_init(0, 1, None)
    e0.tables_add(engines[1])
    ctr = np.arange(21, dtype=np.int64).reshape(3, 7)
    assert np.array_equal(e0.tables_allreduce(ctr), ctr)
    assert np.array_equal(e0.allreduce_i64(ctr), ctr)
    got = e0.tables_fetch()
    for a, x in zip(got, host):
        assert np.array_equal(a, x)
    assert got[0][..., 0].sum() > 0
    for e in engines:
        e.close()


def test_group_of_two_processes():
    """Two processes, one GPU each, the group id handed over in a file (as a Go host would): ncclAllReduce through the C ABI.
    Skipped without two GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    import os, subprocess, sys, tempfile
    d = tempfile.mkdtemp()
    code = r"""
import os, sys, time, numpy as np
sys.path.insert(0, %r)
from elprep_amd.engine import Engine, group_unique_id
from tests import kat_cases
rank, d = int(sys.argv[1]), sys.argv[2]
e = Engine(kat_cases.header2(), rank)
idf = os.path.join(d, "id")
if rank == 0:
    open(idf + ".tmp", "wb").write(group_unique_id()); os.rename(idf + ".tmp", idf)
while not os.path.exists(idf): time.sleep(0.05)
e.group_init(rank, 2, open(idf, "rb").read())
out = e.allreduce_i64(np.arange(1000, dtype=np.int64) * (rank + 1))
assert np.array_equal(out, np.arange(1000, dtype=np.int64) * 3)
print("ok", rank)
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), d], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def _regions_for(cfg, rng):
    out = []
    for ln in cfg.ref_len:
        k = int(rng.integers(0, 40))
        s = np.sort(rng.integers(0, max(ln - 2000, 1), k))
        iv = np.stack([s, s + rng.integers(1, 1500, k)], axis=1).astype(np.int32)
        out.append(orc.flatten(orc.sort_by_start(iv)) if k else iv.reshape(0, 2))
    return out


@pytest.mark.parametrize("sel", [dict(remove_unmapped=True), dict(remove_unmapped_strict=True, min_mapq=20), dict(remove_non_exact=True),
                                 dict(use_regions=True), dict(remove_unmapped=True, min_mapq=1, remove_non_exact=True, use_regions=True)])
def test_fused_predicates_in_front_of_the_path(sel):
    """elp_filter_records against the restated filters (oracle/simple_filters.py), then the whole path on the survivors: a
    rejected record must behave as if it had never been staged (the reference applies these filters before MarkDuplicates)."""
    from oracle import simple_filters as sf
    cfg, b, h, refs, sites = dataset("tiny", 3000, 6, 0.05)
    rng = np.random.default_rng(9)
    sel = dict(sel)
    regions = _regions_for(cfg, rng) if sel.pop("use_regions", False) else None
    keep = sf.keep_mask(b, regions=regions, **sel)
    assert 0 < keep.sum() < b.n
    e = Engine(h)
    e.stage(b)
    assert e.filter_records(regions=regions, **sel) == int((~keep).sum())
    assert e.n_sorted == int(keep.sum())
    kept = np.nonzero(keep)[0]
    kb = b.take(kept)
    oflags = orc.mark_duplicates(kb, h)
    flags = e.mark_duplicates(True)
    assert np.array_equal(flags[kept], oflags) and np.array_equal(flags[~keep], b.flag[~keep])
    operm = orc.sort_coordinate(kb, oflags)
    assert np.array_equal(e.sort_coordinate()[:e.n_sorted], kept[operm])
    _, octr, _ = orc.dup_metrics(kb, h, operm, 100)
    assert np.array_equal(e.dup_metrics(100), octr)
    for r in range(h.n_ref):
        e.set_reference(r, refs[r])
        e.set_known_sites(r, sites[r])
    oq, oc, ox = orc.bqsr_gather(kb, h, orc.BqsrRef(refs, sites), oflags, 500)
    qt, ct, xt = e.recalibrate(500)
    assert np.array_equal(qt, oq) and np.array_equal(ct, oc) and np.array_equal(xt, ox)
    # RemoveDuplicateReads works on the flags as they are now (filters2, at output time)
    n_dup = e.filter_records(remove_duplicates=True)
    assert n_dup == int(((oflags & 0x400) != 0).sum())
    e.close()


def test_split_classify_and_merge_on_device():
    """`elprep split` routing and `elprep merge` order without the payload: device results against the record-by-record
    restatements (oracle/simple_filters.py) of sam/split-merge.go:280-293 and :410-576."""
    from elprep_amd import sfm
    from oracle import simple_filters as sf
    cfg, b, h, refs, sites = dataset("tiny", 4000, 7, 0.03)
    n_groups, gof = orc.contig_groups(cfg.ref_len, 80000)
    e = Engine(h)
    e.stage(b)
    split, spread, counts = e.split_classify(gof, n_groups)
    osplit, ospread = sf.split_records(b, gof)
    assert np.array_equal(split, osplit) and np.array_equal(spread, ospread) and spread.sum() > 20
    assert counts.tolist() == [int((osplit == g).sum()) for g in range(n_groups + 1)] + [int(ospread.sum())]
    e.close()
    # merge: the group splits (with the tagged copies, which drop out) in one context, the spread split in another
    tagged = sfm.with_sr(b, ospread.astype(bool), osplit)
    eg, es = Engine(h), Engine(h)
    eg.stage(tagged)
    sp = b.take(np.nonzero(ospread)[0])
    es.stage(sp)
    pg, ps = eg.sort_coordinate()[:eg.n_sorted], es.sort_coordinate()[:es.n_sorted]
    slots = eg.merge_spread(es)
    gk = [(int(tagged.refid[i]) & 0xFFFFFFFF, int(tagged.pos[i])) for i in pg]
    sk = [(int(sp.refid[i]) & 0xFFFFFFFF, int(sp.pos[i])) for i in ps]
    assert np.array_equal(slots, sf.merge_slots(gk, sk))
    # and it is what the host-side order model (sfm.merge_order, tested against the reference loop on CPU) gives for the mapped part
    mapped = [k for k in gk if k[0] != 0xFFFFFFFF]
    code = sfm.merge_order(np.asarray([k[0] for k in mapped], np.int64), np.asarray([k[1] for k in mapped], np.int64),
                           np.asarray([k[0] for k in sk], np.int64), np.asarray([k[1] for k in sk], np.int64))
    assert np.array_equal(np.nonzero(code < 0)[0], slots.astype(np.int64))
    eg.close()
    es.close()


@pytest.mark.parametrize("pinned", [False, True])
def test_stage_from_bam_and_emit_sorted_bam(pinned):
    """BAM records in, BAM records out (sam/bam-files.go:299-400, 635-737 on the device).  The staged columns must be what elp_stage
    makes of the same batch (every output of the path equal), and the emitted stream byte-equal to the oracle's formatBamAlignment
    of the sorted, duplicate-marked, recalibrated records with the optional fields re-encoded as elPrep does."""
    import ctypes
    from elprep_amd import _lib
    from elprep_amd import sfm
    cfg, b, h, refs, sites = dataset("tiny", 3000, 8, 0.04)
    # a few sr-tagged copies, as in a contig-group split file
    b = sfm.with_sr(b, (np.arange(b.n) % 37 == 5) & ((b.flag & 0x904) == 0) & ((b.flag & 0x9) == 1))
    raw = orc.bam_encode(b, h.rg_ids)
    L = _lib.hip()
    if pinned:
        ptr = L.elp_pinned_alloc(raw.size)
        buf = np.frombuffer((ctypes.c_uint8 * raw.size).from_address(ptr), dtype=np.uint8)
        buf[:] = raw
    else:
        buf = raw
    e, e2 = Engine(h), Engine(h)
    e.set_read_group_ids(h.rg_ids)
    # split the byte stream at a record boundary in the middle
    off, p = [], 0
    while p < raw.size:
        off.append(p)
        p += 4 + int(raw[p:p + 4].view(np.uint32)[0])
    mid = off[len(off) // 2]
    e.stage_bam(buf[:mid])  # record starts by walking the block_size chain
    ro = np.asarray(off[len(off) // 2:] + [raw.size], dtype=np.uint64) - np.uint64(mid)
    assert np.array_equal(ro, orc.bam_offsets(b, h.rg_ids)[len(off) // 2:] - np.uint64(mid))
    e.stage_bam(buf[mid:], rec_off=ro)  # record starts handed over
    e2.stage(b)
    assert e.n == b.n and e.n_sorted == e2.n_sorted == orc.num_sorted(b)
    up, sc = e.adapted()
    up2, sc2 = e2.adapted()
    assert np.array_equal(up, up2) and np.array_equal(sc, sc2)
    oflags = orc.mark_duplicates(b, h)
    assert np.array_equal(e.mark_duplicates(True), oflags)
    operm = orc.sort_coordinate(b, oflags)
    perm = e.sort_coordinate()
    assert np.array_equal(perm[:e.n_sorted], operm[:e.n_sorted])
    _, octr, _ = orc.dup_metrics(b, h, operm, 100)
    assert np.array_equal(e.dup_metrics(100), octr)
    for r in range(h.n_ref):
        e.set_reference(r, refs[r])
        e.set_known_sites(r, sites[r])
    qt, ct, xt = e.recalibrate(500)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), oflags, 500)
    assert np.array_equal(qt, oq) and np.array_equal(ct, oc) and np.array_equal(xt, ox)
    lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
    qual = e.apply_bqsr(lut, present, 500)
    oqual = orc.BqsrFinal(oq, oc, ox, 500).apply(b, h, 0)
    assert np.array_equal(qual, oqual)
    got = e.emit_sorted_bam()
    want = orc.bam_encode(b, h.rg_ids, order=operm[:orc.num_sorted(b)], flags=oflags, qual=oqual, normalize_tags=True)
    assert got.size == want.size and np.array_equal(got, want)
    assert not np.array_equal(orc.bam_encode(b, h.rg_ids, order=operm[:orc.num_sorted(b)], flags=oflags, qual=oqual), want)  # the re-encoding matters
    e.close()
    e2.close()
    if pinned:
        del buf
        L.elp_pinned_free(ptr)


def test_stage_bam_rejects_malformed_input():
    cfg, b, h, refs, sites = dataset("tiny", 50, 8, 0.0)
    raw = orc.bam_encode(b, h.rg_ids)
    e = Engine(h)
    e.set_read_group_ids(h.rg_ids)
    with pytest.raises(ElpError, match="block_size|truncated"):
        e.stage_bam(raw[:-3])
    e2 = Engine(h)
    e2.set_read_group_ids(["nope"] * len(h.rg_ids))
    with pytest.raises(ElpError, match="read group"):
        e2.stage_bam(raw)
    e.close()
    e2.close()


def test_full_quality_range_tables_and_apply():
    """~40 distinct quality values x 4 read groups: the BQSR count takes its one-workgroup-per-CU form (observation-only cycle
    cells, mismatches by global atomics), the apply its two-level LUT with 16-bit row offsets; tables and qualities vs the oracle."""
    from tools import synth
    cfg = synth.config("tiny", 9)
    cfg.qual_mode = 1
    b = synth.generate(cfg, 0, 20000)
    h = cfg.header()
    assert np.unique(b.qual).size >= 30
    refs = [synth.reference(cfg, r) for r in range(h.n_ref)]
    sites = [orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))) for r in range(h.n_ref)]
    e = Engine(h)
    e.stage(b)
    flags = e.mark_duplicates(True)
    for r in range(h.n_ref):
        e.set_reference(r, refs[r])
        e.set_known_sites(r, sites[r])
    qt, ct, xt = e.recalibrate(500)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), flags, 500)
    assert np.array_equal(qt, oq) and np.array_equal(ct, oc) and np.array_equal(xt, ox)
    assert ct[..., 1].sum() > 1000
    lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
    got = e.apply_bqsr(lut, present, 500)
    assert np.array_equal(got, orc.BqsrFinal(oq, oc, ox, 500).apply(b, h, 0))
    e.close()


@pytest.mark.parametrize("kind", ["no_true_fragments", "only_fragments", "fragments_at_pair_keys"])
def test_fragment_phase_extremes(kind):
    """The fragment phase groups only TRUE fragments and lets the true pairs look their keys up (classifyFragment,
    filters/mark-duplicates.go:210-251).  The extremes: no true fragment at all (the phase is skipped), single-end data (every
    candidate is a fragment: the table holds everything), and fragments that sit exactly on keys of pairs (all of them lose)."""
    from tools import synth
    cfg = synth.config("tiny", 3)
    if kind == "no_true_fragments":
        cfg.p_frag = 0.0; cfg.p_mate_unmapped = 0.0; cfg.p_unmapped_pair = 0.0
    elif kind == "only_fragments":
        cfg.p_frag = 1.0
    else:
        cfg.p_frag = 0.3; cfg.p_dup = 0.5
    b = synth.generate(cfg, 0, 6000)
    h = cfg.header()
    if kind == "fragments_at_pair_keys":
        # turn every fifth true pair read's twin into a fragment at the same key: copy the record, clear the pair flags
        cand = np.flatnonzero(((b.flag & 0x1) != 0) & ((b.flag & 0x8) == 0) & ((b.flag & 0x904) == 0))[::5]
        twin = b.take(cand)
        twin.flag[:] = twin.flag & 0x10  # unpaired, strand kept
        twin.pnext[:] = 0; twin.next_refid[:] = -1; twin.tlen[:] = 0
        b = Batch.concat([b, twin])
    oflags = orc.mark_duplicates(b, h)
    frag = ((oflags & 0x904) == 0) & (((oflags & 0x1) == 0) | ((oflags & 0x8) != 0))
    if kind == "no_true_fragments":
        assert frag.sum() == 0
    elif kind == "only_fragments":
        assert frag.sum() > 0.9 * ((oflags & 0x904) == 0).sum() and ((oflags[frag] & 0x400) != 0).sum() > 0
    else:
        assert ((oflags[b.n - len(cand):] & 0x400) != 0).all()  # a fragment at a pair's key always loses
    e = Engine(h)
    e.stage(b)
    flags = e.mark_duplicates(also_opticals=True)
    assert np.array_equal(flags, oflags)
    perm = e.sort_coordinate()
    assert np.array_equal(perm, orc.sort_coordinate(b, oflags))
    _, octr, _ = orc.dup_metrics(b, h, perm, 100)
    assert np.array_equal(e.dup_metrics(100), octr)
    e.close()
