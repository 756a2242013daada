"""GPU tests of round 3 (-m gpu): the flat-argument entry points a cgo binding calls, rejected sr-tagged copies in front of duplicate
marking, staging from page-locked columns, the device group's table all-reduce through a caller-supplied transport."""
import ctypes as C

import numpy as np
import pytest

import oracle as orc
from elprep_amd import _lib
from elprep_amd.batch import Batch
from elprep_amd.engine import BqsrTables, Engine
from tests.common import dataset

pytestmark = pytest.mark.gpu


def _whole_path(e, b, h, refs, sites):
    flags = e.mark_duplicates(True)
    perm = e.sort_coordinate()
    ctr = e.dup_metrics(100)
    for r in range(h.n_ref):
        e.set_reference(r, refs[r])
        e.set_known_sites(r, sites[r])
    qt, ct, xt = e.recalibrate(500)
    lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
    qual = e.apply_bqsr(lut, present, 500)
    return flags, perm, ctr, (qt, ct, xt), qual


def _oracle_path(b, h, refs, sites):
    oflags = orc.mark_duplicates(b, h)
    operm = orc.sort_coordinate(b, oflags)
    _, octr, _ = orc.dup_metrics(b, h, operm, 100)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), oflags, 500)
    oqual = orc.BqsrFinal(oq, oc, ox, 500).apply(b, h, 0)
    return oflags, operm, octr, (oq, oc, ox), oqual


def _same(got, want):
    flags, perm, ctr, tabs, qual = got
    oflags, operm, octr, otabs, oqual = want
    assert np.array_equal(flags, oflags) and np.array_equal(perm, operm) and np.array_equal(ctr, octr)
    assert all(np.array_equal(a, o) for a, o in zip(tabs, otabs))
    assert np.array_equal(qual, oqual)


def test_flat_entry_points_run_the_whole_path():
    """elp_set_header_columns / elp_stage_columns (ragged batches) give what the struct forms give: every output against the oracle"""
    cfg, b, h, refs, sites = dataset("tiny", 6000, 4, 0.03)
    e = Engine(h, flat_abi=True)
    cuts = [0, 1, 700, 701, 5000, b.n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        e.stage(b.take(np.arange(lo, hi)))
    assert e.n == b.n
    _same(_whole_path(e, b, h, refs, sites), _oracle_path(b, h, refs, sites))
    e.close()


def test_flat_filter_records_and_read_group_ids():
    from oracle import simple_filters as sf
    from tools import synth
    cfg, b, h, refs, sites = dataset("tiny", 2500, 8, 0.05)
    rng = np.random.default_rng(4)
    regions = []
    for r in range(h.n_ref):
        st = np.sort(rng.integers(1, cfg.ref_len[r] - 300, size=12))
        iv = np.stack([st, st + rng.integers(50, 250, size=12)], axis=1)
        regions.append(orc.flatten(orc.sort_by_start(iv)))
    sel = dict(remove_unmapped_strict=True, min_mapq=20, remove_non_exact=True)
    keep = sf.keep_mask(b, regions=regions, **sel)
    assert 0 < keep.sum() < b.n
    for flat in (False, True):
        e = Engine(h, flat_abi=flat)
        e.stage(b)
        assert e.filter_records(regions=regions, **sel) == int((~keep).sum())
        kept = np.nonzero(keep)[0]
        oflags = orc.mark_duplicates(b.take(kept), h)
        flags = e.mark_duplicates(True)
        assert np.array_equal(flags[kept], oflags)
        e.close()
    # RG:Z look-up through the flat id table: staging from BAM bytes gives the same rgid column as the struct form
    buf, off = synth.bam_records(b, h.rg_ids)
    quals = []
    for flat in (False, True):
        e = Engine(h, flat_abi=flat)
        e.set_read_group_ids(h.rg_ids)
        e.stage_bam(buf, rec_off=off)
        assert e.n == b.n
        _same(_whole_path(e, b, h, refs, sites), _oracle_path(b, h, refs, sites))
        e.close()


def test_rejected_tagged_copies_do_not_take_part_in_duplicate_marking():
    """A call of elp_filter_records that rejects ONLY sr-tagged copies (state 1 -> 2): the copies must not knock out fragments any more
    (in the reference they never reach MarkDuplicates: the filters stand in front of it, cmd/filter.go:696-803)."""
    cfg, b, h, refs, sites = dataset("tiny", 3000, 11, 0.3)
    # tag every paired read whose mate is mapped and has MAPQ < 30 as a copy; reject exactly those by MAPQ
    paired = (b.flag & 0x1) != 0
    tag = paired & ((b.flag & 0x908) == 0) & (b.mapq < 30)
    assert tag.sum() > 20
    cols = {name: getattr(b, name) for name in b.__dataclass_fields__}
    cols["has_sr"] = tag.astype(np.uint8)
    cols["mapq"] = np.where(tag, b.mapq, np.maximum(b.mapq, 30)).astype(np.uint8)  # nothing else falls below the threshold
    tb = Batch(**cols)
    e = Engine(h)
    e.stage(tb)
    assert e.filter_records(min_mapq=30) == 0          # the call rejects tagged copies only: they were not part of the output anyway
    assert e.n_sorted == int((~tag).sum())
    flags = e.mark_duplicates(True)
    kept = np.nonzero(~tag)[0]
    oflags = orc.mark_duplicates(tb.take(kept), h)
    assert np.array_equal(flags[kept], oflags)
    # and the copies did matter before the filter: with them the flags differ
    oflags_with = orc.mark_duplicates(tb, h)
    assert not np.array_equal(oflags_with[kept], oflags)
    e.close()


def test_stage_columns_from_page_locked_memory():
    """columns in elp_pinned_alloc memory (the fast route over PCIe): same staged content"""
    cfg, b, h, refs, sites = dataset("tiny", 3000, 5, 0.02)
    L = _lib.hip()
    keep, ptrs = [], {}
    for name in Engine._STAGE_COLS:
        a = np.ascontiguousarray(getattr(b, name))
        p = L.elp_pinned_alloc(max(a.nbytes, 8))
        assert p
        C.memmove(p, a.ctypes.data, a.nbytes)
        keep.append(p)
        ptrs[name] = p
    e = Engine(h)
    e.stage_pointers(b.n, ptrs)
    for p in keep:
        L.elp_pinned_free(p)
    _same(_whole_path(e, b, h, refs, sites), _oracle_path(b, h, refs, sites))
    e.close()


def test_stage_error_leaves_the_context_unchanged():
    """a batch the host-side scan rejects (rgid not in the header) is not committed, although its copies were already on their way"""
    cfg, b, h, refs, sites = dataset("tiny", 1000, 2, 0.0)
    e = Engine(h)
    e.stage(b)
    cols = {name: getattr(b, name).copy() for name in b.__dataclass_fields__}
    cols["rgid"][b.n // 2] = 999
    with pytest.raises(Exception):
        e.stage(Batch(**cols))
    assert e.n == b.n
    _same(_whole_path(e, b, h, refs, sites), _oracle_path(b, h, refs, sites))
    e.close()


# ---- the kernels for read sets of one length (count3.hip, apply3.hip) on adversarial inputs, and against the general kernels
from tests.test_gpu_ragged import _check_gather_apply, _random_case, _stage  # noqa: E402


def _uniform_case(seed, n, length, quals, **kw):
    """tests/test_gpu_ragged.py's generator (soft / hard clips, up to four indels, adaptor read-through, N bases, reads over the contig
    end, low-quality tails) with every read `length` bases long; its two odd records (other lengths) are left out"""
    b, h, refs, sites = _random_case(seed, n, quals, len_mix=((length, length, 1.0),), **kw)
    return b.take(np.arange(b.n - 2)), h, refs, sites


@pytest.mark.parametrize("length,seed", [(150, 1), (151, 2), (16, 3), (17, 4), (33, 5), (100, 6), (250, 7), (1, 8), (48, 9)])
def test_one_length_read_sets(length, seed):
    """the dispatch takes count3 / apply3 (uniform length); every output against the oracle"""
    b, h, refs, sites = _uniform_case(seed, 5000 if length > 8 else 20000, length, quals=[2, 5, 6, 12, 23, 37, 41], ref_len=(6000, 2500, 900))
    assert len(set(np.diff(b.qual_off).tolist())) == 1
    _check_gather_apply(b, h, refs, sites)


def test_one_length_many_contigs_and_reads_at_contig_starts():
    """300 contigs of ~700 bases and 150-base reads: most reads start within a few bases of a contig start or hang over its end (the
    records' general path), every contig pointer comes from the record"""
    b, h, refs, sites = _uniform_case(12, 4000, 150, quals=[2, 9, 22, 35], ref_len=tuple([700 + 13 * (k % 7) for k in range(300)]))
    _check_gather_apply(b, h, refs, sites, chunks=2)


@pytest.mark.parametrize("drop", ["6", "27", "38"])
def test_one_length_quality_hint_incomplete(monkeypatch, drop):
    """a quality missing from the sampled hint - the smallest one >= 6, a middle one, the largest one: the gather retries with the
    exact set; apply3's level 1 is resident from quality 6 whatever the hint says, qualities above its range take the fix-up loop"""
    b, h, refs, sites = _uniform_case(13, 4000, 120, quals=[2, 6, 13, 27, 38])
    monkeypatch.setenv("ELP_TUNE", "qual_hint_drop=" + drop)
    _check_gather_apply(b, h, refs, sites)


def test_one_length_empty_quality_hint(monkeypatch):
    b, h, refs, sites = _uniform_case(14, 3000, 101, quals=[2, 6, 13, 27, 38, 64, 93])
    monkeypatch.setenv("ELP_TUNE", "qual_hint=1")
    _check_gather_apply(b, h, refs, sites)


def test_general_kernels_and_one_length_kernels_agree(monkeypatch):
    """the same staged reads through k_bqsr_count / k_bqsr_apply_flat (forced) and through count3 / apply3: identical tables and bytes"""
    cfg, b, h, refs, sites = dataset("tiny", 30000, 3, 0.02)
    out = []
    for force in ("1", "0", "2", "3"):  # general kernels; the library's choice; one-length count with one table / split by covariate
        e = Engine(h, tuning={"count_kernel": int(force), "apply_kernel": 1 if force == "1" else 0})
        e.stage(b)
        e.mark_duplicates(True)
        for r in range(h.n_ref):
            e.set_reference(r, refs[r])
            e.set_known_sites(r, sites[r])
        qt, ct, xt = e.recalibrate(500)
        lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
        out.append((qt.copy(), ct.copy(), xt.copy(), e.apply_bqsr(lut, present, 500)))
        e.close()
    for other in out[1:]:
        for a, bb in zip(out[0], other):
            assert np.array_equal(a, bb)
    assert (out[0][3] != b.qual).any()


def test_one_length_runs_are_reproducible():
    """two contexts, the same 2 M reads: identical QUAL bytes (the loads of the pipelined kernels are hidden from the compiler,
    elprep_amd/csrc/gload.hpp: a register read before its data has landed shows up as run-to-run differences at this size)"""
    from tools import synth
    cfg = synth.config("c3")
    h = cfg.header()
    parts = [synth.generate(cfg, lo, lo + 250_000) for lo in range(0, 1_000_000, 250_000)]
    refs_sites = [(r, synth.reference(cfg, r), orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r)))) for r in range(h.n_ref)]
    quals, tabs = [], []
    for _ in range(2):
        e = Engine(h)
        for p in parts:
            e.stage(p)
        for r, ref, st in refs_sites:
            e.set_reference(r, ref)
            e.set_known_sites(r, st)
        e.mark_duplicates(True)
        qt, ct, xt = e.recalibrate(500)
        if not tabs:
            lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
        tabs.append((qt.copy(), ct.copy(), xt.copy()))
        quals.append(e.apply_bqsr(lut, present, 500))
        e.close()
    assert all(np.array_equal(a, b_) for a, b_ in zip(tabs[0], tabs[1]))
    assert np.array_equal(quals[0], quals[1])


def test_known_sites_and_reference_in_either_order_and_replaced():
    """the known-site flags live inside the packed reference (bit 2 of a base's nibble): whichever of set_reference / set_known_sites
    comes last, and a site list that replaces another one, must leave exactly the flags of the current list"""
    b, h, refs, sites = _uniform_case(21, 4000, 150, quals=[2, 6, 13, 27, 38], ref_len=(5000, 3000))
    flags = orc.mark_duplicates(b, h)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), flags, 500)
    other = [np.ascontiguousarray(s + 7) for s in sites]  # every interval moved by seven bases
    oq2, oc2, ox2 = orc.bqsr_gather(b, h, orc.BqsrRef(refs, other), flags, 500)
    assert not np.array_equal(oc, oc2)
    e = Engine(h)
    e.stage(b)
    e.mark_duplicates(True)
    for r in range(h.n_ref):  # sites first, reference second
        e.set_known_sites(r, sites[r])
        e.set_reference(r, refs[r])
    qt, ct, xt = e.recalibrate(500)
    assert np.array_equal(ct, oc) and np.array_equal(xt, ox) and np.array_equal(qt, oq)
    for r in range(h.n_ref):  # another list replaces the first one: no flag of the first list may survive
        e.set_known_sites(r, other[r])
    qt, ct, xt = e.recalibrate(500)
    assert np.array_equal(ct, oc2) and np.array_equal(xt, ox2) and np.array_equal(qt, oq2)
    for r in range(h.n_ref):  # the reference set again behind the sites, then the first list again
        e.set_reference(r, refs[r])
    qt, ct, xt = e.recalibrate(500)
    assert np.array_equal(ct, oc2)
    for r in range(h.n_ref):
        e.set_known_sites(r, sites[r])
    qt, ct, xt = e.recalibrate(500)
    assert np.array_equal(ct, oc) and np.array_equal(xt, ox) and np.array_equal(qt, oq)
    e.close()


def test_tables_allreduce_through_a_caller_supplied_transport():
    """elp_group_init_transport: two contexts (two 'ranks' of one process, each on its own thread) sum their device tables through a
    host reduction the test supplies; the extra counters ride behind the tables in the same call (the tail packing of
    elp_bqsr_tables_allreduce) and come back summed on both ranks"""
    import threading
    cfg, b, h, refs, sites = dataset("tiny", 24000, 5, 0.02)
    parts = [b.take(np.arange(0, b.n // 2)), b.take(np.arange(b.n // 2, b.n))]
    engs, own = [], []
    for part in parts:
        e = Engine(h)
        e.stage(part)
        for r in range(h.n_ref):
            e.set_reference(r, refs[r])
            e.set_known_sites(r, sites[r])
        e.recalibrate_device(500)
        own.append([t.copy() for t in e.tables_fetch()])
        engs.append(e)
    assert own[0][1].sum() > 0 and own[1][1].sum() > 0
    shared, barrier = {}, threading.Barrier(2)

    def transport(rank):
        def allreduce(values):
            shared[rank] = values.copy()
            barrier.wait()
            total = shared[0] + shared[1]
            barrier.wait()
            values[:] = total
        return allreduce
    for rank, e in enumerate(engs):
        e.group_init_transport(rank, 2, transport(rank))
    got = [None, None]

    def run(rank):
        got[rank] = engs[rank].tables_allreduce(np.array([rank + 1, 10 * (rank + 1), 7], dtype=np.int64))
    ths = [threading.Thread(target=run, args=(k,)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for rank, e in enumerate(engs):
        assert got[rank].tolist() == [3, 30, 14]
        for have, a, bb in zip(e.tables_fetch(), own[0], own[1]):
            assert np.array_equal(have, a + bb)
        e.close()


def test_copy_records_between_contexts_whole_path():
    """elp_copy_records (the write side of the split routing, device to device): a context that was staged with a third of the reads and
    received the others from another context - in a shuffled order, in three calls - is the context the host would have staged with the
    same records in that order: every output of the path against the oracle on that batch"""
    cfg, b, h, refs, sites = dataset("tiny", 9000, 11, 0.03)
    rng = np.random.default_rng(5)
    k = b.n // 3
    idx = rng.permutation(np.arange(k, b.n))
    a, e = Engine(h), Engine(h)
    a.stage(b)
    e.stage(b.take(np.arange(k)))
    for part in np.array_split(idx, 3):
        e.copy_records_from(a, part)
    assert e.n == b.n
    equiv = b.take(np.concatenate([np.arange(k), idx]))
    _same(_whole_path(e, equiv, h, refs, sites), _oracle_path(equiv, h, refs, sites))
    # the source is untouched: its own path still gives what the oracle gives for ITS order
    _same(_whole_path(a, b, h, refs, sites), _oracle_path(b, h, refs, sites))
    with pytest.raises(Exception):
        e.copy_records_from(a, np.array([b.n], dtype=np.uint32))  # not a record of the source
    assert e.n == b.n
    a.close()
    e.close()


def test_copy_records_tagged_copies_and_split_ids():
    """tag_sr / new_split: the copies arrive as sr-tagged copies of another split, exactly as if the host had staged them so"""
    from elprep_amd import sfm
    cfg, b, h, refs, sites = dataset("tiny", 4000, 12, 0.03)
    k = b.n // 2
    pick = np.arange(k, b.n)[(np.arange(k, b.n) % 5 == 0) & ((b.flag[k:] & 0x904) == 0)]
    a, e, d = Engine(h), Engine(h), Engine(h)
    a.stage(b)
    first = b.take(np.arange(k))
    e.stage(first)
    e.copy_records_from(a, pick, new_split=3, tag_sr=True)
    tagged = sfm.with_sr(b.take(pick), np.ones(pick.size, dtype=bool), split=np.full(pick.size, 3, dtype=np.uint16))
    d.stage(first)
    d.stage(tagged)
    assert e.n == d.n and e.n_sorted == d.n_sorted
    for x, y in zip(_whole_path(e, None, h, refs, sites)[:3], _whole_path(d, None, h, refs, sites)[:3]):
        assert np.array_equal(x, y)
    for eng in (a, e, d):
        eng.close()


def test_copy_records_carries_the_inflated_bam_records():
    """contexts staged from BAM bytes: the records' bytes travel with the columns, the destination emits what the source emits"""
    cfg, b, h, refs, sites = dataset("tiny", 2500, 13, 0.03)
    raw = orc.bam_encode(b, h.rg_ids)
    a, e = Engine(h), Engine(h)
    for eng in (a, e):
        eng.set_read_group_ids(h.rg_ids)
    a.stage_bam(raw)
    for part in np.array_split(np.arange(b.n), 4):
        e.copy_records_from(a, part)
    outs = []
    for eng in (a, e):
        eng.mark_duplicates(True)
        eng.sort_coordinate()
        outs.append(eng.emit_sorted_bam())
    assert outs[0].size > 0 and np.array_equal(outs[0], outs[1])
    f = Engine(h)
    f.stage(b)  # columns only: mixing is refused
    with pytest.raises(Exception):
        f.copy_records_from(a, np.arange(10))
    for eng in (a, e, f):
        eng.close()


def _bam_records(buf):
    """the alignment records of a BAM record stream, one bytes object each (block_size field included)"""
    out, p = [], 0
    while p < buf.size:
        q = p + 4 + int(buf[p:p + 4].view(np.uint32)[0])
        out.append(buf[p:q].tobytes())
        p = q
    return out


def test_emit_merged_bam_is_the_merge_of_the_two_sorted_outputs():
    """elp_emit_merged_bam (MergeSortedFilesSplitPerChromosome with payloads): the group splits (with their sr-tagged copies, which drop
    out) in one context, the spread split in another, both staged from BAM bytes, marked, sorted.  The expectation is built WITHOUT the
    device (round 3 built it from the device's own sorted outputs): the oracle's flags and coordinate order of either batch, the oracle's
    BAM encoder (formatBamAlignment) for the records, and the transliteration of the reference's insertion loop
    (sam/split-merge.go:519-549, tests/test_sfm_cpu.py) for where every spread read goes."""
    from elprep_amd import sfm
    from oracle import simple_filters as sf
    from tests.test_sfm_cpu import _merge_reference
    cfg, b, h, refs, sites = dataset("tiny", 4000, 17, 0.03)
    n_groups, gof = orc.contig_groups(cfg.ref_len, 80000)
    osplit, ospread = sf.split_records(b, gof)
    assert ospread.sum() > 20
    tagged = sfm.with_sr(b, ospread.astype(bool), osplit)
    tagged.split[:] = 0  # (stage_bam gives every record of a call one split id; the tagged copies are recognised by their sr tag)
    sp = b.take(np.nonzero(ospread)[0])
    eg, es = Engine(h), Engine(h)
    recs, keys = [], []
    for eng, batch in ((eg, tagged), (es, sp)):
        eng.set_read_group_ids(h.rg_ids)
        eng.stage_bam(orc.bam_encode(batch, h.rg_ids), split_id=0)
        eng.mark_duplicates(True)
        eng.sort_coordinate()
        oflags = orc.mark_duplicates(batch, h)
        order = orc.sort_coordinate(batch, oflags)[:orc.num_sorted(batch)]
        recs.append(_bam_records(orc.bam_encode(batch, h.rg_ids, order=order, flags=oflags, normalize_tags=True)))
        keys.append([(int(batch.refid[i]), int(batch.pos[i])) for i in order])
        assert len(recs[-1]) == eng.n_sorted
    rg, rs = recs
    # the merge runs over the mapped group reads; the unmapped split (refid -1, at the end of the group context's output) follows it
    n_mapped = sum(1 for k in keys[0] if k[0] >= 0)
    codes = _merge_reference(keys[0][:n_mapped], keys[1])
    want = [rg[c] if c >= 0 else rs[-c - 1] for c in codes] + rg[n_mapped:]
    got = eg.emit_merged_bam(es)
    assert got.tobytes() == b"".join(want)
    # the device's own pieces agree with the same expectation
    assert eg.emit_sorted_bam().tobytes() == b"".join(rg) and es.emit_sorted_bam().tobytes() == b"".join(rs)
    # an empty spread context
    e0 = Engine(h)
    e0.set_read_group_ids(h.rg_ids)
    e0.stage_bam(orc.bam_encode(sp.take(np.arange(0)), h.rg_ids))
    e0.sort_coordinate()
    assert eg.emit_merged_bam(e0).tobytes() == b"".join(rg)
    for eng in (eg, es, e0):
        eng.close()


def test_lut_uploaded_ahead_of_apply():
    """elp_bqsr_lut_upload + elp_bqsr_apply(NULL, NULL): the same bytes as the call that brings its LUT along; without an upload the
    call is refused"""
    cfg, b, h, refs, sites = dataset("tiny", 6000, 19, 0.03)
    outs = []
    for ahead in (False, True):
        e = Engine(h)
        e.stage(b)
        e.mark_duplicates(True)
        for r in range(h.n_ref):
            e.set_reference(r, refs[r])
            e.set_known_sites(r, sites[r])
        qt, ct, xt = e.recalibrate(500)
        lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
        if ahead:
            with pytest.raises(Exception):
                e.apply_bqsr(None, None, 500)
            e.lut_upload(lut, present, 500)
            outs.append(e.apply_bqsr(None, None, 500))
        else:
            outs.append(e.apply_bqsr(lut, present, 500))
        e.close()
    assert np.array_equal(outs[0], outs[1]) and (outs[0] != b.qual).any()
