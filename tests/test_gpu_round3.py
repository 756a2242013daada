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
