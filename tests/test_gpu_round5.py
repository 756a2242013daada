"""GPU parity tests (-m gpu) of round 5, every output bit-exact against the CPU oracle through the C ABI:
any number of read groups in BQSR gather and apply (the reference's tables are maps that just grow, filters/bqsr.go:467-551, :936-1005) -
the general count kernel in passes over covariate subsets, the one-length count kernel on exactly-sized per-covariate segments with its
trips shared evenly among the workgroups, ApplyBQSR split by covariate with one row dictionary per covariate."""
import dataclasses

import numpy as np
import pytest

import oracle as orc
from elprep_amd.engine import BqsrTables, Engine
from tests.test_gpu_ragged import _random_case
from tests.test_gpu_round3 import _uniform_case

pytestmark = pytest.mark.gpu

QUALS7 = [2, 5, 6, 12, 23, 37, 41]


def _gather_apply(b, h, refs, sites, tuning=None, max_cycle=500, oracle=None, chunks=2):
    """stage, mark, gather, finalize, apply on the device; -> (tables, qual); compared with `oracle` = (flags, tables, qual) if given"""
    e = Engine(h, tuning=tuning or {})
    cuts = np.linspace(0, b.n, chunks + 1).astype(int)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        if hi > lo:
            e.stage(b.take(np.arange(lo, hi)))
    flags = e.mark_duplicates()
    for r in range(h.n_ref):
        e.set_reference(r, refs[r])
        e.set_known_sites(r, sites[r])
    qt, ct, xt = e.recalibrate(max_cycle)
    lut, present = BqsrTables(qt, ct, xt, max_cycle).finalize().build_lut(0)
    qual = e.apply_bqsr(lut, present, max_cycle)
    e.close()
    if oracle is not None:
        oflags, (oq, oc, ox), oqual = oracle
        assert np.array_equal(flags, oflags), tuning
        assert np.array_equal(ct, oc), ("cycle table", tuning)
        assert np.array_equal(xt, ox) and np.array_equal(qt, oq), ("context / quality table", tuning)
        assert np.array_equal(qual, oqual), ("recalibrated qualities", tuning)
    return (qt, ct, xt), qual


def _oracle(b, h, refs, sites, max_cycle=500):
    oflags = orc.mark_duplicates(b, h)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), oflags, max_cycle)
    assert oq[..., 0].sum() > 0
    oqual = orc.BqsrFinal(oq, oc, ox, max_cycle).apply(b, h, 0)
    return oflags, (oq, oc, ox), oqual


@pytest.mark.parametrize("n_cov", [17, 32, 64])
def test_many_read_groups_ragged_reads(n_cov):
    """ragged read lengths (adapter-trimmed data) with more read groups than one workgroup's LDS holds table rows for: round 4 returned
    ELP_ERR_UNSUPPORTED here; the general count kernel now runs one pass per covariate subset, the general apply
    kernel gathers from the dense LUT"""
    b, h, refs, sites = _random_case(100 + n_cov, 6000, quals=QUALS7, n_cov=n_cov)
    assert h.n_cov == n_cov
    _gather_apply(b, h, refs, sites, oracle=_oracle(b, h, refs, sites))


@pytest.mark.parametrize("n_cov,length", [(17, 150), (32, 151), (64, 100), (130, 76)])
def test_many_read_groups_one_length(n_cov, length):
    """read sets of one length with 17 .. 130 read groups: the one-length count kernel split by covariate (64, 128 or 256 exactly-sized
    segments), ApplyBQSR split by covariate with per-covariate row dictionaries; and the general kernels (forced) on the same reads"""
    b, h, refs, sites = _uniform_case(200 + n_cov, 7000, length, quals=QUALS7, n_cov=n_cov)
    assert h.n_cov == n_cov
    want = _oracle(b, h, refs, sites)
    _gather_apply(b, h, refs, sites, oracle=want)
    if n_cov <= 32:
        _gather_apply(b, h, refs, sites, tuning={"count_kernel": 1, "apply_kernel": 1}, oracle=want, chunks=1)


@pytest.mark.parametrize("n_cov,length,n_q", [(2, 150, 7), (4, 151, 7), (3, 33, 4), (4, 250, 12), (1, 150, 7), (5, 16, 3)])
def test_split_forms_forced_on_few_read_groups(n_cov, length, n_q):
    """the covariate split of both one-length kernels on read sets that would fit one table (what decides is only the LDS): same bytes"""
    quals = [2] + list(range(6, 6 + n_q))
    b, h, refs, sites = _uniform_case(300 + length, 6000, length, quals=quals, n_cov=n_cov)
    want = _oracle(b, h, refs, sites)
    for tune in ({"count_kernel": 3, "apply_kernel": 3}, {}, {"apply_kernel": 3, "count_kernel": 2}):
        _gather_apply(b, h, refs, sites, tuning=tune, oracle=want, chunks=3)


def test_read_groups_of_very_different_sizes():
    """nine reads in ten belong to one read group, three read groups are empty: the split kernels' workgroups share the trips evenly
    whatever the segments' sizes (round 4: a covariate's share of the workgroups was fixed); empty covariates have no tables"""
    b, h, refs, sites = _uniform_case(77, 9000, 150, quals=QUALS7, n_cov=12)
    rng = np.random.default_rng(5)
    rg = b.rgid.copy()
    rg[rng.random(b.n) < 0.9] = 3
    rg[np.isin(rg, [5, 6, 7])] = 8
    b.rgid[:] = rg
    want = _oracle(b, h, refs, sites)
    for tune in ({"count_kernel": 3, "apply_kernel": 3}, {}):
        _gather_apply(b, h, refs, sites, tuning=tune, oracle=want)


def _random_tables(h, quals, max_cycle, length, seed, spread):
    """count tables that did not come from a gather: random observations / mismatches for the given qualities and the cycles of
    `length`-base reads, so that the LUT has many distinct rows"""
    rng = np.random.default_rng(seed)
    nc = 2 * max_cycle + 1
    qt = np.zeros((h.n_cov, 94, 2), np.int64)
    ct = np.zeros((h.n_cov, 94, nc, 2), np.int64)
    xt = np.zeros((h.n_cov, 94, 16, 2), np.int64)
    for c in range(h.n_cov):
        for q in quals:
            obs = rng.integers(20000, 400000, 2 * length + 1)
            mis = (obs * spread * 10.0 ** (-4.0 * rng.random(2 * length + 1))).astype(np.int64)  # error rates over four decades
            ct[c, q, max_cycle - length:max_cycle + length + 1, 0] = obs
            ct[c, q, max_cycle - length:max_cycle + length + 1, 1] = mis
            ct[c, q, max_cycle, :] = 0  # cycle 0 does not exist
            qt[c, q] = ct[c, q].sum(axis=0)
            o = rng.integers(100, 5000, 16)
            xt[c, q, :, 0] = o
            xt[c, q, :, 1] = (o * rng.random(16) * spread).astype(np.int64)
    return qt, ct, xt


@pytest.mark.parametrize("n_q,expect", [(3, "per covariate"), (14, "general")])
def test_apply_with_more_distinct_lut_rows_than_one_dictionary_holds(n_q, expect):
    """a LUT whose rows are (nearly) all distinct: the one dictionary of ApplyBQSR's one-length kernel overflows its one-byte ids, the
    kernel leaves without touching a byte and the host takes one dictionary per covariate - or, if a covariate's rows do not fit
    either, the general kernel; the bytes are the oracle's ApplyBQSR on the same tables each time"""
    quals = [2] + [6 + 3 * k for k in range(n_q)]
    b, h, refs, sites = _uniform_case(55, 5000, 120, quals=quals, n_cov=4)
    qt, ct, xt = _random_tables(h, quals[1:], 500, 120, 9, 0.5)
    want = orc.BqsrFinal(qt, ct, xt, 500).apply(b, h, 0)
    lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
    rows = lut.reshape(h.n_cov, 94, 1001, 17)[:, quals[1:], 500 - 120:500 + 121]
    n_all = len({r.tobytes() for r in rows.reshape(-1, 17)})
    n_per = max(len({r.tobytes() for r in rows[c].reshape(-1, 17)}) for c in range(h.n_cov))
    assert n_all > 249
    assert (n_per <= 249) == (expect == "per covariate"), (n_all, n_per)
    for tune in ({}, {"apply_kernel": 3}, {"apply_kernel": 1}):
        e = Engine(h, tuning=tune)
        e.stage(b)
        got = e.apply_bqsr(lut, present, 500)
        e.close()
        assert np.array_equal(got, want), tune


def test_genome_with_hg38_sized_contigs():
    """the key and window arithmetic at the REAL genome's sizes: a 249 Mbp and a 57 Mbp contig among 24 (POS needs 28
    bits: 34 live bits in the coordinate-sort key = five radix passes; positions above 2^27; reference windows 120 MB into a packed
    contig), reads at the very start of the contigs and hanging over their ends - every output against the oracle"""
    from tools import synth
    ref_len = [248_956_422] + [3000 + 100 * k for k in range(22)] + [57_227_415]
    cfg = synth.SynthConfig(ref_len=ref_len, seed=synth.BASE_SEED + 77)
    h = cfg.header()
    b = synth.generate(cfg, 0, 30000)
    rng = np.random.default_rng(3)
    big = np.nonzero(((b.refid == 0) | (b.refid == 23)) & ((b.flag & 0x4) == 0))[0]
    assert big.size > 40000 and int(b.pos.max()) > (1 << 27)
    ends = rng.choice(big, 600, replace=False)
    rl = np.asarray(ref_len, np.int64)[b.refid[ends]]
    b.pos[ends[:400]] = (rl[:400] - rng.integers(0, 170, 400)).astype(np.int32)   # the last bases of the contig, or over its end
    b.pos[ends[400:]] = rng.integers(1, 20, 200).astype(np.int32)                  # its first bases
    refs = [synth.reference(cfg, r) for r in range(h.n_ref)]
    sites = [orc.flatten(orc.sort_by_start(synth.known_sites_raw(cfg, r))) for r in range(h.n_ref)]
    oflags = orc.mark_duplicates(b, h)
    operm = orc.sort_coordinate(b, oflags)
    _, octr, _ = orc.dup_metrics(b, h, operm, 100)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), oflags, 500)
    oqual = orc.BqsrFinal(oq, oc, ox, 500).apply(b, h, 0)
    for tune in ({}, {"count_kernel": 1, "apply_kernel": 1}):
        e = Engine(h, tuning=tune)
        e.stage(b)
        flags = e.mark_duplicates(True)
        perm = e.sort_coordinate()
        ctr = e.dup_metrics(100)
        for r in range(h.n_ref):
            e.set_reference(r, refs[r])
            e.set_known_sites(r, sites[r])
        qt, ct, xt = e.recalibrate(500)
        lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
        qual = e.apply_bqsr(lut, present, 500)
        e.close()
        assert np.array_equal(flags, oflags) and np.array_equal(perm, operm) and np.array_equal(ctr, octr), tune
        assert np.array_equal(ct, oc) and np.array_equal(xt, ox) and np.array_equal(qt, oq), tune
        assert np.array_equal(qual, oqual), tune


def _two_ranks(h, parts, body):
    """two ranks as two host threads on the one GPU; the group's messages go through Python queues (elp_group_set_p2p)"""
    import queue
    import threading
    chan = {(0, 1): queue.Queue(), (1, 0): queue.Queue()}
    readers = [Engine(h), Engine(h)]
    dests = [Engine(h), Engine(h)]
    errors = [None, None]

    def rank(r):
        def sendrecv(sp, data, rp, nbytes):
            if sp >= 0 and data is not None:
                chan[(r, sp)].put(data)
            return chan[(rp, r)].get(timeout=60) if rp >= 0 and nbytes else None
        try:
            if parts[r].n:
                readers[r].stage(parts[r])
            readers[r].group_init_transport(r, 2, lambda v: None)
            readers[r].group_set_p2p(sendrecv)
            body(r, readers[r], dests[r])
        except Exception as ex:  # noqa: BLE001
            errors[r] = ex
    th = [threading.Thread(target=rank, args=(r,)) for r in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join(180)
    assert not any(t.is_alive() for t in th), "a rank still waits for its peer"
    return readers, dests, errors


def test_exchange_in_several_pieces_and_a_failure_that_reaches_both_ranks():
    """elp_exchange_records moves its records in pieces (here 700 records each: rank 0 sends five pieces and receives two); and when one
    rank's side of a step fails - an index that is no record of its source - BOTH calls return an error and neither rank hangs in a
    message the other will never post (exchange.hip)"""
    from elprep_amd.batch import Batch
    from elprep_amd.engine import ElpError
    from tests.common import dataset
    cfg, b, h, refs, sites = dataset("tiny", 8000, 23, 0.03)
    half = b.n // 2
    parts = [b.take(np.arange(half)), b.take(np.arange(half, b.n))]
    pick = [np.arange(0, 3300), np.arange(100, 1300)]

    def good(r, reader, dest):
        reader.set_tuning("exchange_piece", 700)
        reader.exchange_records(1 - r, pick[r], dest, 1 - r)
    readers, dests, errors = _two_ranks(h, parts, good)
    assert errors == [None, None], errors
    for r in (0, 1):
        want = parts[1 - r].take(pick[1 - r])
        ref = Engine(h)
        ref.stage(want)
        assert dests[r].n == want.n
        assert np.array_equal(dests[r].mark_duplicates(True), ref.mark_duplicates(True))
        assert np.array_equal(dests[r].sort_coordinate(), ref.sort_coordinate())
        ref.close()
    for e in readers + dests:
        e.close()

    def bad(r, reader, dest):
        reader.set_tuning("exchange_piece", 700)
        idx = pick[r].copy()
        if r == 0:
            idx[1500] = parts[0].n + 5  # in the third piece: two pieces arrive, then the failure
        reader.exchange_records(1 - r, idx, dest, 1 - r)
    readers, dests, errors = _two_ranks(h, parts, bad)
    assert isinstance(errors[0], ElpError) and "not records of the source" in str(errors[0]), errors
    assert isinstance(errors[1], ElpError) and "failed on its side" in str(errors[1]), errors
    assert dests[1].n == 1400 and dests[0].n == pick[1].size  # what arrived before the failure stays; rank 1's records all reached rank 0
    for e in readers + dests:
        e.close()


def test_split_phase_through_the_c_abi_between_two_ranks():
    """sfm.route_device - what bench.py --gpus N and SfmRank.route run: every rank stages its input batches into a
    reader context, elp_split_classify on the device, elp_copy_records inside the rank and elp_exchange_records between the ranks (two host
    threads on the one GPU, the group's messages through queues).  Every rank's contexts must hold exactly what SplitFilePerChromosome
    (sam/split-merge.go:280-293) writes into its split files: the group files' records in input order with the sr-tagged copies among
    them and their split ids, the spread file's records with the spread owner - checked against contexts the host staged with the
    same records, and against the oracle's duplicate marking"""
    from elprep_amd import sfm
    from elprep_amd.batch import Batch
    from tests import sfm_worker
    world = 2
    inputs = []
    for r in range(world):
        cfg, gof, G, owner, b = sfm_worker.make_rank_input(r, world, pairs_per_rank=3000)
        cut = b.n // 2
        inputs.append([b.take(np.arange(cut)), b.take(np.arange(cut, b.n))])
    h = cfg.header()
    spread_owner = int(owner[G + 1])
    # what arrives where: per routing round the rank's own records, then the other rank's
    want_local = [[] for _ in range(world)]
    want_spread = []
    for k in range(2):
        for r in range(world):
            for src in (r, 1 - r):
                b = inputs[src][k]
                g, sp = sfm.split_records(b, gof)
                idx = np.nonzero(owner[g] == r)[0]
                if idx.size:
                    want_local[r].append(sfm.with_sr(b, sp, g).take(idx))
                if r == spread_owner and sp.any():
                    want_spread.append(sfm.with_sr(b, np.zeros(b.n, bool), np.zeros(b.n, np.uint16)).take(np.nonzero(sp)[0]))
    got = {}

    def body(r, reader, dest):
        spread_ctx = Engine(h)
        got[r] = spread_ctx
        for k in range(2):
            sfm.route_device(reader, inputs[r][k], gof, G, owner, r, world, dest, spread_ctx)
    readers, dests, errors = _two_ranks(h, [sfm.empty_batch(), sfm.empty_batch()], body)
    assert errors == [None, None], errors
    for r in range(world):
        want = Batch.concat(want_local[r])
        assert dests[r].n == want.n
        ref = Engine(h)
        ref.stage(want)
        oflags = orc.mark_duplicates(want, h)
        assert np.array_equal(dests[r].mark_duplicates(True), oflags) and np.array_equal(ref.mark_duplicates(True), oflags)
        assert np.array_equal(dests[r].sort_coordinate(), ref.sort_coordinate())
        assert dests[r].n_sorted == ref.n_sorted
        assert np.array_equal(dests[r].dup_metrics(100), ref.dup_metrics(100))
        ref.close()
    assert sum(int(Batch.concat(w).has_sr.sum()) for w in want_local) > 30
    wsp = Batch.concat(want_spread)
    assert got[spread_owner].n == wsp.n and wsp.n > 30 and got[1 - spread_owner].n == 0
    assert np.array_equal(got[spread_owner].mark_duplicates(True), orc.mark_duplicates(wsp, h))
    for e in readers + dests + list(got.values()):
        e.close()


def test_merge_phase_across_two_ranks_through_the_c_abi():
    """sfm.emit_merged_device (VERDICT r4 missing #2, the merge half): BAM records in, split phase through the C ABI (the inflated records
    travel with the columns), mark duplicates + sort of every rank's contexts, then the spread owner sends every rank the spread reads of
    its contig groups and every rank emits the merge of its groups' output with them (elp_emit_merged_bam).  Expectation without the
    device: the oracle's flags and order per context, the oracle's BAM encoder, the transliteration of the reference's insertion loop
    (sam/split-merge.go:519-549)"""
    from elprep_amd import sfm
    from elprep_amd.batch import Batch
    from tests import sfm_worker
    from tests.test_gpu_round3 import _bam_records
    from tests.test_sfm_cpu import _merge_reference
    world = 2
    inputs = []
    for r in range(world):
        cfg, gof, G, owner, b = sfm_worker.make_rank_input(r, world, pairs_per_rank=2500)
        inputs.append(b)
    h = cfg.header()
    spread_owner = int(owner[G + 1])
    # what arrives where (as in the split-phase test), and for every arriving record the input batch and index it came from: the oracle's
    # BAM encoder derives a record's optional fields from its index in the batch it encodes, so the expected records are encoded from
    # the INPUT batches
    want_local, want_spread = [[] for _ in range(world)], []
    from_local, from_spread = [[] for _ in range(world)], []
    for r in range(world):
        for src in (r, 1 - r):
            b = inputs[src]
            g, sp = sfm.split_records(b, gof)
            idx = np.nonzero(owner[g] == r)[0]
            if idx.size:
                want_local[r].append(sfm.with_sr(b, sp, g).take(idx))
                from_local[r] += [(src, int(i)) for i in idx]
            if r == spread_owner and sp.any():
                want_spread.append(b.take(np.nonzero(sp)[0]))
                from_spread += [(src, int(i)) for i in np.nonzero(sp)[0]]

    def encoded(origin, flags):
        """the records of a context (arrival order; origin[j] = (input batch, index), flags[j] = the oracle's FLAG) as elprep writes them"""
        recs = [None] * len(origin)
        for src in range(world):
            js = [j for j, (s_, _) in enumerate(origin) if s_ == src]
            if not js:
                continue
            fl = inputs[src].flag.copy()
            idx = np.asarray([origin[j][1] for j in js], dtype=np.uint32)
            fl[idx] = flags[js]
            for j, rec in zip(js, _bam_records(orc.bam_encode(inputs[src], h.rg_ids, order=idx, flags=fl, normalize_tags=True))):
                recs[j] = rec
        return recs
    out = {}

    def body(r, reader, dest):
        spread_ctx, part = Engine(h), Engine(h)
        for e in (reader, dest, spread_ctx, part):
            e.set_read_group_ids(h.rg_ids)
        part.group_share(reader)
        spread_ctx.group_share(reader)
        sfm.route_device(reader, inputs[r], gof, G, owner, r, world, dest, spread_ctx, stage=lambda e, x: e.stage_bam(orc.bam_encode(x, h.rg_ids)))
        for e in (dest, spread_ctx):
            e.mark_duplicates(True, fetch=False)
            e.sort_coordinate(fetch=False)
        out[r] = sfm.emit_merged_device(dest, spread_ctx, part, gof, G, owner, r, world).tobytes()
        spread_ctx.close()
        part.close()
    readers, dests, errors = _two_ranks(h, [sfm.empty_batch(), sfm.empty_batch()], body)
    assert errors == [None, None], errors
    # the spread file: the oracle's flags, order and records
    wsp = Batch.concat(want_spread)
    sflags = orc.mark_duplicates(wsp, h)
    sorder = orc.sort_coordinate(wsp, sflags)[:orc.num_sorted(wsp)]
    s_all = encoded(from_spread, sflags)
    srecs = [s_all[j] for j in sorder]
    sgroup = gof[wsp.refid[sorder]]
    for r in range(world):
        loc = Batch.concat(want_local[r])
        lflags = orc.mark_duplicates(loc, h)
        lorder = orc.sort_coordinate(loc, lflags)[:orc.num_sorted(loc)]
        l_all = encoded(from_local[r], lflags)
        lrecs = [l_all[j] for j in lorder]
        keys = [(int(loc.refid[i]), int(loc.pos[i])) for i in lorder]
        n_mapped = sum(1 for k in keys if k[0] >= 0)
        mine = np.nonzero(owner[sgroup] == r)[0]  # the spread reads of this rank's contig groups, in the spread file's order
        skeys = [(int(wsp.refid[sorder[j]]), int(wsp.pos[sorder[j]])) for j in mine]
        codes = _merge_reference(keys[:n_mapped], skeys)
        want = [lrecs[c] if c >= 0 else srecs[mine[-c - 1]] for c in codes] + lrecs[n_mapped:]
        assert len(mine) > 10 or r != spread_owner
        if out[r] != b"".join(want):  # say what differs: a missing part, an order, or a record
            got = _bam_records(np.frombuffer(out[r], dtype=np.uint8))
            first = next((k for k, (a, c) in enumerate(zip(got, want)) if a != c), min(len(got), len(want)))
            def who(rec):
                rid, pos = np.frombuffer(rec[4:12], dtype=np.int32)
                return (int(rid), int(pos) + 1, rec[36:36 + rec[12] - 1].decode(), "spread" if rec in set(srecs) else ("group" if rec in set(lrecs) else "neither"))
            raise AssertionError((r, "records", len(got), len(want), "same multiset", sorted(got) == sorted(want), "first difference at", first,
                                  [who(x) for x in got[first:first + 3]], [who(x) for x in want[first:first + 3]], "spread reads of the rank", len(mine)))
    for e in readers + dests:
        e.close()


@pytest.mark.parametrize("n_pairs", [3000, 70000])
def test_emit_sorted_bgzf_compresses_on_the_device(n_pairs):
    """elp_emit_sorted_bgzf compresses (VERDICT r4 missing #1): every member inflates (zlib) to its part of elp_emit_sorted_bam's record
    stream, CRC-32 and ISIZE hold, the file is well below the stored form's size; the stored form (tuning) still frames the same stream;
    the compressed blocks come back through the device's own inflate (elp_stage_bgzf) as the same records.  70 000 pairs: more blocks
    than workgroups (the compressor's workgroups loop over the blocks)"""
    from tests.test_gpu_round4 import _bam_case, _members
    b, h, raw, rec_off = _bam_case(n_pairs=n_pairs, seed=4)
    e = Engine(h)
    e.set_read_group_ids(h.rg_ids)
    e.stage_bam(raw, rec_off=rec_off)
    e.mark_duplicates(True)
    e.sort_coordinate()
    want = e.emit_sorted_bam().tobytes()
    bz = e.emit_sorted_bgzf().tobytes()
    e.set_tuning("bgzf_fixed", 1)
    bz1 = e.emit_sorted_bgzf().tobytes()  # fixed Huffman codes only (round 5's form)
    e.set_tuning("bgzf_fixed", 0)
    e.set_tuning("bgzf_stored", 1)
    bz0 = e.emit_sorted_bgzf().tobytes()
    e.close()
    assert len(bz) < 0.93 * len(bz1), (len(bz), len(bz1))  # the blocks' own codes (round 6) against the fixed ones
    for blob in (bz, bz1, bz0):
        mem = _members(blob)
        assert b"".join(m for _, m in mem) == want
        assert all(size <= 65536 and 0 < len(m) <= 65280 for size, m in mem)
        assert [len(m) for _, m in mem[:-1]] == [65280] * (len(mem) - 1)
    assert len(bz) < 0.62 * len(bz0), (len(bz), len(bz0))
    if n_pairs > 50000:
        assert len(_members(bz)) > 300
    # back in through the device's inflate: the same records in the same (sorted) order
    e2, e3 = Engine(h), Engine(h)
    for eng, src in ((e2, np.frombuffer(bz, dtype=np.uint8)), (e3, np.frombuffer(bz0, dtype=np.uint8))):
        eng.set_read_group_ids(h.rg_ids)
        eng.stage_bgzf(src)
    assert e2.n == e3.n > 0
    f2, f3 = e2.mark_duplicates(True), e3.mark_duplicates(True)
    assert np.array_equal(f2, f3)
    assert np.array_equal(e2.sort_coordinate(), e3.sort_coordinate())
    assert e2.emit_sorted_bam().tobytes() == e3.emit_sorted_bam().tobytes()
    e2.close()
    e3.close()


@pytest.mark.parametrize("n_cov,length", [(3, 150), (20, 100)])
def test_tables_and_lut_in_rows_form_on_the_device(n_cov, length):
    """the rows form of the host's table path (round 5): elp_bqsr_tables_fetch_rows gives the dense tables' rows of the counted qualities,
    the LUT uploaded as rows + defaults (elp_bqsr_lut_upload_rows, expanded on the device) recalibrates to the same bytes as the dense LUT
    - the oracle's; tables that hold another quality's rows (summed with another context's) are reported, not truncated"""
    b, h, refs, sites = _uniform_case(400 + n_cov, 6000, length, quals=QUALS7, n_cov=n_cov)
    want = _oracle(b, h, refs, sites)
    e = Engine(h)
    e.stage(b)
    e.mark_duplicates()
    for r in range(h.n_ref):
        e.set_reference(r, refs[r])
        e.set_known_sites(r, sites[r])
    e.recalibrate_device(500)
    qt, ct, xt = e.tables_fetch()
    quals = e.quals_counted()
    assert quals == [q for q in QUALS7 if q >= 6]
    qr, cr, xr = e.tables_fetch_rows(quals)
    assert np.array_equal(qr, qt[:, quals]) and np.array_equal(cr, ct[:, quals]) and np.array_equal(xr, xt[:, quals])
    assert not np.delete(qt, quals, axis=1).any() and np.array_equal(ct, want[1][1])
    assert e.tables_fetch_rows(quals[1:]) is None            # a counted quality that was not asked for: reported
    tb = BqsrTables.from_rows(h.n_cov, quals, qr, cr, xr, 500).finalize()
    e.lut_upload_rows(quals, *tb.build_lut_rows(quals, 0), 500)
    assert np.array_equal(e.apply_bqsr(None, None, 500), want[2])
    e.close()


@pytest.mark.gpu
def test_small_read_set_behind_a_large_one_in_one_context():
    """A context's scratch buffers only grow.  Mark duplicates keeps its candidate codes and its pair list in two of them across the
    radix passes of the big-group pairing, the partitioned mate pass and the pair partition; until round 5 the radix sort's digit
    histograms and the scan's partial sums lived in the SAME two slots: a five-record read set staged behind a large one (slots larger
    than the small call asks for: no reallocation, the histograms land on the live codes) came out with the wrong pairs."""
    from tests import kat_cases
    from tests.kat_cases import _rec, batch_from_records
    h = kat_cases.header2()
    rng = np.random.default_rng(11)
    big = []
    for k in range(6000):  # pairs all over the two contigs, a few of them duplicates of each other
        p, q = int(rng.integers(1, 400)), int(rng.integers(500, 900))
        big += [_rec("b%d" % k, 99, k & 1, p, k & 1, q, q - p + 10), _rec("b%d" % k, 147, k & 1, q, k & 1, p, -(q - p + 10))]
    bb = batch_from_records(big)
    for path in (0, 1, 2):
        e = Engine(h, tuning={"mate_path": path})
        for k, (b, want) in enumerate(kat_cases.toggling_cases()):
            e.reset()
            e.stage(bb)
            assert np.array_equal(e.mark_duplicates(True), orc.mark_duplicates(bb, h)), (path, k)
            e.reset()
            e.stage(b)
            flags = e.mark_duplicates(True)
            assert np.nonzero(flags & 0x400)[0].tolist() == want, (path, k)
            # (the sort is the pipeline's Finalize behind the filter: CoordinateLess's modFlag tie-break sees the duplicate bits)
            assert np.array_equal(e.sort_coordinate(), orc.sort_coordinate(dataclasses.replace(b, flag=flags.astype(b.flag.dtype)))), (path, k)
        e.close()
