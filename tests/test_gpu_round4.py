"""GPU parity tests (-m gpu) added in round 4: the pair phase of mark duplicates as a hash partition + one LDS table per bucket
(elprep_amd/csrc/markdup.hip: k_pair_list, k_pair_bounds, k_pair_bucket) on its three paths - table, entries beyond the remembered
slots, table overflow (groups peeled off one by one) - against the oracle's restatement of filters/mark-duplicates.go:329-396."""
import numpy as np
import pytest

import oracle as orc
from elprep_amd.batch import Header
from elprep_amd.engine import Engine
from tests.common import dataset
from tests.test_gpu_round2 import _pileup

pytestmark = pytest.mark.gpu


def _flags_and_metrics(b, h, tuning=None):
    e = Engine(h, tuning=tuning)
    e.stage(b)
    flags = e.mark_duplicates(True)
    ctr, hist = e.dup_metrics(100, hist_len=16)
    e.close()
    return flags, ctr, hist


@pytest.mark.parametrize("slots", [2, 16, 1024])
def test_pair_buckets_table_sizes_and_overflow_path(slots):
    """the same reads with LDS tables of 1024 slots (the default), 16 slots (most buckets overflow) and 2 slots (every bucket with more
    than two keys takes the peeling path): flags, counters and set-size histograms are the oracle's every time"""
    cfg, b, h, refs, sites = dataset("tiny", 30000, 5, 0.03)
    oflags, octr, ohist = orc.dup_metrics(b, h, None, 100, hist_len=16)
    flags, ctr, hist = _flags_and_metrics(b, h, {"pair_table_slots": slots})
    assert np.array_equal(flags, oflags)
    assert np.array_equal(ctr, octr) and np.array_equal(hist, ohist)
    assert int(((flags & 0x400) != 0).sum()) > 1000


@pytest.mark.parametrize("slots", [2, 1024])
def test_pair_tournament_on_score_ties(slots):
    """5 500 pairs on ONE pair key, all with the same score sum: the QNAME decides (filters/mark-duplicates.go:383-389); the bucket is
    larger than the remembered slots, so the later phases look their table slot up again"""
    rng = np.random.default_rng(11)
    b = _pileup(3000, 2500, rng)
    b.qual[:] = 30  # no best pair by score
    h = Header(ref_len=np.array([5000], np.int32), rg_lib=np.array([0, 0], np.uint16), rg_cov=np.array([0, 1], np.uint16))
    oflags, octr, ohist = orc.dup_metrics(b, h, None, 100, hist_len=16)
    flags, ctr, hist = _flags_and_metrics(b, h, {"pair_table_slots": slots})
    assert np.array_equal(flags, oflags)
    assert int(((flags & 0x400) == 0).sum()) == 2
    assert np.array_equal(ctr, octr) and np.array_equal(hist, ohist)


def test_pairs_of_several_libraries_and_splits_share_positions():
    """pairs at the same two ends but in different libraries / split files are different keys (the key carries LIBID of the first end and
    the split id): the pileup's read groups are given two libraries, half of the records a second split id"""
    rng = np.random.default_rng(12)
    b = _pileup(600, 500, rng)
    b.split = np.zeros(b.n, np.uint16)
    b.split[(np.arange(b.n) // 2) % 2 == 1] = 1  # both mates of a pair share the split
    h = Header(ref_len=np.array([5000], np.int32), rg_lib=np.array([0, 1], np.uint16), rg_cov=np.array([0, 1], np.uint16))
    e = Engine(h)
    e.stage(b)
    flags = e.mark_duplicates(True)
    e.close()
    # the oracle runs split file by split file
    want = np.zeros(b.n, np.uint16)
    for sp in (0, 1):
        idx = np.nonzero(b.split == sp)[0]
        sub = b.take(idx)
        sub.split[:] = 0
        want[idx] = orc.mark_duplicates(sub, h)
    assert np.array_equal(flags, want)
    assert int(((flags & 0x400) == 0).sum()) == 2 * 4  # one surviving pair per (library, split)


def test_set_tuning_rejects_unknown_keys_and_bad_values():
    from elprep_amd.engine import ElpError
    h = Header(ref_len=np.array([5000], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))
    e = Engine(h)
    with pytest.raises(ElpError, match="unknown key"):
        e.set_tuning("no_such_key", 1)
    with pytest.raises(ElpError, match="power of two"):
        e.set_tuning("pair_table_slots", 3)
    e.set_tuning("pair_table_slots", 64)
    e.close()


@pytest.mark.parametrize("n_q", [23, 24])
def test_one_length_quality_hint_incomplete_at_the_lds_boundary(monkeypatch, n_q):
    """150-base reads, 4 covariates: the one-length count kernel holds 104 LDS rows = 4 x (23 qualities + 3).  A hint that misses one of
    23 qualities is retried on the same kernel with the last row count that fits; a hint that misses one of 24 fitted, the exact set does
    not: the gather runs the prologues again and the general count kernel (round 3 returned ELP_ERR_UNSUPPORTED there; the reference
    just runs, filters/bqsr.go:467-551)"""
    from tests.test_gpu_round3 import _check_gather_apply, _uniform_case
    quals = [2] + list(range(6, 6 + n_q))
    b, h, refs, sites = _uniform_case(21, 6000, 150, quals=quals, n_cov=4)
    assert h.n_cov == 4 and len(set(b.qual[b.qual >= 6].tolist())) == n_q
    monkeypatch.setenv("ELP_TUNE", "qual_hint_drop=17")
    _check_gather_apply(b, h, refs, sites)


@pytest.mark.parametrize("length,seed", [(150, 31), (100, 32), (37, 33), (250, 34), (19, 35)])
def test_read_per_lane_score_kernel_against_the_flat_one_and_the_oracle(length, seed):
    """k_score_uniform (a read per lane over an LDS tile, one-length read sets) and k_score_flat (tuning score_kernel = 1) on the same
    reads - low-quality tails at both ends, reads without any quality > 2, non-candidates: the Phred sums are the oracle's, and the
    low-quality-tail bounds (visible through the BQSR tables and the recalibrated qualities) agree between the kernels and with the oracle"""
    from tests.test_gpu_round3 import _uniform_case
    from elprep_amd.engine import BqsrTables
    b, h, refs, sites = _uniform_case(seed, 5000, length, quals=[0, 1, 2, 3, 14, 15, 16, 30, 41, 93])
    rng = np.random.default_rng(seed)
    qual = b.qual.reshape(b.n, length).copy()
    k = rng.integers(0, length + 1, b.n)
    for r in range(0, b.n, 3):  # a tail of qualities <= 2 at the end, at the start, or the whole read
        if r % 9 == 0:
            qual[r, :] = rng.integers(0, 3, length)
        elif r % 2:
            qual[r, int(k[r]):] = 2
        else:
            qual[r, :int(k[r])] = 1
    b.qual = np.ascontiguousarray(qual.reshape(-1))
    oflags, oupos, oscore = orc.mark_duplicates(b, h, with_adapted=True)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), oflags, 500)
    out = []
    for kern in (0, 1):
        e = Engine(h, tuning={"score_kernel": kern})
        e.stage(b)
        up, sc = e.adapted()
        assert np.array_equal(up, oupos) and np.array_equal(sc, oscore)
        assert np.array_equal(e.mark_duplicates(), oflags)
        for r in range(h.n_ref):
            e.set_reference(r, refs[r])
            e.set_known_sites(r, sites[r])
        qt, ct, xt = e.recalibrate(500)
        assert np.array_equal(qt, oq) and np.array_equal(ct, oc) and np.array_equal(xt, ox)
        lut, present = BqsrTables(qt, ct, xt, 500).finalize().build_lut(0)
        out.append(e.apply_bqsr(lut, present, 500))
        e.close()
    assert np.array_equal(out[0], out[1])
    assert np.array_equal(out[0], orc.BqsrFinal(oq, oc, ox, 500).apply(b, h, 0))


# ---- BGZF on the device (elprep_amd/csrc/bgzf.hip; utils/bgzf/bgzf-files.go)
def _bam_case(n_pairs=3000, seed=2):
    from tools import synth
    cfg, b, h, refs, sites = dataset("tiny", n_pairs, seed, 0.02)
    raw, rec_off = synth.bam_records(b, h.rg_ids)
    return b, h, raw, rec_off


def _members(bz: bytes):
    """the gzip members of a BGZF stream: (total size from the BC field, inflated bytes) - zlib checks every CRC-32 and ISIZE"""
    import struct
    import zlib
    out, p = [], 0
    while p < len(bz):
        assert bz[p:p + 4] == b"\x1f\x8b\x08\x04" and bz[p + 12:p + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", bz, p + 16)[0] + 1
        out.append((bsize, zlib.decompress(bz[p:p + bsize], wbits=31)))
        p += bsize
    return out


def test_emit_sorted_bgzf_inflates_to_the_record_stream():
    b, h, raw, rec_off = _bam_case()
    e = Engine(h)
    e.set_read_group_ids(h.rg_ids)
    e.stage_bam(raw, rec_off=rec_off)
    e.mark_duplicates(True)
    e.sort_coordinate()
    want = e.emit_sorted_bam().tobytes()
    bz = e.emit_sorted_bgzf().tobytes()
    e.close()
    mem = _members(bz)
    assert b"".join(m for _, m in mem) == want
    assert all(size <= 65536 and 0 < len(m) <= 65280 for size, m in mem)
    assert [len(m) for _, m in mem[:-1]] == [65280] * (len(mem) - 1) and len(mem) >= 3


def _bgzf(stream: bytes, level: int, strategy: int = 0, cut: int = 65280, mem_level: int = 8) -> bytes:
    import struct
    import zlib
    out = []
    for k in range(0, len(stream), cut):
        part = stream[k:k + cut]
        co = zlib.compressobj(level, zlib.DEFLATED, -15, mem_level, strategy)
        data = co.compress(part) + co.flush()
        out.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(data) + 25) + data +
                   struct.pack("<II", zlib.crc32(part), len(part)))
    out.append(bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"))  # the end-of-file block
    return b"".join(out)


@pytest.mark.parametrize("level,strategy,cut,tuning", [
    (6, 0, 65280, None),                                         # dynamic Huffman blocks, records span the BGZF blocks
    (1, 0, 30011, None),
    (0, 0, 65280, None),                                         # stored
    (6, 4, 65280, None),                                         # Z_FIXED: fixed Huffman codes
    (6, 0, 65280, {"bgzf_piece": 200_000}),                     # many device passes: a record pending at the end of each
    (6, 0, 4099, {"bgzf_weak_guess": 1, "bgzf_piece": 1 << 20}),  # every guess wrong: the repair pass finds the same starts
    # round 6 (the decoder in two phases: tokens, then matches + CRC):
    (6, 0, 65280, {"bgzf_inflate_piece": 150_000, "bgzf_piece": 70_000}),  # several inflate pieces, scan pieces inside each
    (9, 0, 65280, {"mem_level": 1}),                             # many small DEFLATE blocks per member: a header every few hundred symbols
    (6, 2, 65280, None),                                         # Z_HUFFMAN_ONLY: literals only, codes longer than the tables' bits
    (6, 3, 65280, None),                                         # Z_RLE: matches at distance 1 (a match that overlaps itself)
    (6, 1, 2111, {"mem_level": 3}),                              # Z_FILTERED, small members
    (1, 0, 30011, {"bgzf_inflate": 1}),                          # round 5's one-kernel decoder stays available
    (6, 0, 4099, {"bgzf_tok_fail_above": 37}),                   # the token scratch "does not fit": the decoder's launches halve
])
def test_stage_bgzf_gives_the_records_of_stage_bam(level, strategy, cut, tuning):
    import numpy as _np
    b, h, raw, rec_off = _bam_case()
    header = bytes(_np.random.default_rng(1).integers(0, 256, 1234, dtype=_np.uint8))  # stands for magic + header text + dictionary
    tuning = dict(tuning or {})
    mem_level = tuning.pop("mem_level", 8)
    bz = _np.frombuffer(_bgzf(header + raw.tobytes(), level, strategy, cut, mem_level), dtype=_np.uint8)
    outs = []
    for how in ("bam", "bgzf"):
        e = Engine(h, tuning=tuning)
        e.set_read_group_ids(h.rg_ids)
        if how == "bam":
            e.stage_bam(raw, rec_off=rec_off)
        else:
            e.stage_bgzf(bz, first_record=len(header))
        assert e.n == b.n
        flags = e.mark_duplicates(True)
        perm = e.sort_coordinate()
        outs.append((flags, perm, e.emit_sorted_bam().tobytes()))
        e.close()
    assert _np.array_equal(outs[0][0], outs[1][0]) and _np.array_equal(outs[0][1], outs[1][1]) and outs[0][2] == outs[1][2]
    assert _np.array_equal(outs[0][0], orc.mark_duplicates(b, h))


def test_stage_bgzf_reports_corrupt_blocks():
    import numpy as _np
    from elprep_amd.engine import ElpError
    b, h, raw, rec_off = _bam_case(600)
    good = bytearray(_bgzf(raw.tobytes(), 6))
    for what, at, msg in (("crc", None, "invalid CRC-32"), ("data", 40, "does not inflate|invalid CRC-32")):
        bad = bytearray(good)
        if what == "crc":
            import struct
            bsize = struct.unpack_from("<H", bad, 16)[0] + 1
            bad[bsize - 8] ^= 0x55
        else:
            bad[at] ^= 0xFF
        e = Engine(h)
        e.set_read_group_ids(h.rg_ids)
        with pytest.raises(ElpError, match=msg):
            e.stage_bgzf(_np.frombuffer(bytes(bad), dtype=_np.uint8))
        e.close()
    e = Engine(h)
    e.set_read_group_ids(h.rg_ids)
    with pytest.raises(ElpError, match="ends inside an alignment record"):
        e.stage_bgzf(_np.frombuffer(_bgzf(raw.tobytes()[:-7], 6), dtype=_np.uint8))
    e.close()


# ---- the split phase between processes through the C ABI (elp_exchange_records), over a caller's point-to-point transport
def test_exchange_records_between_two_ranks_through_a_transport():
    """two ranks as two host threads on the one GPU, each with a reader context (its half of the reads) and a destination context; the
    group's messages go through a Python queue (elp_group_set_p2p).  Rank 0 sends its records at odd indices (as sr-tagged copies of split
    5), rank 1 sends its records on contig 0; both receive what the other sent: the destination contexts behave exactly like contexts the
    host staged with the same records - every output of mark duplicates / sort against them.  Then a step in which only one side sends."""
    import queue
    import threading
    from elprep_amd import sfm
    cfg, b, h, refs, sites = dataset("tiny", 6000, 21, 0.03)
    half = b.n // 2
    parts = [b.take(np.arange(half)), b.take(np.arange(half, b.n))]
    pick = [np.arange(1, parts[0].n, 2), np.nonzero(parts[1].refid == 0)[0]]
    chan = {(0, 1): queue.Queue(), (1, 0): queue.Queue()}
    readers = [Engine(h), Engine(h)]
    dests = [Engine(h), Engine(h)]
    errors = []

    def rank(r):
        try:
            other = 1 - r

            def sendrecv(sp, data, rp, nbytes):
                if sp >= 0 and data is not None:
                    chan[(r, sp)].put(data)
                return chan[(rp, r)].get(timeout=60) if rp >= 0 and nbytes else None
            readers[r].stage(parts[r])
            readers[r].group_init_transport(r, 2, lambda v: None)
            readers[r].group_set_p2p(sendrecv)
            if r == 0:
                readers[0].exchange_records(other, pick[0], dests[0], other, new_split=5, tag_sr=True)
                readers[0].exchange_records(-1, None, dests[0], other)          # second step: rank 0 only receives
            else:
                readers[1].exchange_records(other, pick[1], dests[1], other)
                readers[1].exchange_records(other, pick[1][:7], None, -1)       # ... rank 1 only sends
        except Exception as ex:  # noqa: BLE001
            errors.append((r, repr(ex)))
    th = [threading.Thread(target=rank, args=(r,)) for r in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errors, errors
    # rank 0 got rank 1's contig-0 records (twice the first seven), rank 1 got rank 0's odd records as tagged copies of split 5
    from elprep_amd.batch import Batch
    want0 = Batch.concat([parts[1].take(pick[1]), parts[1].take(pick[1][:7])])
    want1 = sfm.with_sr(parts[0].take(pick[0]), np.ones(pick[0].size, dtype=bool), split=np.full(pick[0].size, 5, dtype=np.uint16))
    for got, want in ((dests[0], want0), (dests[1], want1)):
        ref = Engine(h)
        ref.stage(want)
        assert got.n == want.n and got.n_sorted == ref.n_sorted
        assert np.array_equal(got.mark_duplicates(True), ref.mark_duplicates(True))
        assert np.array_equal(got.sort_coordinate(), ref.sort_coordinate())
        assert np.array_equal(got.dup_metrics(100), ref.dup_metrics(100))
        ref.close()
    assert np.array_equal(dests[1].mark_duplicates(True), orc.mark_duplicates(want1, h))
    for e in readers + dests:
        e.close()


@pytest.mark.parametrize("n_cov,n_q,length", [(4, 40, 150), (16, 7, 150), (3, 30, 101), (8, 12, 250)])
def test_count_split_by_covariate(n_cov, n_q, length):
    """read sets whose private count tables do not fit one workgroup's LDS with the rows of every covariate (40 qualities x 4 read groups,
    7 x 16, ...): the one-length count kernel runs split by covariate - records in per-covariate segments, the other region sorted by
    covariate, a workgroup's table holds ONE covariate's rows - and gives the oracle's tables; so does every forced form"""
    from tests.test_gpu_round3 import _uniform_case
    from elprep_amd.engine import BqsrTables
    quals = [2] + list(range(6, 6 + n_q))
    b, h, refs, sites = _uniform_case(40 + n_cov, 6000, length, quals=quals, n_cov=n_cov)
    assert h.n_cov == n_cov
    oflags = orc.mark_duplicates(b, h)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), oflags, 500)
    for force in (0, 3, 1):
        e = Engine(h, tuning={"count_kernel": force})
        e.stage(b)
        e.mark_duplicates(True)
        for r in range(h.n_ref):
            e.set_reference(r, refs[r])
            e.set_known_sites(r, sites[r])
        qt, ct, xt = e.recalibrate(500)
        assert np.array_equal(ct, oc), ("cycle table", force)
        assert np.array_equal(xt, ox) and np.array_equal(qt, oq), force
        e.close()


@pytest.mark.parametrize("mate_path", [0, 1, 2])
def test_mates_by_every_path(mate_path):
    """the neighbour shortcut (0, aligner order), the partitioned pass (1: hash partition + LDS tables, what shuffled input takes) and the
    table in HBM (2) pair the same records: aligner order, shuffled order, and the DeleteOrStore toggling cases with three and four records
    per QNAME (the arrival-order pairing of big groups) - flags, counters and set-size histograms against the oracle"""
    from tests import kat_cases
    from elprep_amd.batch import Batch
    tune = {"mate_path": mate_path}
    cfg, b, h, refs, sites = dataset("tiny", 20000, 8, 0.03)
    rng = np.random.default_rng(4)
    for batch in (b, b.take(rng.permutation(b.n))):
        oflags, octr, ohist = orc.dup_metrics(batch, h, None, 100, hist_len=16)
        flags, ctr, hist = _flags_and_metrics(batch, h, tune)
        assert np.array_equal(flags, oflags)
        assert np.array_equal(ctr, octr) and np.array_equal(hist, ohist)
    h2 = kat_cases.header2()
    for k, (tb, want) in enumerate(kat_cases.toggling_cases()):
        e = Engine(h2, tuning=tune)
        e.stage(tb)
        flags = e.mark_duplicates(True)
        assert np.nonzero(flags & 0x400)[0].tolist() == want, k
        e.close()
    # a third primary record for 300 QNAMEs, anywhere in a shuffled batch
    cand = np.nonzero(((b.flag & 0x904) == 0) & ((b.flag & 0x9) == 0x1))[0]
    ext = b.take(rng.choice(cand, 300, replace=False))
    ext.pos[:] = np.maximum(1, ext.pos + rng.integers(-40, 40, ext.n)).astype(np.int32)
    pb = Batch.concat([b, ext])
    pb = pb.take(rng.permutation(pb.n))
    e = Engine(h, tuning=tune)
    e.stage(pb)
    assert np.array_equal(e.mark_duplicates(True), orc.mark_duplicates(pb, h))
    e.close()


def test_clean_sam_against_the_oracle():
    """elp_clean_sam: the contigs' LN is cut so that a few hundred alignments end behind their reference sequence; MAPQ and CIGAR of every
    record (read back through the BAM encoder) and everything downstream of the rewritten CIGARs (unclipped positions -> duplicate flags)
    equal the oracle's restatement of CleanSam / softClipEndOfRead - with that function's arithmetic as the reference has it"""
    from oracle import simple_filters as sf
    from elprep_amd.batch import Header
    cfg, b, h, refs, sites = dataset("tiny", 8000, 9, 0.03)
    cut = np.array([41000, 30000, 22000], np.int32)
    keep = np.nonzero((b.refid < 0) | (b.pos <= cut[np.clip(b.refid, 0, None)] - 140))[0]   # (no alignment starts near / behind the new end:
    over = np.nonzero((b.refid >= 0) & (b.pos > cut[np.clip(b.refid, 0, None)] - 140) & (b.pos <= cut[np.clip(b.refid, 0, None)] - 20))[0]  # ... but some hang over it)
    sel = np.sort(np.concatenate([keep, over]))
    bb = b.take(sel)
    h2 = Header(ref_len=cut, rg_lib=h.rg_lib, rg_cov=h.rg_cov, ref_names=h.ref_names, rg_ids=h.rg_ids, lib_names=h.lib_names, cov_names=h.cov_names)
    want, n_changed = sf.clean_sam(bb, cut)
    assert n_changed > 30
    e = Engine(h2)
    e.set_read_group_ids(h2.rg_ids)
    e.stage_bam(orc.bam_encode(bb, h2.rg_ids))
    assert e.clean_sam() == n_changed
    assert e.clean_sam() == 0  # (the clipped alignments now end at the reference's end)
    e.close()
    e = Engine(h2)
    e.set_read_group_ids(h2.rg_ids)
    e.stage_bam(orc.bam_encode(bb, h2.rg_ids))
    e.clean_sam()
    flags = e.mark_duplicates(True)
    oflags = orc.mark_duplicates(want, h2)
    assert np.array_equal(flags, oflags)
    perm = e.sort_coordinate()
    operm = orc.sort_coordinate(want, oflags)
    assert np.array_equal(perm, operm)
    got = e.emit_sorted_bam().tobytes()
    assert got == orc.bam_encode(want, h2.rg_ids, order=operm[:orc.num_sorted(want)], flags=oflags, normalize_tags=True).tobytes()
    e.close()
    # columns staged without BAM bytes take the same path
    e = Engine(h2)
    e.stage(bb)
    assert e.clean_sam() == n_changed
    assert np.array_equal(e.mark_duplicates(True), oflags)
    e.close()


@pytest.mark.parametrize("tie_rounds", [0, 1])
@pytest.mark.parametrize("big_group", [False, True])
def test_sort_long_runs_one_key_then_compare(big_group, tie_rounds):
    """Long runs of equal coordinates (pile-ups, the unmapped block) whose names do not fit one 64-bit key of position ranks: the sort
    takes ONE radix round on the run id and the leading live positions and settles the groups of equal keys by comparing the whole
    comparator strings (k_large_ties) - mates with one name, names that differ only far behind the key, three runs that must stay apart.
    With a group of more than 1024 names that agree in everything the key holds, it falls back to the rounds over all positions
    (what "tie_rounds" = 1 forces).  Same permutation as the oracle's stable sort (sam/sam-types.go:425-473, :639-641) every time."""
    from elprep_amd.batch import batch_from_records
    rng = np.random.default_rng(11 + (1 if big_group else 0))
    recs = []
    def rec(name, refid, pos, k):
        paired = bool(rng.integers(0, 2))
        return dict(qname=name, flag=(0x1 | (0x40 if k % 2 else 0x80)) if paired else 0, refid=refid, pos=pos, cigar="10M", mapq=int(rng.integers(0, 3)),
                    next_refid=int(rng.integers(-1, 2)), pnext=int(rng.integers(0, 4)), tlen=int(rng.integers(-2, 3)), seq="A" * 10, qual=[30] * 10, rgid=0)
    heads = ["%020d" % rng.integers(0, 10**18) for _ in range(40)]
    for k in range(3000):  # run 1: 40 heads of 20 digits (the key ends inside or right behind them), tails that differ behind the key, exact ties
        tail = "%020d" % rng.integers(0, 10**18) if rng.integers(0, 4) else "7" * 20
        recs.append(rec("N" + heads[int(rng.integers(0, 40))] + tail, 0, 100, k))
    for k in range(400):   # run 2: names of two alternatives per position (one bit each), of several lengths
        recs.append(rec("N" + "".join("ab"[int(x)] for x in rng.integers(0, 2, int(rng.integers(30, 42)))), 0, 200, k))
    if big_group:          # run 3: 1500 names that agree in the 33 leading positions and differ behind them
        for k in range(1500):
            recs.append(rec("N" + "5" * 32 + "%07d" % rng.integers(0, 3000), 1, 50, k))
    for k in range(600):   # the unmapped block: pairs of mates with one name
        recs.append(dict(qname="N%030d" % rng.integers(0, 250), flag=0x4 | 0x1 | (0x40 if k % 2 else 0x80) | 0x8, refid=-1, pos=0, cigar="*", seq="A" * 10,
                         qual=[30] * 10, rgid=0))
    for k in range(100):
        recs.append(rec("s%d" % k, 1, int(rng.integers(1, 30)), k))
    order = rng.permutation(len(recs))
    b = batch_from_records([recs[i] for i in order])
    h = Header(ref_len=np.array([1000, 1000], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))
    e = Engine(h, tuning={"tie_rounds": tie_rounds, "sort_pairs": tie_rounds})
    e.stage(b)
    assert np.array_equal(e.sort_coordinate(), orc.sort_coordinate(b))
    e.close()


@pytest.mark.parametrize("radix_tile,sort_pairs", [(1, 0), (2, 0), (3, 0), (1, 1), (3, 1)])
def test_radix_passes_in_tiles_of_4096_and_8192_keys(radix_tile, sort_pairs):
    """every radix pass of the path (coordinate sort, tie-break rounds, the pair list's partition with its device-side length, the
    metrics' group sort) in tiles of 4096 keys (256 threads) and of 8192 (512 threads, what arrays of 8 M keys and more take by
    themselves; 16384: 1024 threads), the coordinate sort on words key << b | index and on (key, index) pairs: permutation, flags and
    counters are the oracle's"""
    cfg, b, h, refs, sites = dataset("tiny", 40000, 9, 0.03)
    oflags, octr, ohist = orc.dup_metrics(b, h, None, 100, hist_len=16)
    e = Engine(h, tuning={"radix_tile": radix_tile, "sort_pairs": sort_pairs})
    e.stage(b)
    flags = e.mark_duplicates(True)
    perm = e.sort_coordinate()
    ctr, hist = e.dup_metrics(100, hist_len=16)
    e.close()
    assert np.array_equal(flags, oflags)
    assert np.array_equal(ctr, octr) and np.array_equal(hist, ohist)
    assert np.array_equal(perm, orc.sort_coordinate(b, oflags))
