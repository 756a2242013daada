"""Known-answer tests that pin the CPU oracle.

(a) the reference's own interval tests (intervals/intervals_test.go:53-213): their vectors, kept as data in tests/golden/;
(b) hand-derived vectors of SURVEY.md §8(c), re-derived here from the cited reference lines.
The oracle is otherwise PARITY UNPINNED (the reference has no tests for sort/markdup/BQSR).
"""
import json
import os

import numpy as np
import pytest

import oracle as orc
from elprep_amd.batch import parse_cigar, batch_from_records, Header, NIL16


def iv(x):
    return np.asarray(x, dtype=np.int32).reshape(-1, 2)


# ---------- (a) intervals/intervals_test.go: the vectors live in tests/golden/intervals_test_go.json ----------
_GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "intervals_test_go.json")))


@pytest.mark.parametrize("case", _GOLD["flatten"])
def test_flatten_reference_vectors(case):  # :53-73
    assert orc.flatten(iv(case["in"])).tolist() == iv(case["out"]).tolist()


def test_flatten_large_random():  # :75-84 with makeLargeIntervalsSlice :38-50
    rng = np.random.default_rng(7)
    start = rng.integers(0, 1 << 20, 0x30000)
    raw = np.stack([start, start + rng.integers(0, 64, start.size)], axis=1)
    out = orc.flatten(orc.sort_by_start(raw))
    assert (out[:, 0] <= out[:, 1]).all()
    assert (out[1:, 0] > out[:-1, 1]).all()


@pytest.mark.parametrize("case", _GOLD["overlap"])
def test_overlap_reference_vectors(case):  # :137-174
    assert orc.overlap(iv(case["intervals"]), case["start"], case["end"]) == case["out"]


@pytest.mark.parametrize("case", _GOLD["intersect"])
def test_intersect_reference_vectors(case):  # :176-213
    assert orc.intersect(iv(case["intervals"]), case["start"], case["end"]).tolist() == iv(case["out"]).tolist()


# ---------- (b) hand-derived KATs ----------
@pytest.mark.parametrize("pos,flag,cigar,exp", [
    (100, 0, "5S95M", 95), (100, 0, "3H2S95M", 95), (100, 16, "90M10S", 199), (100, 16, "5S95M", 194),
    (100, 16, "50M2D50M", 201), (100, 16, "10S80M5I5M10S", 194), (100, 16, "95M5S3H", 202), (100, 16, "150S", 249),
    (100, 0, "150S", -50), (100, 0, "*", 100), (100, 16, "*", 100),
])
def test_unclipped_position(pos, flag, cigar, exp):  # filters/mark-duplicates.go:79-110
    assert orc.unclipped_position(pos, flag, parse_cigar(cigar)) == exp


def test_phred_score():  # filters/mark-duplicates.go:36-68
    assert orc.phred_score(np.array([14, 15, 40], dtype=np.uint8)) == 55
    assert orc.phred_score(np.array([], dtype=np.uint8)) == 0
    assert orc.phred_score(np.array([93, 93], dtype=np.uint8)) == 186
    with pytest.raises(ValueError):
        orc.phred_score(np.array([94], dtype=np.uint8))


def test_mod_flag():  # sam/sam-types.go:408-421
    assert orc.mod_flag(0x30) == 0x10
    assert orc.mod_flag(0x1 | 0x8 | 0x20) == 0x1 | 0x8
    assert orc.mod_flag(0x4 | 0x10) == 0x4
    assert orc.mod_flag(99) == 99


@pytest.mark.parametrize("flag,l,i,exp", [
    (0x40, 150, 0, 1), (0x40, 150, 149, 150), (0x40 | 0x10, 150, 0, 150), (0x80, 150, 0, -1),
    (0x80 | 0x10, 150, 0, -150), (0x80 | 0x10, 150, 149, -1),
])
def test_cycle(flag, l, i, exp):  # filters/bqsr.go:376-387
    assert orc.cycle(flag, l, i) == exp


@pytest.mark.parametrize("seq,exp", [
    (b"ACGT", [-1, 66, 146, 226]), (b"NACGT", [-1, -1, 66, 146, 226]), (b"ANCGT", [-1, -1, -1, 146, 226]),
    (b"ACNGT", [-1, 66, -1, -1, 226]), (b"NNACG", [-1, -1, -1, 66, 146]), (b"A", [-1]), (b"", []), (b"TT", [-1, 242]),
])
def test_context_with(seq, exp):  # filters/bqsr.go:87-131
    assert orc.context_with(seq).tolist() == exp


@pytest.mark.parametrize("cigar,ss,ref,left,right", [
    ("10M", 100, 105, (5, True), (5, True)), ("10M", 100, 100, (0, True), (0, True)), ("10M", 100, 109, (9, True), (9, True)),
    ("10M", 100, 110, (-1, False), (-1, False)), ("10M", 100, 99, (-1, False), (-1, False)),
    ("5M2D5M", 100, 104, (4, True), (4, True)), ("5M2D5M", 100, 105, (4, True), (5, True)),
    ("5M2D5M", 100, 106, (4, True), (5, True)), ("5M2D5M", 100, 107, (5, True), (5, True)),
    ("5M3I5M", 100, 104, (4, True), (4, True)), ("5M3I5M", 100, 105, (8, True), (8, True)),
    ("3I7M", 100, 100, (3, True), (0, True)), ("3S7M", 97, 100, (3, True), (3, True)),
])
def test_read_coordinate_for_reference_coordinate(cigar, ss, ref, left, right):  # filters/utils.go:267-349
    c = parse_cigar(cigar)
    assert orc.read_coordinate_for_reference_coordinate(c, ss, ref, False) == left
    assert orc.read_coordinate_for_reference_coordinate(c, ss, ref, True) == right


def test_tile_info():  # filters/mark-optical-duplicates.go:50-71
    assert orc.tile_info(b"SIM:1:FC1:3:1101:12345:6789") == (1101, 12345, 6789)
    assert orc.tile_info(b"FC:3:1101:12345:6789") == (1101, 12345, 6789)
    assert orc.tile_info(b"read1") == (-1, -1, -1)
    assert orc.tile_info(b"a:b:c:d:e:f") == (-1, -1, -1)


def test_contig_groups():  # sam/split-merge.go:178-213
    n, g = orc.contig_groups(np.array([100, 40, 50, 30, 100], dtype=np.int32))
    # target = 100: [100] | [40,50] | [30] would overflow with 100 -> [30] | [100]
    assert g.tolist() == [1, 2, 2, 3, 4] and n == 4
    n, g = orc.contig_groups(np.array([10, 10, 10], dtype=np.int32), 20)
    assert g.tolist() == [1, 1, 2] and n == 2


def _rec(qname, flag, refid, pos, cigar="10M", mapq=60, qual=None, next_refid=-1, pnext=0, tlen=0, rgid=0, seq=None):
    n = sum(int(c) >> 4 for c in parse_cigar(cigar) if (int(c) & 0xF) in (0, 1, 4, 7, 8)) if cigar != "*" else 10
    return dict(qname=qname, flag=flag, refid=refid, pos=pos, cigar=cigar, mapq=mapq, next_refid=next_refid, pnext=pnext,
                tlen=tlen, rgid=rgid, seq=seq or "A" * n, qual=qual if qual is not None else [30] * n)


def test_coordinate_order_rules():  # sam/sam-types.go:425-473
    recs = [
        _rec("u", 4, -1, 0, "*"),          # 0 unmapped: last
        _rec("b", 16, 0, 100),             # 1 reverse at 100
        _rec("a", 0, 0, 100),              # 2 forward at 100 -> before reverse
        _rec("c", 0, 1, 5),                # 3 refid 1
        _rec("a", 0, 0, 99),               # 4
        _rec("a", 0, 0, 100, mapq=10),     # 5 same as 2 but lower MAPQ -> before 2
        _rec("Z", 0, 0, 100),              # 6 'Z' < 'a' bytewise
        _rec("a", 0, 0, 100, tlen=-5),     # 7 same as 2 but smaller TLEN -> before 2 (after 5: mapq 10 < 60)
    ]
    b = batch_from_records(recs)
    perm = orc.sort_coordinate(b).tolist()
    assert perm == [4, 6, 5, 7, 2, 1, 3, 0]


def test_sort_is_stable_for_identical_records():
    recs = [_rec("x", 0, 0, 10) for _ in range(5)] + [_rec("w", 0, 0, 10)]
    perm = orc.sort_coordinate(batch_from_records(recs)).tolist()
    assert perm == [5, 0, 1, 2, 3, 4]


def _hdr(n_ref=2, libs=(0,), ref_len=1000):
    rg_lib = np.asarray(libs, dtype=np.uint16)
    return Header(ref_len=np.full(n_ref, ref_len, dtype=np.int32), rg_lib=rg_lib, rg_cov=np.arange(len(libs), dtype=np.uint16))


def test_markdup_fragments():  # filters/mark-duplicates.go:210-254
    h = _hdr()
    q = lambda v: [v] * 10
    recs = [
        _rec("f1", 0, 0, 100, qual=q(30)),   # score 300
        _rec("f2", 0, 0, 100, qual=q(35)),   # score 350  -> best
        _rec("f3", 0, 0, 100, qual=q(20)),   # score 200
        _rec("f4", 16, 0, 100, qual=q(20)),  # other strand: own group (upos = 109), alone
        _rec("f5", 0, 0, 102, "2S8M", qual=q(35)),  # upos 100, score 350, QNAME 'f5' > 'f2' -> duplicate
        _rec("f0", 0, 0, 100, qual=q(35)),   # ties on score, QNAME 'f0' < 'f2' -> new best, f2 becomes duplicate
    ]
    flags = orc.mark_duplicates(batch_from_records(recs), h)
    assert [(int(f) & 0x400) != 0 for f in flags] == [True, True, True, False, True, False]


def test_markdup_pair_knocks_out_fragments_and_pairs():  # :222-253, :329-396
    h = _hdr()
    q = lambda v: [v] * 10
    recs = [
        _rec("frag", 0, 0, 100, qual=q(40)),                                            # true fragment at (0,100,+)
        _rec("p1", 99, 0, 100, qual=q(20), next_refid=0, pnext=300, tlen=210),           # pair read, same fragment key
        _rec("p1", 147, 0, 300, qual=q(20), next_refid=0, pnext=100, tlen=-210),
        _rec("p2", 99, 0, 100, qual=q(30), next_refid=0, pnext=300, tlen=210),           # better pair, same ends
        _rec("p2", 147, 0, 300, qual=q(30), next_refid=0, pnext=100, tlen=-210),
        _rec("mu", 73, 0, 100, qual=q(40), next_refid=0, pnext=100),                     # mate unmapped => true fragment
        _rec("mu", 133, 0, 100, "*", mapq=0, qual=q(40), next_refid=0, pnext=100),       # unmapped mate: not a candidate
    ]
    flags = orc.mark_duplicates(batch_from_records(recs), h)
    dup = [(int(f) & 0x400) != 0 for f in flags]
    assert dup == [True, True, True, False, False, True, False]


def test_markdup_library_separates_groups():
    h = _hdr(libs=(0, 1, NIL16))
    q = [30] * 10
    recs = [_rec("a", 0, 0, 100, qual=q, rgid=0), _rec("b", 0, 0, 100, qual=q, rgid=1), _rec("c", 0, 0, 100, qual=q, rgid=2),
            _rec("d", 0, 0, 100, qual=q, rgid=NIL16), _rec("e", 0, 0, 100, qual=q, rgid=0)]
    flags = orc.mark_duplicates(batch_from_records(recs), h)
    # rg2 (no LB) and the read without RG share the nil library; a/e share lib 0 (a < e wins on QNAME)
    assert [(int(f) & 0x400) != 0 for f in flags] == [False, False, False, True, True]


def test_float_helpers():
    assert orc.go_log10(1000.0) == pytest.approx(3.0, abs=1e-15)
    assert orc.go_log10(0.001) == pytest.approx(-3.0, abs=1e-15)
    assert orc.go_pow10(-3.0) == pytest.approx(1e-3, rel=1e-15)
    assert orc.go_pow10(2.0) == 100.0
    # bqsr.go:623-642: 1002 observations, 2 mismatches (smoothed), prior 30 -> posterior mode stays near Q27..30
    assert 25 <= orc.bayesian_estimate(1002, 2, 30.0) <= 30
    # a large clean sample at reported Q30 keeps Q30
    assert orc.bayesian_estimate(10_000_002, 10_001, 30.0) == 30
    # library size estimate (mark-optical-duplicates.go:541-569): no duplicates -> 0
    assert orc.estimate_library_size(1000, 1000) == 0
    assert orc.estimate_library_size(1000, 900) > 900


def test_duplicate_set_size_histograms():  # filters/mark-optical-duplicates.go:150-174, 275-325, 469-525 (hand-derived)
    """One set of four pairs at the same ends: the origin (best score) and two of its duplicates lie within 100 px of each other on one
    tile (a cluster of three listed reads = 2 optical duplicates, :244-273), the third duplicate is on another tile -> bins all[4],
    non_optical[4 - 2], optical[2 + 1]; one pair without duplicates -> all[1], non_optical[1]."""
    h = _hdr()
    q = lambda v: [v] * 10
    recs = []
    for name, score in (("M:1:F:1:1101:1000:2000", 40), ("M:1:F:1:1101:1050:2050", 20), ("M:1:F:1:1101:1080:2010", 20), ("M:1:F:1:2205:1000:2000", 20)):
        recs.append(_rec(name, 99, 0, 100, qual=q(score), next_refid=0, pnext=300, tlen=210))
        recs.append(_rec(name, 147, 0, 300, qual=q(score), next_refid=0, pnext=100, tlen=-210))
    recs.append(_rec("M:1:F:1:1101:5:5", 99, 0, 500, qual=q(30), next_refid=0, pnext=700, tlen=210))
    recs.append(_rec("M:1:F:1:1101:5:5", 147, 0, 700, qual=q(30), next_refid=0, pnext=500, tlen=-210))
    b = batch_from_records(recs)
    perm = orc.sort_coordinate(b)
    flags, ctr, hist = orc.dup_metrics(b, h, perm, 100, hist_len=8)
    assert ctr[0, 1] == 5 and ctr[0, 5] == 3 and ctr[0, 6] == 2   # pairs examined, pair duplicates, optical duplicates
    want = np.zeros((3, 8), np.int64)
    want[0, 4] = 1; want[1, 2] = 1; want[2, 3] = 1
    want[0, 1] = 1; want[1, 1] = 1
    assert np.array_equal(hist[0], want), hist[0]


@pytest.mark.parametrize("rec,exp", [
    # forward read reading through into the adaptor: boundary = POS + |TLEN| = 130 lies inside [100, 149] -> right tail from read
    # coordinate 30 on is clipped (filters/utils.go:149-180, 214-222, 253-262): bases [0, 30), 30M20H, POS unchanged
    (dict(flag=0x1 | 0x20 | 0x40, pos=100, cigar="50M", pnext=110, tlen=30), (0, 30, 100, [(30, "M"), (20, "H")])),
    # reverse read, mate forward at 120: boundary = PNEXT - 1 = 119 -> left tail up to read coordinate 19 is clipped: bases
    # [20, 50), 20H30M, POS moves by the 20 clipped bases (calculateAlnStartShift)
    (dict(flag=0x1 | 0x10 | 0x80, pos=100, cigar="50M", pnext=120, tlen=-30), (20, 50, 120, [(20, "H"), (30, "M")])),
    # soft clips become hard clips (hardClipSoftClippedBases); POS is the first aligned base already
    (dict(flag=0, pos=100, cigar="5S40M5S", pnext=0, tlen=0), (5, 45, 100, [(5, "H"), (40, "M"), (5, "H")])),
    # insert longer than the read: the boundary (350) lies behind the read's end (149): nothing to clip
    (dict(flag=0x1 | 0x20 | 0x40, pos=100, cigar="50M", pnext=300, tlen=250), (0, 50, 100, [(50, "M")])),
])
def test_clipping_for_bqsr(rec, exp):  # hand-derived from filters/utils.go:149-262, 374-470
    r = dict(qname="q", refid=0, mapq=60, next_refid=0, seq="A" * 50, qual=[30] * 50, rgid=0)
    r.update(rec)
    a, e, npos, cg = orc.clip_for_bqsr(batch_from_records([r]), 0)
    assert (a, e, npos, [(int(c) >> 4, "MIDNSHP=X"[int(c) & 15]) for c in cg]) == exp


def test_bqsr_tables_of_one_read_from_first_principles():  # filters/bqsr.go:254-285 (SNP events), 376-387 (cycle), 87-146 (context)
    """A 12-base forward single-end read, all qualities 30, one mismatch against the reference at read index 5, no known sites:
    every base is observed once in its (quality, cycle) cell, cycle = index + 1; bases 1..11 once in the context cell of
    (previous base, base) with index prev | cur << 2 (A0 C1 G2 T3); the mismatch counts in all three tables."""
    from elprep_amd.batch import Header
    seq = "ACGTACGTACGT"
    ref = list("N" * 10 + seq + "N" * 8)
    ref[10 + 5] = "A"  # the read has C there
    refb = np.frombuffer("".join(ref).encode(), dtype=np.uint8)
    b = batch_from_records([dict(qname="q", flag=0, refid=0, pos=11, cigar="12M", mapq=60, seq=seq, qual=[30] * 12, rgid=0)])
    h = Header(ref_len=np.array([len(ref)], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))
    qt, ct, xt = orc.bqsr_gather(b, h, orc.BqsrRef([refb], [np.zeros((0, 2), np.int32)]), None, 500)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    wq = np.zeros_like(qt); wc = np.zeros_like(ct); wx = np.zeros_like(xt)
    for i in range(12):
        e = 1 if i == 5 else 0
        wq[0, 30] += (1, e)
        wc[0, 30, 500 + (i + 1)] += (1, e)
        if i >= 1:
            wx[0, 30, code[seq[i - 1]] | (code[seq[i]] << 2)] += (1, e)
    assert np.array_equal(qt, wq) and np.array_equal(ct, wc) and np.array_equal(xt, wx)
    # the same read as the reverse-strand second of a pair: cycles are negative and run from the read's end (-(12 - i)), the
    # context is taken on the reverse complement (previous base = the complement of base i + 1)
    b2 = batch_from_records([dict(qname="q", flag=0x1 | 0x10 | 0x80, refid=0, pos=11, cigar="12M", mapq=60, next_refid=0, pnext=400, tlen=0,
                                  seq=seq, qual=[30] * 12, rgid=0)])
    qt, ct, xt = orc.bqsr_gather(b2, h, orc.BqsrRef([refb], [np.zeros((0, 2), np.int32)]), None, 500)
    comp = {"A": 3, "C": 2, "G": 1, "T": 0}
    wc[:] = 0; wx[:] = 0
    for i in range(12):
        e = 1 if i == 5 else 0
        wc[0, 30, 500 - (12 - i)] += (1, e)
        if i <= 10:
            wx[0, 30, comp[seq[i + 1]] | (comp[seq[i]] << 2)] += (1, e)
    assert np.array_equal(qt, wq) and np.array_equal(ct, wc) and np.array_equal(xt, wx)


def test_bqsr_known_site_skips_only_its_own_bases():  # filters/bqsr.go:389-414 (calculateSkipSlice), :301-305
    """A known site over reference positions 14..15 (read indices 3, 4 of a read at POS 11): exactly those two bases drop out of all
    three tables; base 5 keeps its context (the base in front of it is still its predecessor)."""
    from elprep_amd.batch import Header
    seq = "ACGTACGTACGT"
    refb = np.frombuffer(("N" * 10 + seq + "N" * 8).encode(), dtype=np.uint8)
    b = batch_from_records([dict(qname="q", flag=0, refid=0, pos=11, cigar="12M", mapq=60, seq=seq, qual=[30] * 12, rgid=0)])
    h = Header(ref_len=np.array([len(refb)], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))
    qt, ct, xt = orc.bqsr_gather(b, h, orc.BqsrRef([refb], [np.array([[14, 15]], np.int32)]), None, 500)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    wq = np.zeros_like(qt); wc = np.zeros_like(ct); wx = np.zeros_like(xt)
    for i in range(12):
        if i in (3, 4):
            continue
        wq[0, 30, 0] += 1
        wc[0, 30, 500 + (i + 1), 0] += 1
        if i >= 1:
            wx[0, 30, code[seq[i - 1]] | (code[seq[i]] << 2), 0] += 1
    assert np.array_equal(qt, wq) and np.array_equal(ct, wc) and np.array_equal(xt, wx)


def test_bqsr_low_quality_tails_mask_the_context():  # filters/bqsr.go:310-345 (computeStrandedClippedSeq), :301-305 (quality < 6)
    """Qualities <= 2 at both ends: those bases read as N for the context covariate and are not counted themselves (quality < 6);
    the first base behind the left tail is counted for quality and cycle but has no context (its predecessor is N)."""
    from elprep_amd.batch import Header
    seq = "ACGTACGTACGT"
    refb = np.frombuffer(("N" * 10 + seq + "N" * 8).encode(), dtype=np.uint8)
    qual = [2, 2] + [30] * 9 + [2]
    b = batch_from_records([dict(qname="q", flag=0, refid=0, pos=11, cigar="12M", mapq=60, seq=seq, qual=qual, rgid=0)])
    h = Header(ref_len=np.array([len(refb)], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))
    qt, ct, xt = orc.bqsr_gather(b, h, orc.BqsrRef([refb], [np.zeros((0, 2), np.int32)]), None, 500)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    wq = np.zeros_like(qt); wc = np.zeros_like(ct); wx = np.zeros_like(xt)
    for i in range(2, 11):
        wq[0, 30, 0] += 1
        wc[0, 30, 500 + (i + 1), 0] += 1
        if i >= 3:
            wx[0, 30, code[seq[i - 1]] | (code[seq[i]] << 2), 0] += 1
    assert np.array_equal(qt, wq) and np.array_equal(ct, wc) and np.array_equal(xt, wx)


def test_bqsr_insertion_and_deletion_in_snp_events():  # filters/bqsr.go:254-285
    """5M2I5M against a reference without the two inserted bases, then 5M3D7M against a reference with three extra bases: inserted
    bases are observed and never mismatch, a deletion only advances the reference; one planted mismatch behind each indel lands at
    its read index (cycle = index + 1)."""
    from elprep_amd.batch import Header
    seq = "ACGTACGTACGT"
    h1 = lambda n: Header(ref_len=np.array([n], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))
    no_sites = [np.zeros((0, 2), np.int32)]
    # insertion: read indices 5, 6 are inserted; read index 8 is aligned to reference offset 6
    ref = list("N" * 10 + seq[:5] + seq[7:] + "N" * 8)
    ref[10 + 6] = "G"  # read has A at index 8
    refb = np.frombuffer("".join(ref).encode(), dtype=np.uint8)
    b = batch_from_records([dict(qname="q", flag=0, refid=0, pos=11, cigar="5M2I5M", mapq=60, seq=seq, qual=[30] * 12, rgid=0)])
    qt, ct, xt = orc.bqsr_gather(b, h1(len(refb)), orc.BqsrRef([refb], no_sites), None, 500)
    assert qt[0, 30].tolist() == [12, 1]
    assert ct[0, 30, 501:513, 0].tolist() == [1] * 12 and np.nonzero(ct[0, 30, :, 1])[0].tolist() == [500 + 9]
    # deletion: read index 5.. continues three reference bases later; read index 6 is aligned to reference offset 9
    ref = list("N" * 10 + seq[:5] + "TTT" + seq[5:] + "N" * 8)
    ref[10 + 9] = "A"  # read has G at index 6
    refb = np.frombuffer("".join(ref).encode(), dtype=np.uint8)
    b = batch_from_records([dict(qname="q", flag=0, refid=0, pos=11, cigar="5M3D7M", mapq=60, seq=seq, qual=[30] * 12, rgid=0)])
    qt, ct, xt = orc.bqsr_gather(b, h1(len(refb)), orc.BqsrRef([refb], no_sites), None, 500)
    assert qt[0, 30].tolist() == [12, 1]
    assert ct[0, 30, 501:513, 0].tolist() == [1] * 12 and np.nonzero(ct[0, 30, :, 1])[0].tolist() == [500 + 7]


def test_bayesian_estimate_against_a_second_restatement():  # filters/bqsr.go:560-642
    """calculateBayesianEstimateOfEmpiricalQuality written a second time from the source, in Python: Gaussian prior table
    log10(0.9) - 2 d^2 log10(e) for d = 0..19 (what the reference's literal table holds), -MaxFloat64 from d = 20 on; binomial
    log-likelihood with Lgamma; first maximum over the 61 candidate qualities wins.  Cases where the two best posteriors are closer
    than 1e-9 are skipped (libm vs Python last-bit differences could flip them)."""
    import math
    import sys
    log10e = math.log10(math.e)
    prior = [math.log10(0.9) - 2.0 * d * d * log10e for d in range(20)] + [-sys.float_info.max]

    def post(i, n, k, pq):
        p1 = prior[min(abs(int(float(i) - pq)), 20)]
        if n == 0:
            return p1
        lp = float(i) / -10.0
        if lp == 0.0:
            return p1 + -sys.float_info.max
        coeff = (math.lgamma(n + 1) - math.lgamma(k + 1) - math.lgamma(n - k + 1)) * log10e
        return p1 + coeff + lp * k + math.log10(1.0 - 10.0 ** lp) * (n - k)

    rng = np.random.default_rng(9)
    checked = 0
    for _ in range(400):
        n = int(rng.choice([0, 1, 2, 10, 100, 5000, 10 ** 6, 3 * 10 ** 8]))
        k = int(rng.integers(0, min(n, 10 ** 5) + 1)) if n else 0
        pq = float(rng.choice([0.0, 2.0, 17.0, 30.0, 37.5, 45.0, 60.0, 93.0]))
        ps = [post(i, n, k, pq) for i in range(61)]
        best = max(range(61), key=lambda i: (ps[i], -i))
        top = sorted(ps, reverse=True)
        if top[0] - top[1] < 1e-9 * max(1.0, abs(top[0])):
            continue
        assert orc.bayesian_estimate(n, k, pq) == best, (n, k, pq)
        checked += 1
    assert checked > 300
    # the literal value the table starts with and ends with before the sentinel
    assert abs(prior[5] - -21.760481585723266) < 1e-9 and abs(prior[19] - -313.60637342472336) < 1e-7


def test_estimate_library_size_against_a_second_restatement():  # filters/mark-optical-duplicates.go:532-569
    """The bisection of estimateLibrarySize written a second time in Python, and one value that can be checked by hand: with
    n = 1000 pairs and c = 900 unique ones the root x of c/x - 1 + exp(-n/x) lies near 4 600 molecules."""
    import math

    def f(x, c, n):
        return c / x - 1 + math.exp(-n / x)

    def est(n_pairs, n_unique):
        n, c = float(n_pairs), float(n_unique)
        if not (n_pairs > 0 and n_pairs - n_unique > 0):
            return 0
        m, M = 1.0, 100.0
        while f(M * c, c, n) >= 0.0:
            M *= 10.0
        for _ in range(40):
            r = (m + M) / 2.0
            u = f(r * c, c, n)
            if u == 0.0:
                break
            if u > 0.0:
                m = r
            if u < 0.0:
                M = r
        return int(c * ((m + M) / 2.0))

    rng = np.random.default_rng(4)
    for _ in range(200):
        n = int(rng.integers(1, 10 ** 7))
        c = int(rng.integers(max(1, n // 50), n + 1))
        assert orc.estimate_library_size(n, c) == est(n, c), (n, c)
    x = orc.estimate_library_size(1000, 900)
    assert abs(f(float(x), 900.0, 1000.0)) < 1e-4 and 4000 < x < 5200  # x is truncated to an integer
    assert orc.estimate_library_size(1000, 1000) == 0 and orc.estimate_library_size(0, 0) == 0


def test_hierarchical_estimate_against_a_second_restatement():  # filters/bqsr.go:899-919, 970-999
    """estimateHierarchicalBayesianQuality + the final rounding / bounding, written a second time in Python on top of the Bayesian
    estimate (checked above), for tables with ONE reported quality per read group (then the combined read-group entry is that entry
    and no map-iteration order enters): cycle and context entries present or absent."""
    rng = np.random.default_rng(12)

    def emp(obs, mism, prior):  # calculateEmpiricalQuality :644-649
        return min(orc.bayesian_estimate(int(obs) + 2, int(mism) + 1, float(prior)), 93)

    checked = 0
    for trial in range(40):
        q = int(rng.choice([8, 20, 30, 37]))
        qt = np.zeros((1, 94, 2), np.int64); ct = np.zeros((1, 94, 1001, 2), np.int64); xt = np.zeros((1, 94, 16, 2), np.int64)
        cycles = rng.choice(np.arange(-150, 151), size=6, replace=False)
        cycles = cycles[cycles != 0]
        for cy in cycles:
            o = int(rng.integers(1, 20000)); ct[0, q, 500 + cy] = (o, int(rng.integers(0, o // 20 + 1)))
        for cx in rng.choice(16, size=5, replace=False):
            o = int(rng.integers(1, 50000)); xt[0, q, cx] = (o, int(rng.integers(0, o // 20 + 1)))
        qt[0, q] = ct[0, q].sum(axis=0)
        fo = orc.BqsrFinal(qt, ct, xt, 500)
        _, quantized = fo.quantize(0)
        eps = float(q)
        d_g = emp(qt[0, q, 0], qt[0, q, 1], eps) - eps
        d_q = emp(qt[0, q, 0], qt[0, q, 1], d_g + eps) - d_g - eps
        cond = d_q + d_g + eps
        for cy in list(cycles[:3]) + [151 if 151 not in cycles else 152]:
            for cx in (int(np.nonzero(xt[0, q, :, 0])[0][0]), int(np.nonzero(xt[0, q, :, 0] == 0)[0][0]), -1):
                d_c = 0.0
                if ct[0, q, 500 + cy, 0] > 0:
                    d_c = emp(ct[0, q, 500 + cy, 0], ct[0, q, 500 + cy, 1], cond) - cond
                if cx >= 0 and xt[0, q, cx, 0] > 0:
                    d_c += emp(xt[0, q, cx, 0], xt[0, q, cx, 1], cond) - cond
                est = cond + d_c
                want = int(quantized[max(1, min(int(np.floor(est + 0.5)) if est >= 0 else -int(np.floor(-est + 0.5)), 93))])
                key = -1 if cx < 0 else (2 | ((cx & 3) << 4) | ((cx >> 2) << 6))
                assert fo.recal_qual(0, q, int(cy), key, quantized, None) == want, (trial, q, cy, cx)
                checked += 1
    assert checked > 300


def test_coordinate_sort_against_a_second_restatement():  # sam/sam-types.go:408-473, :639-641 (stable sort)
    """CoordinateLess written a second time in Python and fed to Python's stable sort, on random records full of ties (few positions,
    few names, paired and unpaired, unmapped, both strands)."""
    import functools
    rng = np.random.default_rng(21)

    def mod_flag(f):
        if f & 0x1 == 0:
            f &= ~0x8 & ~0x20
        if f & 0x4:
            f &= ~0x10
        if f & 0x8:
            f &= ~0x20
        return f

    def less(a, b):
        r1, r2 = a["refid"], b["refid"]
        if r1 < r2:
            return r1 >= 0
        if r2 < r1:
            return r2 < 0
        if a["pos"] != b["pos"]:
            return a["pos"] < b["pos"]
        rv1, rv2 = bool(a["flag"] & 0x10), bool(b["flag"] & 0x10)
        if rv1 != rv2:
            return not rv1
        if a["qname"] != b["qname"]:
            return a["qname"].encode() < b["qname"].encode()
        f1, f2 = mod_flag(a["flag"]), mod_flag(b["flag"])
        if f1 != f2:
            return f1 < f2
        if a["mapq"] != b["mapq"]:
            return a["mapq"] < b["mapq"]
        if (a["flag"] & 0x1) and (b["flag"] & 0x1):
            if a["next_refid"] != b["next_refid"]:
                return a["next_refid"] < b["next_refid"]
            if a["pnext"] != b["pnext"]:
                return a["pnext"] < b["pnext"]
        return a["tlen"] < b["tlen"]

    for trial in range(6):
        recs = []
        for k in range(400):
            unm = rng.random() < 0.15
            paired = rng.random() < 0.6
            flag = (0x1 if paired else 0) | (0x10 if rng.random() < 0.5 else 0) | (0x4 if unm else 0)
            if paired:
                flag |= (0x40 if rng.random() < 0.5 else 0x80) | (0x20 if rng.random() < 0.3 else 0) | (0x8 if rng.random() < 0.1 else 0)
            recs.append(dict(qname="n%d" % rng.integers(0, 12) + "x" * int(rng.integers(0, 2)), flag=flag,
                             refid=-1 if unm and rng.random() < 0.7 else int(rng.integers(0, 3)), pos=0 if unm else int(rng.integers(1, 6)),
                             cigar="*" if unm else "10M", mapq=int(rng.integers(0, 3)), next_refid=int(rng.integers(-1, 3)),
                             pnext=int(rng.integers(0, 4)), tlen=int(rng.integers(-2, 3)), seq="A" * 10, qual=[30] * 10, rgid=0))
        want = sorted(range(len(recs)), key=functools.cmp_to_key(lambda i, j: -1 if less(recs[i], recs[j]) else (1 if less(recs[j], recs[i]) else 0)))
        got = orc.sort_coordinate(batch_from_records(recs))
        assert got.tolist() == want, trial


def _restate_markdup(b, h):
    """filters/mark-duplicates.go:177-445 as plain sequential Python over dicts (one goroutine: arrival = index order).
    -> (flags, pair table {key: (score, aln1, aln2)}, unclipped positions, names, library per record)"""
    from elprep_amd.batch import NIL16
    _, upos, score = orc.mark_duplicates(b, h, with_adapted=True)
    flag = b.flag.astype(np.int64).copy()
    names = [b.qname_of(i) for i in range(b.n)]
    lib = [None if b.rgid[i] == NIL16 or h.rg_lib[b.rgid[i]] == NIL16 else int(h.rg_lib[b.rgid[i]]) for i in range(b.n)]
    true_pair = lambda i: (int(b.flag[i]) & (0x1 | 0x8)) == 0x1
    rev = lambda i: bool(int(b.flag[i]) & 0x10)
    frags, waiting, pairs = {}, {}, {}
    for i in range(b.n):
        if int(b.flag[i]) & (0x4 | 0x100 | 0x800):
            continue
        k = (lib[i], int(b.refid[i]), int(upos[i]), rev(i))
        if k not in frags:
            frags[k] = i
        else:
            best = frags[k]
            if not true_pair(i):
                if true_pair(best) or score[best] > score[i] or (score[best] == score[i] and names[i] > names[best]):
                    flag[i] |= 0x400
                else:
                    frags[k] = i; flag[best] |= 0x400
            elif not true_pair(best):
                frags[k] = i; flag[best] |= 0x400
        if not true_pair(i):
            continue
        wk = (lib[i], names[i])
        if wk not in waiting:
            waiting[wk] = i
            continue
        a1, a2 = i, waiting.pop(wk)
        sc = int(score[a1]) + int(score[a2])
        if (b.refid[a1] > b.refid[a2] or (b.refid[a1] == b.refid[a2] and (upos[a1] > upos[a2] or (upos[a1] == upos[a2] and rev(a1) and not rev(a2))))):
            a1, a2 = a2, a1
        pk = (lib[a1], int(b.refid[a1]), int(b.refid[a2]), (int(upos[a1]) << 32) + int(upos[a2]), rev(a1), rev(a2))
        if pk not in pairs:
            pairs[pk] = (sc, a1, a2)
            continue
        bs, b1, b2 = pairs[pk]
        if bs > sc or (bs == sc and names[a1] > names[b1]):
            flag[a1] |= 0x400; flag[a2] |= 0x400
        else:
            pairs[pk] = (sc, a1, a2); flag[b1] |= 0x400; flag[b2] |= 0x400
    return flag.astype(np.uint16), pairs, upos, names, lib


def test_mark_duplicates_against_a_second_restatement():  # filters/mark-duplicates.go:177-445, one goroutine (arrival = index order)
    """classifyFragment / classifyPair written a second time as plain sequential Python over dicts, on synthetic batches with many
    duplicates, on the same batches shuffled, and on a pile-up with exact (score, QNAME) ties."""
    from tests.common import dataset
    from elprep_amd.batch import Header
    restate = lambda bb, hh: _restate_markdup(bb, hh)[0]
    rng = np.random.default_rng(2)
    for seed in (0, 1):
        cfg, b, h, refs, sites = dataset("tiny", 2500, seed, 0.05)
        for bb in (b, b.take(rng.permutation(b.n))):
            got = orc.mark_duplicates(bb, h)
            assert np.array_equal(got, restate(bb, h)) and ((got & 0x400) != 0).sum() > 100
    q = [30] * 10
    recs = [dict(qname=n, flag=0, refid=0, pos=100, cigar="10M", mapq=60, seq="A" * 10, qual=q, rgid=0) for n in ("b", "a", "b", "a", "c")]
    recs += [dict(qname=n, flag=f, refid=0, pos=p, cigar="10M", mapq=60, next_refid=0, pnext=pn, tlen=t, seq="A" * 10, qual=q, rgid=0)
             for n in ("p2", "p1", "p2b") for f, p, pn, t in ((99, 200, 300, 110), (147, 300, 200, -110))]
    bb = batch_from_records(recs)
    hh = Header(ref_len=np.array([1000], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))
    assert np.array_equal(orc.mark_duplicates(bb, hh), restate(bb, hh))


def test_duplication_metrics_against_a_second_restatement():  # filters/mark-optical-duplicates.go:176-525
    """MarkOpticalDuplicates written a second time in Python on top of the mark-duplicates restatement: the seven counters per
    library, optical duplicates per duplicate set (pairwise rules for two and three listed reads, connected components per
    (read group, tile) from four on) and the three set-size histograms."""
    from tests.common import dataset
    cfg, b, h, refs, sites = dataset("tiny", 6000, 1, 0.05)
    flags, pairs, upos, names, lib = _restate_markdup(b, h)
    perm = orc.sort_coordinate(b)
    dist = 100
    nl = h.n_lib
    ctr = np.zeros((nl + 1, 7), np.int64)
    true_pair = lambda i: (int(b.flag[i]) & (0x1 | 0x8)) == 0x1
    rev = lambda i: bool(int(b.flag[i]) & 0x10)
    row = lambda i: nl if lib[i] is None else lib[i]
    waiting, lists = {}, {k: [] for k in pairs}
    for i in (int(x) for x in perm):
        f = int(flags[i])
        if f & 0x4:
            ctr[row(i), 3] += 1; continue
        if f & (0x100 | 0x800):
            ctr[row(i), 2] += 1; continue
        if true_pair(i):
            ctr[row(i), 1] += 1
        else:
            ctr[row(i), 0] += 1
        if not f & 0x400:
            continue
        if not true_pair(i):
            ctr[row(i), 4] += 1; continue
        wk = (lib[i], names[i])
        if wk not in waiting:
            waiting[wk] = i; continue
        a1, a2 = i, waiting.pop(wk)
        ctr[row(i), 5] += 1
        if (b.refid[a1] > b.refid[a2] or (b.refid[a1] == b.refid[a2] and (upos[a1] > upos[a2] or (upos[a1] == upos[a2] and rev(a1) and not rev(a2))))):
            a1, a2 = a2, a1
        pk = (lib[a1], int(b.refid[a1]), int(b.refid[a2]), (int(upos[a1]) << 32) + int(upos[a2]), rev(a1), rev(a2))
        if pairs[pk][1] != a1:
            lists[pk].append(a1 if int(b.flag[a1]) & 0x40 else a2)
    ctr[:, 1] //= 2

    def close(x, y):
        tx, ty = orc.tile_info(names[x]), orc.tile_info(names[y])
        return b.rgid[x] == b.rgid[y] and tx[0] != -1 and ty[0] != -1 and tx[0] == ty[0] and abs(tx[1] - ty[1]) <= dist and abs(tx[2] - ty[2]) <= dist

    def optical(members):
        n = len(members)
        if n < 2:
            return 0
        if n < 4:
            return min(sum(close(members[x], members[y]) for x in range(n) for y in range(x + 1, n)), n - 1)
        parent = list(range(n))
        def find(x):
            while parent[x] != x:
                x = parent[x]
            return x
        for x in range(n):
            for y in range(x + 1, n):
                if close(members[x], members[y]):
                    parent[find(y)] = find(x)
        return n - len({find(x) for x in range(n)})

    hl = 16
    hist = np.zeros((nl + 1, 3, hl), np.int64)
    for pk, (sc, a1, a2) in pairs.items():
        origin = a1 if int(b.flag[a1]) & 0x40 else a2
        members = [origin] + lists[pk]
        fw, rv = [m for m in members if not rev(m)], [m for m in members if rev(m)]
        opt = optical(fw) + optical(rv)
        r = nl if lib[a1] is None else lib[a1]
        ctr[r, 6] += opt
        n = len(members)
        hist[r, 0, min(n, hl - 1)] += 1
        if n - opt > 0:
            hist[r, 1, min(n - opt, hl - 1)] += 1
        if opt > 0:
            hist[r, 2, min(opt + 1, hl - 1)] += 1
    oflags, octr, ohist = orc.dup_metrics(b, h, perm, dist, hist_len=hl)
    assert np.array_equal(oflags, flags) and octr[:, 6].sum() > 0
    assert np.array_equal(octr, ctr) and np.array_equal(ohist, hist)


def test_bqsr_gather_against_a_second_restatement_on_plain_reads():  # filters/bqsr.go:225-551 for reads that need no clipping
    """Recalibrate written a second time in Python for unpaired reads with a single match operation (no adaptor, no soft clips):
    eligibility of a base (no known site, A/C/G/T, quality >= 6), mismatch against the reference, cycle, the two-base context on
    the stranded read with low-quality tails and N masked - on random reads of two read groups, both strands, with N bases, qualities
    below 6 and below 3, and random known sites."""
    from elprep_amd.batch import Header
    rng = np.random.default_rng(33)
    L_ref = 3000
    ref = rng.choice(list("ACGT"), size=L_ref)
    ref[rng.random(L_ref) < 0.01] = "N"
    refb = np.frombuffer("".join(ref).encode(), dtype=np.uint8)
    sites = np.sort(rng.choice(np.arange(1, L_ref - 5), size=60, replace=False))
    iv = orc.flatten(orc.sort_by_start(np.stack([sites, sites + rng.integers(0, 4, size=60)], axis=1)))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    recs = []
    for k in range(400):
        L = int(rng.integers(1, 60))
        pos = int(rng.integers(1, L_ref - L))
        seq = [ref[pos - 1 + i] if rng.random() > 0.05 and ref[pos - 1 + i] != "N" else "ACGT"[rng.integers(0, 4)] for i in range(L)]
        for i in range(L):
            if rng.random() < 0.02:
                seq[i] = "N"
        qual = rng.choice([2, 5, 11, 25, 37], size=L, p=[0.05, 0.05, 0.3, 0.3, 0.3])
        if rng.random() < 0.3:
            qual[: rng.integers(0, min(4, L) + 1)] = 2
        if rng.random() < 0.3 and L > 1:
            qual[L - int(rng.integers(1, min(4, L) + 1)):] = 2
        recs.append(dict(qname="r%d" % k, flag=16 if rng.random() < 0.5 else 0, refid=0, pos=pos, cigar="%dM" % L, mapq=int(rng.choice([0, 30, 60])),
                         seq="".join(seq), qual=qual.tolist(), rgid=int(rng.integers(0, 2))))
    b = batch_from_records(recs)
    h = Header(ref_len=np.array([L_ref], np.int32), rg_lib=np.array([0, 0], np.uint16), rg_cov=np.array([0, 1], np.uint16))
    wq = np.zeros((2, 94, 2), np.int64); wc = np.zeros((2, 94, 1001, 2), np.int64); wx = np.zeros((2, 94, 16, 2), np.int64)
    for r in recs:
        if not (0 < r["mapq"] < 255):
            continue
        seq, qual, L, pos, revd, cov = r["seq"], r["qual"], len(r["seq"]), r["pos"], bool(r["flag"] & 16), r["rgid"]
        skip = [False] * L
        for s, e in iv:
            if e >= pos and s <= pos + L - 1:  # intersects [softStart, softEnd]
                lo = s - pos if 0 <= s - pos < L else 0
                hi = e - pos if 0 <= e - pos <= L - 1 else L - 1
                for i in range(lo, hi + 1):
                    skip[i] = True
        good = [i for i in range(L) if qual[i] > 2]
        if good:
            left, right = good[0], good[-1]
            strand = [seq[i] if left <= i <= right else "N" for i in range(L)]
        else:
            strand = None
        for i in range(L):
            if skip[i] or seq[i] not in code or qual[i] < 6:
                continue
            e = 1 if seq[i] != ref[pos - 1 + i] else 0  # baseToIntMap: N in the reference is its own class
            q = qual[i]
            cyc = (L - i) if revd else (i + 1)
            wq[cov, q] += (1, e); wc[cov, q, 500 + cyc] += (1, e)
            if strand is None:
                continue
            if not revd:
                prev, cur = (strand[i - 1] if i >= 1 else None), strand[i]
            else:  # the stranded read is the reverse complement: the base sequenced before base i is base i + 1
                prev, cur = (comp.get(strand[i + 1], "N") if i + 1 < L else None), comp.get(strand[i], "N")
            if prev in code and cur in code:
                wx[cov, q, code[prev] | (code[cur] << 2)] += (1, e)
    qt, ct, xt = orc.bqsr_gather(b, h, orc.BqsrRef([refb], [iv]), None, 500)
    assert qt.sum() > 3000 and np.array_equal(qt, wq)
    assert np.array_equal(ct, wc)
    assert np.array_equal(xt, wx)


def test_bqsr_apply_covariates_against_a_second_restatement():  # filters/bqsr.go:936-1005
    """ApplyBQSR's per-base keys written a second time: every base with quality >= 6 of a read with a known read group is replaced
    by the memo value of (read group, quality, cycle, context) - cycle and context taken on the whole read (no clipping in this
    pass), context with low-quality tails and N masked, on either strand, second-of-pair cycles negative."""
    from elprep_amd.batch import Header
    from tests.common import dataset
    cfg, b, h, refs, sites = dataset("tiny", 1500, 5, 0.03)
    flags = orc.mark_duplicates(b, h)
    qt, ct, xt = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), flags, 500)
    fo = orc.BqsrFinal(qt, ct, xt, 500)
    got = fo.apply(b, h, 0)
    _, quantized = fo.quantize(0)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    want = b.qual.copy()
    changed = 0
    for i in range(b.n):
        from elprep_amd.batch import NIL16
        if b.rgid[i] == NIL16:
            continue
        cov = int(h.rg_cov[b.rgid[i]])
        if qt[cov, :, 0].sum() == 0:
            continue
        o = int(b.qual_off[i]); L = int(b.qual_off[i + 1]) - o
        if L == 0:
            continue
        qual = b.qual[o:o + L]; seq = b.seq_of(i)
        f = int(b.flag[i]); revd = bool(f & 0x10); last = bool(f & 0x80)
        good = np.nonzero(qual > 2)[0]
        strand = [seq[k] if len(good) and good[0] <= k <= good[-1] else "N" for k in range(L)] if len(good) else None
        for k in range(L):
            q = int(qual[k])
            if q < 6:
                continue
            cyc = (L - k) if revd else (k + 1)
            if last:
                cyc = -cyc
            key = -1
            if strand is not None:
                if not revd:
                    prev, cur = (strand[k - 1] if k >= 1 else None), strand[k]
                else:
                    prev, cur = (comp.get(strand[k + 1], "N") if k + 1 < L else None), comp.get(strand[k], "N")
                if prev in code and cur in code:
                    key = 2 | (code[prev] << 4) | (code[cur] << 6)
            want[o + k] = fo.recal_qual(cov, q, cyc, key, quantized, None)
            changed += want[o + k] != q
    assert changed > 1000 and np.array_equal(got, want)


# ---- the clipping chain of filters/utils.go written a second time (used by two tests below)
_READ = set("MIS=X"); _REF = set("MDN=X")

def _c_read_coord(cig, soft_start, ref_index):  # :267-321
    goal = ref_index - soft_start
    if goal < 0:
        return -1, False
    read_bases = ref_bases = 0
    falls = before = False
    index = 0
    fall_or = False
    while ref_bases != goal and index < len(cig):
        op, ln = cig[index]; index += 1
        shift = 0
        if op in _REF or op == "S":
            shift = ln if ref_bases + ln < goal else goal - ref_bases
            ref_bases += shift
        if ref_bases != goal:
            read_bases += ln if op in _READ else 0
        else:
            if shift >= ln and index == len(cig):
                return -1, False
            nxt = None
            if shift < ln:
                falls = op in "DN"
            else:
                nxt = cig[index]; index += 1
                if nxt[0] == "I":
                    read_bases += nxt[1]
                    if index == len(cig):
                        return -1, False
                    nxt = cig[index]; index += 1
                before = nxt[0] in "DN"
            fall_or = before or falls
            if not fall_or:
                read_bases += shift if op in _READ else 0
            elif before:
                read_bases += (shift - 1) if op in _READ else 0
            elif falls:
                read_bases -= 1
    if ref_bases != goal:
        return -1, False
    return read_bases, fall_or

def _c_get_read_coord(cig, soft_start, ref_index, right):  # :330-349
    rb, fall_or = _c_read_coord(cig, soft_start, ref_index)
    if rb == -1:
        return -1, False
    if right and fall_or:
        rb += 1
    if not right and rb == 0:
        for op, ln in cig:
            if op == "I":
                rb = min(ln, sum(l for o, l in cig if o in _READ) - 1)
                break
            if op in "HS":
                continue
            break
    return rb, True

def _c_shift_of(op, ln, n):  # calculateHardClippingAlignmentShift
    return -n if op == "I" else (ln if op in "DN" else 0)

def _c_clean(cig):  # :472-512
    total = idx = 0
    while idx < len(cig) and cig[idx][0] in "HDN":
        total += cig[idx][1]; idx += 1
    if idx > 0:
        cig = [("H", total)] + cig[idx:]
    total = 0
    idx = len(cig) - 1
    while idx >= 0 and cig[idx][0] in "HDN":
        total += cig[idx][1]; idx -= 1
    if idx < len(cig) - 1:
        cig = cig[:idx + 1] + [("H", total)]
    return cig

def _c_hard_clip_cigar(cig, start, stop):  # :407-470
    index = 0
    total = stop - start + 1
    ashift = 0
    new = []
    if start == 0:
        k = 0
        while k < len(cig) and cig[k][0] == "H":
            total += cig[k][1]; k += 1
        while index <= stop and k < len(cig):
            op, ln = cig[k]
            shift = ln if op in _READ else 0
            if index + shift == stop + 1:
                ashift += _c_shift_of(op, ln, ln)
                new.append(("H", total + ashift))
            elif index + shift > stop + 1:
                ashift += _c_shift_of(op, ln, stop - index + 1)
                new += [("H", total + ashift), (op, ln - (stop - index + 1))]
            index += shift
            ashift += _c_shift_of(op, ln, shift)
            k += 1
        new += cig[k:]
    else:
        k = 0
        while index < start and k < len(cig):
            op, ln = cig[k]
            shift = ln if op in _READ else 0
            if index + shift < start:
                new.append((op, ln))
            else:
                ashift += _c_shift_of(op, ln, ln - (start - index))
                if op == "H":
                    total += start - index
                else:
                    new.append((op, start - index))
            index += shift
            k += 1
        while k < len(cig):
            op, ln = cig[k]
            ashift += _c_shift_of(op, ln, ln)
            if op == "H":
                total += ln
            k += 1
        new.append(("H", total + ashift))
    return _c_clean(new)

def _c_hs_offset(cig):
    size = i = 0
    while i < len(cig) and cig[i][0] == "H":
        size += cig[i][1]; i += 1
    while i < len(cig) and cig[i][0] == "S":
        size += cig[i][1]; i += 1
    return size

class _ClipPanic(Exception):
    pass

def _c_clip(rec):
    cig = [(op, ln) for ln, op in rec["ops"]]
    st = dict(a=0, n=rec["L"], pos=rec["pos"], cig=cig)
    def soft_start():  # :224-234
        start = st["pos"]
        for op, ln in st["cig"]:
            if op == "S":
                start -= ln
            elif op != "H":
                break
        return start

    end = lambda: st["pos"] + sum(l for o, l in st["cig"] if o in _REF) - 1

    def hard_clip(start, stop):
        new = _c_hard_clip_cigar(st["cig"], start, stop)
        new_len = st["n"] - (stop - start + 1)
        copy_start = stop + 1 if start == 0 else 0
        old = st["cig"]
        st["a"] += copy_start; st["n"] = new_len; st["cig"] = new
        if start == 0:
            st["pos"] += _c_hs_offset(new) - _c_hs_offset(old)

    f = rec["flag"]; revd = bool(f & 0x10)
    if rec["tlen"] != 0 and f & 0x1 and not (f & 0x8 or rec["next_refid"] < 0 or rec["pnext"] == 0) and revd != bool(f & 0x20):
        aln_end = end()
        well = aln_end > rec["pnext"] if revd else st["pos"] <= rec["pnext"] + rec["tlen"]
        if well:
            boundary = rec["pnext"] - 1 if revd else st["pos"] + abs(rec["tlen"])
            if st["pos"] <= boundary <= aln_end:
                if revd:
                    stop, ok = _c_get_read_coord(st["cig"], soft_start(), boundary, False)
                    if not ok:
                        raise _ClipPanic()
                    hard_clip(0, stop)
                else:
                    start, ok = _c_get_read_coord(st["cig"], soft_start(), boundary, True)
                    if not ok:
                        raise _ClipPanic()
                    hard_clip(start, st["n"] - 1)
    if st["n"] == 0:
        return st
    read_index = 0; cut_left = cut_right = -1; right_tail = False
    for op, ln in st["cig"]:
        if op == "S":
            if right_tail:
                cut_right = read_index
            else:
                cut_left = read_index + ln - 1
        elif op != "H":
            right_tail = True
        read_index += ln if op in _READ else 0
    if cut_right >= 0:
        hard_clip(cut_right, st["n"] - 1)
    if cut_left >= 0:
        hard_clip(0, cut_left)
    return st



def test_clipping_chain_against_a_second_restatement():  # filters/utils.go:149-262 (adaptor), 267-349 (read coordinate), 374-512 (hard clip)
    """hardClipAdaptorSequence + hardClipSoftClippedBases written a second time in Python (computeReadCoordinateForReferenceCoordinate,
    getReadCoordinateForReferenceCoordinate, hardClipCigar, cleanHardClippedCigar, hardClip with the POS shift), on random CIGARs
    with clips, insertions and deletions and random mate geometry: base window, new POS and new CIGAR must be the oracle's; where the
    reference would panic the oracle must say so."""
    rng = np.random.default_rng(77)
    Panic = _ClipPanic
    clip = _c_clip
    checked = panics = 0
    for trial in range(1500):
        ops = []
        if rng.random() < 0.15:
            ops.append((int(rng.integers(1, 4)), "H"))
        if rng.random() < 0.35:
            ops.append((int(rng.integers(1, 6)), "S"))
        ops.append((int(rng.integers(2, 12)), "M"))
        for _ in range(int(rng.integers(0, 3))):
            ops.append((int(rng.integers(1, 4)), "ID"[rng.integers(0, 2)]))
            ops.append((int(rng.integers(1, 10)), "M"))
        if rng.random() < 0.35:
            ops.append((int(rng.integers(1, 6)), "S"))
        if rng.random() < 0.15:
            ops.append((int(rng.integers(1, 4)), "H"))
        L = sum(l for l, o in ops if o in _READ)
        span = sum(l for l, o in ops if o in _REF)
        pos = int(rng.integers(50, 100))
        revd = rng.random() < 0.5
        paired = rng.random() < 0.8
        flag = (0x1 | (0x40 if rng.random() < 0.5 else 0x80) | (0x20 if not revd else 0) if paired else 0) | (0x10 if revd else 0)
        pnext = int(rng.integers(pos - 10, pos + span + 10))
        tlen = int(rng.integers(-40, 41))
        rec = dict(ops=ops, L=L, pos=pos, flag=flag, pnext=pnext, tlen=tlen, next_refid=0)
        b = batch_from_records([dict(qname="q", flag=flag, refid=0, pos=pos, cigar="".join("%d%s" % x for x in ops), mapq=60, next_refid=0, pnext=pnext,
                                     tlen=tlen, seq="A" * L, qual=[30] * L, rgid=0)])
        try:
            st = clip(rec)
        except Panic:
            with pytest.raises(RuntimeError):
                orc.clip_for_bqsr(b, 0)
            panics += 1
            continue
        a, e, npos, cg = orc.clip_for_bqsr(b, 0)
        got = (a, e, npos, [("MIDNSHP=X"[int(c) & 15], int(c) >> 4) for c in cg])
        if st["n"] == 0:
            assert e - a == 0, (trial, ops)
        else:
            assert got == (st["a"], st["a"] + st["n"], st["pos"], st["cig"]), (trial, ops, flag, pnext, tlen, got, st)
        checked += 1
    assert checked > 1200


def test_bqsr_gather_against_a_full_second_restatement():  # filters/bqsr.go:225-551 + the clipping chain above
    """Recalibrate as a whole, written a second time in Python: recalibrateAln, adaptor and soft-clip hard clipping, the skip slice
    from the known sites (read coordinates by the 'left' rule on the clipped CIGAR), SNP events over the clipped CIGAR, cycle and
    context on the clipped read - on a synthetic batch with indels, clips, adaptor read-through, supplementary records, duplicates."""
    from tests.common import dataset
    from elprep_amd.batch import NIL16
    cfg, b, h, refs, sites = dataset("tiny", 2500, 7, 0.03)
    flags = orc.mark_duplicates(b, h)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    b2i = {ord("a"): 1, ord("A"): 1, ord("*"): 1, ord("c"): 2, ord("C"): 2, ord("g"): 3, ord("G"): 3, ord("t"): 4, ord("T"): 4}
    wq = np.zeros((h.n_cov, 94, 2), np.int64); wc = np.zeros((h.n_cov, 94, 1001, 2), np.int64); wx = np.zeros((h.n_cov, 94, 16, 2), np.int64)
    used = clipped_reads = 0
    for i in range(b.n):
        f = int(flags[i]); mq = int(b.mapq[i]); r = int(b.refid[i]); pos = int(b.pos[i])
        ops = [(int(c) >> 4, "MIDNSHP=X"[int(c) & 15]) for c in b.cigar[int(b.cigar_off[i]):int(b.cigar_off[i + 1])]]
        o = int(b.qual_off[i]); L = int(b.qual_off[i + 1]) - o
        if b.has_sr[i] or not (0 < mq < 255) or f & (0x100 | 0x400 | 0x200) or f & 0x4 or r < 0 or pos <= 0 or L == 0 or L != int(b.l_seq[i]):
            continue
        if b.rgid[i] == NIL16 or pos > int(h.ref_len[r]) or any(op == "N" for _, op in ops) or sum(l for l, op in ops if op in _READ) != L:
            continue
        rec = dict(ops=ops, L=L, pos=pos, flag=f, pnext=int(b.pnext[i]), tlen=int(b.tlen[i]), next_refid=int(b.next_refid[i]))
        st = _c_clip(rec)
        n, a, cig, cpos = st["n"], st["a"], st["cig"], st["pos"]
        if n == 0:
            continue
        clipped_reads += n != L
        seq = b.seq_of(i)[a:a + n]; qual = b.qual[o + a:o + a + n]
        revd, last = bool(f & 0x10), bool(f & 0x80)
        # soft start / soft end of the clipped record (no soft clips are left, hard clips do not count)
        ss, se = cpos, cpos + sum(l for op, l in cig if op in _REF) - 1
        skip = [False] * n
        for s, e in orc.intersect(sites[r], ss, se):
            fs, ok = _c_get_read_coord(cig, ss, int(s), False)
            if not ok or fs < 0:
                fs = 0
            fe, ok = _c_get_read_coord(cig, ss, int(e), False)
            if not ok or fe > n - 1:
                fe = n - 1
            for k in range(fs, fe + 1):
                skip[k] = True
        snp = [0] * n
        ri, rj = 0, cpos - 1
        for op, ln in cig:
            if op in "M=X":
                for _ in range(ln):
                    if b2i.get(ord(seq[ri]), 0) != b2i.get(int(refs[r][rj]), 0):
                        snp[ri] = 1
                    ri += 1; rj += 1
            elif op in "DN":
                rj += ln
            elif op in "IS":
                ri += ln
        good = np.nonzero(qual > 2)[0]
        strand = [seq[k] if len(good) and good[0] <= k <= good[-1] else "N" for k in range(n)] if len(good) else None
        rof = -1 if last else 1
        cf = rof + (n - 1) * rof * (1 if revd else 0); inc = (-1 if revd else 1) * rof
        cov = int(h.rg_cov[b.rgid[i]])
        used += 1
        for k in range(n):
            q = int(qual[k])
            if skip[k] or seq[k] not in code or q < 6:
                continue
            e = snp[k]
            wq[cov, q] += (1, e); wc[cov, q, 500 + cf + k * inc] += (1, e)
            if strand is None:
                continue
            if not revd:
                prev, cur = (strand[k - 1] if k >= 1 else None), strand[k]
            else:
                prev, cur = (comp.get(strand[k + 1], "N") if k + 1 < n else None), comp.get(strand[k], "N")
            if prev in code and cur in code:
                wx[cov, q, code[prev] | (code[cur] << 2)] += (1, e)
    qt, ct, xt = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), flags, 500)
    assert used > 3000 and clipped_reads > 100 and wq[..., 1].sum() > 1000
    assert np.array_equal(qt, wq)
    assert np.array_equal(ct, wc)
    assert np.array_equal(xt, wx)


def test_quantization_against_a_second_restatement():  # filters/bqsr.go:743-897
    """initializeQuantizedQualityScores written a second time in Python: the observation count per empirical quality, the
    minimal-penalty merging of neighbouring intervals down to `levels`, the score of every interval."""
    import math
    from tests.common import dataset
    cfg, b, h, refs, sites = dataset("tiny", 3000, 2, 0.02)
    flags = orc.mark_duplicates(b, h)
    qt, ct, xt = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), flags, 500)
    fo = orc.BqsrFinal(qt, ct, xt, 500)
    qe, _, _ = fo.empirical()
    qmap = [0] * 94
    for cov in range(qt.shape[0]):
        for q in range(94):
            if qt[cov, q, 0] > 0:
                qmap[int(qe[cov, q])] += int(qt[cov, q, 0])

    def quantize(levels):
        if levels == 0:
            return [0] * 94, list(range(94))
        iv = []
        for i, nobs in enumerate(qmap):
            er = math.pow(10, i / -10)
            iv.append(dict(next=i + 1 if i + 1 < 94 else -1, er=er, nobs=nobs, leaf=nobs, nerr=int(float(nobs) * er)))
        rate = lambda nobs, nerr: 0.0 if nobs == 0 else float(nerr + 1) / float(nobs + 1)

        def leaf_penalty(k, g):
            return 0.0 if k <= 6 else abs(math.log10(iv[k]["er"]) - math.log10(g)) * float(iv[k]["leaf"])

        def merge_penalty(i, j):
            g = rate(iv[i]["nobs"] + iv[j]["nobs"], iv[i]["nerr"] + iv[j]["nerr"])
            if g == 0:
                return 0.0
            kend = iv[j]["next"] if iv[j]["next"] >= 0 else 94
            return sum(leaf_penalty(k, g) for k in range(i, j)) + sum(leaf_penalty(k, g) for k in range(j, kend))

        n = 94
        while n > levels:
            i, j = 0, iv[0]["next"]
            if j < 0:
                break
            min_i, best = i, merge_penalty(i, j)
            while True:
                i = j; j = iv[i]["next"]
                if j < 0:
                    break
                p = merge_penalty(i, j)
                if p < best:
                    min_i, best = i, p
            a, c = iv[min_i], iv[iv[min_i]["next"]]
            a["next"], a["nobs"], a["nerr"] = c["next"], a["nobs"] + c["nobs"], a["nerr"] + c["nerr"]
            n -= 1
        scores = [0] * 94
        i = 0
        while i >= 0:
            x = iv[i]
            leaf = (i == 93) if x["next"] < 0 else (x["next"] == i + 1)
            if leaf:
                sc = i
            else:
                p = rate(x["nobs"], x["nerr"])
                sc = 93 if p == 0.0 else max(min(int(math.floor(-10 * math.log10(p) + 0.5)), 93), 1)
            for k in range(i, x["next"] if x["next"] >= 0 else 94):
                scores[k] = sc
            i = x["next"]
        return qmap, scores

    for levels in (0, 2, 4, 8, 16, 64):
        counts, quantized = fo.quantize(levels)
        wc, ws = quantize(levels)
        assert quantized.tolist() == ws, levels
        if levels:
            assert counts.tolist() == wc
    assert len(set(fo.quantize(4)[1].tolist())) <= 4 + 7  # four levels above the qualities the penalty ignores


def test_static_quantized_scores_against_a_second_restatement():  # filters/bqsr.go:710-742
    """initializeStaticQuantizedScores (--sqq) written a second time, including its quirk: prevProb / prevQual are updated INSIDE the
    loop over i, so from the second i of a gap on every quality maps to the upper bin."""
    import math
    prob = lambda q: 1 - math.pow(10, float(q) / -10)

    def static(quals):
        s = [0] * 254
        for i in range(6):
            s[i] = i
        if len(quals) == 1:
            for i in range(6, 254):
                s[i] = quals[0]
            return s
        quals = sorted(quals)
        prev_q = 6
        prev_p = prob(prev_q)
        for nq in quals:
            i = prev_q
            while i < nq:  # the Go loop bound nextQual is fixed, its start prevQual was read once
                nxt = prob(nq); ip = prob(i)
                s[i] = nq if ip - prev_p > nxt - ip else prev_q
                prev_p = nxt; prev_q = nq
                i += 1
        for i in range(prev_q, 254):
            s[i] = prev_q
        return s

    for quals in ([20], [10, 20, 30], [10, 20, 30, 40], [7, 8, 50], [25], [6, 93], [15, 16]):
        assert orc.static_quantized_scores(quals).tolist() == static(list(quals)), quals
