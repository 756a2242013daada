"""Hand-derived known-answer cases shared by the CPU (oracle) and GPU (device) tests.

Every expected value below is derived by hand from the reference source, not from running any implementation:

  sr-tagged copies     `elprep split` writes a read whose mate lies in another contig group to the spread file AND, tagged sr:i:1,
                       to its own group file (sam/split-merge.go:286-293).  In the group's `filter` run the copy goes through
                       MarkDuplicates (cmd/filter.go:773) and is then dropped by RemoveOptionalReads (filters/simple-filters.go:146-152,
                       appended at cmd/filter.go:803), so MarkOpticalDuplicates (filters/mark-optical-duplicates.go:469-502) never
                       counts it.
  DeleteOrStore        classifyPair pairs the records of one {library, QNAME} in arrival order: the first stores itself, the second
                       removes it and forms the pair, the third stores itself again ... (filters/mark-duplicates.go:336-340).
"""
import numpy as np

from elprep_amd.batch import Header, batch_from_records

Q30, Q20, Q40 = [30] * 10, [20] * 10, [40] * 10
SEQ = "ACGTACGTAC"


def header2():
    return Header(ref_len=np.array([1000, 1000], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))


def _rec(name, flag, refid, pos, next_refid=-1, pnext=0, tlen=0, qual=Q30, **kw):
    d = dict(qname=name, flag=flag, refid=refid, pos=pos, cigar="10M" if not (flag & 0x4) else "*", mapq=60, next_refid=next_refid, pnext=pnext,
             tlen=tlen, seq=SEQ, qual=qual, rgid=0)
    d.update(kw)
    return d


def sr_case():
    """Two contig groups (A = refid 0, B = refid 1).  Returns (filter-mode batch, {split name: batch}, expected counters of
    library 0 = [UnpairedReadsExamined, ReadPairsExamined, SecondaryOrSupplementary, UnmappedReads, UnpairedReadDuplicates,
    ReadPairDuplicates, ReadPairOpticalDuplicates] - the same for `filter` and for the sum over the splits of `sfm`, and the QNAMEs
    that end up flagged as duplicates per split)."""
    p1 = [_rec("p1", 99, 0, 100, 0, 300, 210), _rec("p1", 147, 0, 300, 0, 100, -210)]
    p2 = [_rec("p2", 99, 1, 100, 1, 300, 210), _rec("p2", 147, 1, 300, 1, 100, -210)]
    # two spread pairs with the same ends (A:500 forward, B:500 reverse): s2 has the lower score, so the pair s2 is a duplicate
    s1 = [_rec("s1", 97, 0, 500, 1, 500), _rec("s1", 145, 1, 500, 0, 500)]
    s2 = [_rec("s2", 97, 0, 500, 1, 500, qual=Q20), _rec("s2", 145, 1, 500, 0, 500, qual=Q20)]
    # a fragment (mate unmapped) on the fragment key of s1/s2's first ends: any pair read there makes it a duplicate
    f1 = [_rec("f1", 73, 0, 500, 0, 500), _rec("f1", 133, 0, 500, 0, 500)]
    everything = p1 + p2 + s1 + s2 + f1
    tag = lambda r: dict(r, has_sr=1)
    splits = {
        "A": p1 + [tag(s1[0]), tag(s2[0])] + f1,        # group file of contig group A: the copies of the spread reads carry sr
        "B": p2 + [tag(s1[1]), tag(s2[1])],
        "spread": s1 + s2,
    }
    #           unpaired  pairs  sec/sup  unmapped  unpaired dup  pair dup  optical
    expected = [1,        4,     0,       1,        1,            1,        0]
    dups = {"filter": {"s2", "f1"}, "A": {"f1"}, "B": set(), "spread": {"s2"}}
    return batch_from_records(everything), {k: batch_from_records(v) for k, v in splits.items()}, expected, dups


def toggling_cases():
    """Three (four) primary mapped records share the QNAME "t".  -> list of (batch, staging indices flagged as duplicates).
    Pair "a" (score 800) holds the pair key {0:100 forward, 0:300 reverse}; t0 (100, forward) + t1 (300, reverse) form a pair on the
    same key with score 400; t2 is a second first-of-pair copy at 700.
      arrival a0 a1 t0 t1 t2 : t0 stores, t1 takes it -> pair (t0, t1) loses against a: both flagged; t2 stores, alone
      arrival a0 a1 t2 t0 t1 : t2 stores, t0 takes it -> pair (t2, t0) has its own key: nothing flagged; t1 stores, alone
    and the same two outcomes when the third record is far away from the other two in staging order."""
    a = [_rec("a", 99, 0, 100, 0, 300, 210, qual=Q40), _rec("a", 147, 0, 300, 0, 100, -210, qual=Q40)]
    t0, t1, t2 = _rec("t", 99, 0, 100, 0, 300, 210, qual=Q20), _rec("t", 147, 0, 300, 0, 100, -210, qual=Q20), _rec("t", 99, 0, 700, 0, 300, -410, qual=Q20)
    filler = [_rec("x%d" % k, 0, 1, 10 + 20 * k) for k in range(40)]  # single-end reads elsewhere
    return [
        (batch_from_records(a + [t0, t1, t2]), [2, 3]),
        (batch_from_records(a + [t2, t0, t1]), []),
        (batch_from_records(a + [t0, t1] + filler + [t2]), [2, 3]),
        (batch_from_records([t2] + filler + a + [t0, t1]), []),
        # four records: (t0, t1) as before, then t2 stores and a second copy of t1 takes it: that pair has its own key
        (batch_from_records(a + [t0, t1] + filler + [t2, dict(t1)]), [2, 3]),
    ]


def sort_sees_duplicate_bits_case():
    """The coordinate sort is the Finalize step of the pipeline the MarkDuplicates filter runs in (sam/filter-pipeline.go:116;
    filters first, cmd/filter.go:773), so CoordinateLess's modFlag(FLAG) tie-break (sam/sam-types.go:447-452) sees the duplicate bits.
    Pair "n" of read group 1 (library 1) loses against pair "m" there and is flagged; pair "n" of read group 0 (library 0, same
    QNAME, same ends) is alone in its library.  Staged Y = (n, rg1) before X = (n, rg0): on (refid, POS, strand, QNAME) they tie;
    FLAG 99 < 99 | 0x400, so X sorts in front of Y although it was staged behind it.
    -> (batch, header, staging indices in sorted order, duplicate-flagged staging indices)"""
    h = Header(ref_len=np.array([1000, 1000], np.int32), rg_lib=np.array([0, 1], np.uint16), rg_cov=np.array([0, 1], np.uint16))
    y = [_rec("n", 99, 0, 100, 0, 300, 210, qual=Q20, rgid=1), _rec("n", 147, 0, 300, 0, 100, -210, qual=Q20, rgid=1)]
    x = [_rec("n", 99, 0, 100, 0, 300, 210, qual=Q20, rgid=0), _rec("n", 147, 0, 300, 0, 100, -210, qual=Q20, rgid=0)]
    m = [_rec("m", 99, 0, 100, 0, 300, 210, qual=Q40, rgid=1), _rec("m", 147, 0, 300, 0, 100, -210, qual=Q40, rgid=1)]
    # staging: y0 y1 x0 x1 m0 m1; sorted: POS 100: m0 ("m" < "n"), x0 (99), y0 (99 | 0x400); POS 300: m1, x1, y1
    return batch_from_records(y + x + m), h, [4, 2, 0, 5, 3, 1], [0, 1]


def flagged_names(b, flags):
    """QNAMEs of the records whose duplicate bit is set"""
    return {b.qname_of(i).decode() for i in range(b.n) if int(flags[i]) & 0x400}
