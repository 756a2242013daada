"""CleanSam (filters/simple-filters.go:292-306) and softClipEndOfRead / elementStradlessClippedRead (filters/utils.go:81-119): known answers
derived by hand from the reference's statements - including the two places where its arithmetic is not what the names suggest
(`pos += endPos`, `clippedBases := ReadLengthFromCigar(cigars) + clipFrom`) - for the oracle's restatement (CPU) and, with `-m gpu`, for
elp_clean_sam through the C ABI."""
import numpy as np
import pytest

import re

from elprep_amd.batch import Header, batch_from_records
from oracle import simple_filters as sf

LN = 1000
# (CIGAR, POS, FLAG, MAPQ) -> (CIGAR, MAPQ) or None where the reference panics
CASES = [
    # End = 951 + 100 - 1 = 1050 > 1000: clipFrom = 50 -> 49; the one operation ends at 100 >= 49: 49M, then S of 100 + 49
    (("100M", 951, 0, 60), ("49M149S", 60)),
    # 10S stays (ends at 10 < 19), pos = 10; 40M ends at 50 >= 19: relative position 9 -> 9M, S of 50 + 19
    (("10S40M", 981, 0, 60), ("10S9M69S", 60)),
    # pos += endPos: 5M -> pos 5; 5M ends at 10 -> pos 15; 90M ends at 105 >= 40: relative 25 -> 25M, S of 100 + 40
    (("5M5M90M", 960, 0, 60), ("5M5M25M140S", 60)),
    # the clip position falls into an insertion (read bases, no reference bases): its relative position goes into the clip
    (("30M10I60M", 961, 0, 60), ("30M148S", 60)),
    # a deletion in front: ends where it starts (30 < 39), pos = 30 + 30 = 60; 50M: relative position -21 -> no M piece, S of 80 + 39
    (("30M20D50M", 961, 0, 60), ("30M20D119S", 60)),
    # pos runs ahead of the read (5, 15, 45): the deletion "ends" at 45 >= 39 with relative position -6: the reference panics
    (("5M5M15M10D65M", 961, 0, 60), None),
    # ends exactly at the reference's end: untouched
    (("100M", 901, 0, 60), ("100M", 60)),
    # unmapped: MAPQ 0, CIGAR as it is (even one that would hang over)
    (("100M", 951, 0x4, 37), ("100M", 0)),
    # reverse strand, hard clip in front (clipFrom = 40 -> 39): H consumes neither: ends at 0 < 39, pos stays 0; 60M ends at 60: 39M, S of 60 + 39
    (("5H60M", 961, 0x10, 13), ("5H39M99S", 13)),
    # mapped flag but RNAME '*' (refid -1): the reference's map lookup gives length 0 (simple-filters.go:300); End = 0 + 10 - 1 = 9 > 0:
    # clipFrom = 1 -> 0; the one operation ends at 10 >= 0 with relative position 0: no M piece, S of 10 + 0 (the port skipped these)
    (("10M", 0, 0, 60, -1), ("10S", 60)),
    # the same at POS 5: clipFrom = -4 -> -5, relative position -5: no M piece, S of 10 - 5 (what the reference writes)
    (("10M", 5, 0, 60, -1), ("5S", 60)),
]


def _batch(cases):
    recs = []
    for k, (rec, _) in enumerate(cases):
        cig, pos, flag, mapq = rec[:4]
        refid = rec[4] if len(rec) > 4 else 0
        rl = sum(int(n) for n, o in re.findall(r"(\d+)([MIDNSHP=X])", cig) if o in "MIS=X")
        recs.append(dict(qname="r%d" % k, flag=flag, refid=refid, pos=pos, mapq=mapq, cigar=cig, seq="A" * rl, qual=[30] * rl, rgid=0))
    return batch_from_records(recs)


def _cigars(b):
    return ["".join("%d%s" % (int(c) >> 4, "MIDNSHP=X"[int(c) & 15]) for c in b.cigar[int(b.cigar_off[i]):int(b.cigar_off[i + 1])]) for i in range(b.n)]


def test_oracle_clean_sam_known_answers():
    ok = [c for c in CASES if c[1] is not None]
    out, changed = sf.clean_sam(_batch(ok), np.array([LN], np.int32))
    assert _cigars(out) == [want[0] for _, want in ok]
    assert out.mapq.tolist() == [want[1] for _, want in ok]
    assert changed == 8
    bad = [c for c in CASES if c[1] is None]
    with pytest.raises(ValueError, match="Unexpected non-0 relative clipping position"):
        sf.clean_sam(_batch(bad), np.array([LN], np.int32))


@pytest.mark.gpu
def test_elp_clean_sam_known_answers():
    import oracle as orc
    from elprep_amd.engine import Engine, ElpError
    rg = ["rg0"]
    h = Header(ref_len=np.array([LN], np.int32), rg_lib=np.array([0], np.uint16), rg_cov=np.array([0], np.uint16))
    ok = [c for c in CASES if c[1] is not None]
    b = _batch(ok)
    e = Engine(h)
    e.set_read_group_ids(rg)
    e.stage_bam(orc.bam_encode(b, rg))
    assert e.clean_sam() == 8
    e.sort_coordinate()
    want, _ = sf.clean_sam(b, h.ref_len)
    assert _cigars(want) == [w[0] for _, w in ok]   # (the oracle's output is the hand-derived one: test above)
    perm = orc.sort_coordinate(want)
    assert e.emit_sorted_bam().tobytes() == orc.bam_encode(want, rg, order=perm[:orc.num_sorted(want)], normalize_tags=True).tobytes()
    e.close()
    e = Engine(h)
    e.stage(_batch([c for c in CASES if c[1] is None]))
    with pytest.raises(ElpError) as err:
        e.clean_sam()
    assert "Unexpected non-0 relative clipping position" in str(err.value)
    e.close()
