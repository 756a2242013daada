"""tools/synth/bam_writer.c (the generator's BAM record writer that stands in for a BAM reader in bench.py) writes records the
SAM specification's layout describes: every field is read back with struct and compared with the batch."""
import struct

import numpy as np

import oracle as orc
from tools import synth


def test_bam_records_round_trip():
    cfg = synth.config("tiny")
    b = synth.generate(cfg, 0, 300)
    h = cfg.header()
    buf, off = synth.bam_records(b, h.rg_ids)
    assert off[0] == 0 and off[-1] == buf.size == synth.bam_records_size(b, h.rg_ids)
    raw = buf.tobytes()
    for i in range(b.n):
        o = int(off[i])
        (bs, refid, pos, lrn, mapq, _bin, nc, flag, ls, nref, npos, tlen) = struct.unpack_from("<IiiBBHHHIiii", raw, o)
        assert o + 4 + bs == int(off[i + 1])
        assert (refid, pos + 1, mapq, flag, ls, nref, npos + 1, tlen) == (b.refid[i], b.pos[i], b.mapq[i], b.flag[i], b.l_seq[i], b.next_refid[i], b.pnext[i], b.tlen[i])
        q = o + 36
        name = raw[q:q + lrn - 1]
        assert name == b.qname[int(b.qname_off[i]):int(b.qname_off[i + 1])].tobytes() and raw[q + lrn - 1] == 0
        q += lrn
        cg = np.frombuffer(raw, dtype="<u4", count=nc, offset=q)
        assert np.array_equal(cg, b.cigar[int(b.cigar_off[i]):int(b.cigar_off[i + 1])])
        q += 4 * nc
        assert raw[q:q + (ls + 1) // 2] == b.seq4[int(b.seq_off[i]):int(b.seq_off[i]) + (ls + 1) // 2].tobytes()
        q += (ls + 1) // 2
        assert raw[q:q + ls] == b.qual[int(b.qual_off[i]):int(b.qual_off[i + 1])].tobytes()
        q += ls
        tags = raw[q:o + 4 + bs]
        assert tags[:3] == b"NMC"
        if b.rgid[i] != 0xFFFF:
            assert b"RGZ" + h.rg_ids[b.rgid[i]].encode() + b"\0" in tags


def test_mark_duplicates_mt_alone_equals_sequential():
    """the mark-only mode of the all-cores oracle (what bench.py's CPU leg times as phase 1) gives the sequential oracle's flags"""
    cfg = synth.config("tiny", 3)
    b = synth.generate(cfg, 0, 4000)
    h = cfg.header()
    assert np.array_equal(orc.mark_duplicates_mt(b, h, 4), orc.mark_duplicates(b, h))
