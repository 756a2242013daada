"""GPU parity tests (-m gpu) on ragged, adversarial inputs for the flat per-base kernels (score, BQSR gather, BQSR apply):
reads of 1..170 bases (several reads per 16-byte chunk), CIGARs with soft/hard clips and up to four indels (more than three
reference pieces), N bases, lower-case / N reference, reads hanging over the contig end, many distinct quality values (LDS
table passes), and the report-and-retry path for quality values missing from the sampled hint.  Everything is compared
bit-exactly with the CPU oracle through the C ABI."""
import os

import numpy as np
import pytest

import oracle as orc
from elprep_amd.batch import Batch, Header, batch_from_records
from elprep_amd.engine import BqsrTables, Engine, ElpError

pytestmark = pytest.mark.gpu


def _random_case(seed, n, quals, n_cov=2, len_mix=((1, 20, 0.3), (21, 60, 0.2), (100, 170, 0.5)), ref_len=(5000, 3000)):
    rng = np.random.default_rng(seed)
    refs = []
    for L in ref_len:
        r = rng.choice(list(b"ACGT"), size=L).astype(np.uint8)
        for _ in range(3):  # N runs and lower-case stretches
            s = int(rng.integers(0, L - 40)); r[s:s + int(rng.integers(1, 30))] = ord("N")
            s = int(rng.integers(0, L - 40)); seg = r[s:s + 25]; r[s:s + 25] = np.where(seg != ord("N"), seg | 0x20, seg)
        refs.append(r)
    sites = []
    for L in ref_len:
        raw = []
        for _ in range(L // 60):
            s = int(rng.integers(1, L)); raw.append((s, s + int(rng.integers(0, 12))))
        sites.append(orc.flatten(orc.sort_by_start(np.asarray(raw, dtype=np.int32))))
    n_rg = n_cov + 1
    rgs = [{"ID": f"rg{k}", "LB": f"lib{k % 2}", "PU": f"pu{k % n_cov}"} for k in range(n_rg)]
    h = Header.from_read_groups([f"c{k}" for k in range(len(ref_len))], list(ref_len), rgs)
    quals = np.asarray(quals, dtype=np.uint8)
    recs = []
    for i in range(n):
        u = rng.random()
        acc = 0.0
        for lo, hi, p in len_mix:
            acc += p
            if u <= acc:
                break
        L = int(rng.integers(lo, hi + 1))
        # CIGAR: [H] [S] M (I|D M)* [S] [H]
        lead_s = int(rng.integers(0, min(8, L))) if rng.random() < 0.2 and L > 3 else 0
        trail_s = int(rng.integers(0, min(8, L - lead_s))) if rng.random() < 0.2 and L - lead_s > 3 else 0
        core = L - lead_s - trail_s
        n_indel = int(rng.choice([0, 0, 0, 1, 1, 2, 3, 4])) if core >= 12 else 0
        ops = []
        if rng.random() < 0.05:
            ops.append((int(rng.integers(1, 6)), "H"))
        if lead_s:
            ops.append((lead_s, "S"))
        kinds = ["I" if rng.random() < 0.5 else "D" for _ in range(n_indel)]
        ilens = [int(rng.integers(1, 4)) if k == "I" else 0 for k in kinds]
        while sum(ilens) + n_indel + 1 > core:  # not enough room: turn insertions into deletions
            k = ilens.index(max(ilens)); ilens[k] = 0; kinds[k] = "D"
        m_total = core - sum(ilens)
        cuts = np.sort(rng.choice(np.arange(1, m_total), size=n_indel, replace=False)) if n_indel else np.zeros(0, dtype=int)
        m_lens = np.diff(np.concatenate([[0], cuts, [m_total]])).astype(int)
        read_pieces = []  # M / I consume the read, D only the reference
        for k in range(n_indel):
            read_pieces.append(("M", int(m_lens[k])))
            read_pieces.append(("I", ilens[k]) if kinds[k] == "I" else ("D", int(rng.integers(1, 6))))
        read_pieces.append(("M", int(m_lens[-1])))
        for op, ln in read_pieces:
            ops.append((ln, op))
        if trail_s:
            ops.append((trail_s, "S"))
        if rng.random() < 0.05:
            ops.append((int(rng.integers(1, 6)), "H"))
        ref_span = sum(ln for op, ln in read_pieces if op in "MD")
        refid = int(rng.integers(0, len(ref_len)))
        RL = ref_len[refid]
        if rng.random() < 0.03:
            pos = int(rng.integers(max(1, RL - ref_span // 2), RL + 1))  # hangs over the contig end
        else:
            pos = int(rng.integers(1, max(2, RL - ref_span - 1)))
        # sequence
        seq = []
        for _ in range(lead_s):
            seq.append("ACGT"[int(rng.integers(0, 4))])
        j = pos - 1
        for op, ln in read_pieces:
            if op == "M":
                for _ in range(ln):
                    c = chr(refs[refid][j]).upper() if 0 <= j < RL else "A"
                    if c not in "ACGT" or rng.random() < 0.03:
                        c = "ACGT"[int(rng.integers(0, 4))]
                    if rng.random() < 0.01:
                        c = "N"
                    seq.append(c); j += 1
            elif op == "I":
                for _ in range(ln):
                    seq.append("ACGT"[int(rng.integers(0, 4))])
            else:
                j += ln
        for _ in range(trail_s):
            seq.append("ACGT"[int(rng.integers(0, 4))])
        assert len(seq) == L
        q = rng.choice(quals, size=L)
        if rng.random() < 0.3:
            q[:int(rng.integers(0, min(4, L) + 1))] = 2
        if rng.random() < 0.3:
            t = int(rng.integers(0, min(4, L) + 1))
            if t:
                q[-t:] = 2
        rev = rng.random() < 0.5
        paired = rng.random() < 0.7
        flag = 0
        pnext, tlen, next_refid = 0, 0, -1
        if paired:
            flag |= 0x1 | (0x40 if rng.random() < 0.5 else 0x80)
            next_refid = refid
            if rng.random() < 0.4 and L > 30:  # FR geometry with a short insert => adaptor clipping (filters/utils.go:149-180)
                ins = int(rng.integers(max(12, ref_span // 2), ref_span + 40))
                if rev:
                    flag |= 0x10
                    pnext = max(1, pos + ref_span - ins); tlen = -ins
                else:
                    flag |= 0x20
                    pnext = pos + max(0, ins - 20); tlen = ins
            else:
                flag |= (0x10 if rev else 0) | (0x20 if rng.random() < 0.5 else 0)
                pnext = int(rng.integers(1, RL)); tlen = 0
        else:
            flag |= 0x10 if rev else 0
        if rng.random() < 0.03:
            flag |= 0x800
        if rng.random() < 0.02:
            flag |= 0x100
        mapq = int(rng.choice([0, 255, 17, 60, 60, 60, 60, 29]))
        recs.append(dict(qname="q%06d" % i, flag=flag, refid=refid, pos=pos, mapq=mapq,
                         cigar="".join(f"{ln}{op}" for ln, op in ops), next_refid=next_refid, pnext=pnext, tlen=tlen,
                         seq="".join(seq), qual=q, rgid=int(rng.integers(0, n_rg))))
    # a few unmapped / sequence-less oddities
    recs.append(dict(qname="un", flag=4, refid=-1, pos=0, mapq=0, cigar="*", seq="ACGTN", qual=[30, 2, 40, 7, 9], rgid=0))
    recs.append(dict(qname="z0", flag=0, refid=0, pos=10, mapq=60, cigar="*", seq="", qual=[], rgid=0))
    b = batch_from_records(recs)
    return b, h, refs, sites


def _stage(b, h, chunks):
    e = Engine(h)
    cuts = np.linspace(0, b.n, chunks + 1).astype(int)
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        if hi > lo:
            e.stage(b.take(np.arange(lo, hi)))
    return e


def _check_gather_apply(b, h, refs, sites, max_cycle=500, chunks=3):
    e = _stage(b, h, chunks)
    oflags, oupos, oscore = orc.mark_duplicates(b, h, with_adapted=True)
    up, sc = e.adapted()
    assert np.array_equal(up, oupos) and np.array_equal(sc, oscore)
    flags = e.mark_duplicates()
    assert np.array_equal(flags, oflags)
    for r in range(h.n_ref):
        e.set_reference(r, refs[r])
        e.set_known_sites(r, sites[r])
    qt, ct, xt = e.recalibrate(max_cycle)
    oq, oc, ox = orc.bqsr_gather(b, h, orc.BqsrRef(refs, sites), flags, max_cycle)
    assert oq[..., 0].sum() > 0
    assert np.array_equal(ct, oc), "cycle tables differ"
    assert np.array_equal(xt, ox), "context tables differ"
    assert np.array_equal(qt, oq), "quality tables differ"
    tb = BqsrTables(qt, ct, xt, max_cycle).finalize()
    lut, present = tb.build_lut(0)
    want = orc.BqsrFinal(oq, oc, ox, max_cycle).apply(b, h, 0)
    got = e.apply_bqsr(lut, present, max_cycle)
    assert np.array_equal(got, want)
    e.close()
    e2 = _stage(b, h, 1)  # a fresh context: its quality hint has not been completed by a gather retry
    got2 = e2.apply_bqsr(lut, present, max_cycle)
    assert np.array_equal(got2, want)
    e2.close()
    return qt


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_ragged_few_quals(seed):
    b, h, refs, sites = _random_case(seed, 6000, quals=[2, 5, 6, 12, 23, 37, 41])
    _check_gather_apply(b, h, refs, sites)


def test_ragged_many_quals_multipass():
    """55 distinct qualities x 3 covariates x reads up to 170 bases do not fit one LDS table: several passes."""
    b, h, refs, sites = _random_case(11, 5000, quals=list(range(2, 57)), n_cov=3)
    _check_gather_apply(b, h, refs, sites, chunks=1)


def test_only_tiny_reads():
    """Every chunk holds several reads (more than two segments per chunk: the round loop of flat_run)."""
    b, h, refs, sites = _random_case(21, 8000, quals=[3, 8, 20, 30, 40], len_mix=((1, 6, 0.7), (7, 15, 0.3)))
    _check_gather_apply(b, h, refs, sites, chunks=2)


def test_long_reads_tile_spanning():
    b, h, refs, sites = _random_case(31, 600, quals=[2, 11, 25, 37], len_mix=((300, 480, 1.0),), ref_len=(9000, 7000))
    _check_gather_apply(b, h, refs, sites, chunks=2)


def test_many_contigs():
    """More than 256 contigs: the count kernel reads contig pointers from HBM instead of its LDS table (hg38 with alts has 3366)."""
    b, h, refs, sites = _random_case(41, 4000, quals=[2, 9, 22, 35], ref_len=tuple([700 + 13 * (k % 7) for k in range(300)]),
                                     len_mix=((1, 20, 0.2), (21, 60, 0.3), (100, 170, 0.5)))
    assert h.n_ref == 300
    _check_gather_apply(b, h, refs, sites, chunks=2)


def test_quality_hint_retry(monkeypatch):
    """With an empty sampling hint the count kernel must report the qualities it met and the host must retry: same tables."""
    b, h, refs, sites = _random_case(5, 3000, quals=[2, 6, 13, 27, 38, 64, 93])
    monkeypatch.setenv("ELP_TUNE", "qual_hint=1")
    _check_gather_apply(b, h, refs, sites)


def test_quality_hint_incomplete(monkeypatch):
    """One quality missing from the hint: the gather retries, the LDS-LUT apply kernel takes its dense-LUT fix-up path."""
    b, h, refs, sites = _random_case(6, 3000, quals=[2, 6, 13, 27, 38])
    monkeypatch.setenv("ELP_TUNE", "qual_hint_drop=27")
    _check_gather_apply(b, h, refs, sites)


def test_cycle_exceeds_max_cycle_is_an_error():
    """checkCycleCovariate (filters/bqsr.go:364-369): a counted base with |cycle| > max_cycle panics in the reference."""
    b, h, refs, sites = _random_case(7, 400, quals=[30, 35], len_mix=((100, 170, 1.0),))
    e = _stage(b, h, 1)
    for r in range(h.n_ref):
        e.set_reference(r, refs[r]); e.set_known_sites(r, sites[r])
    with pytest.raises(ElpError, match="cycle value exceeds"):
        e.recalibrate(50)
    e.close()
    # and a max_cycle that still covers every read is fine and gives the same tables as the default
    b2, h2, refs2, sites2 = _random_case(8, 400, quals=[30, 35], len_mix=((20, 40, 1.0),))
    _check_gather_apply(b2, h2, refs2, sites2, max_cycle=40)
