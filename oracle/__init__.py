"""CPU oracle (TEST INFRASTRUCTURE — see oracle/orc.h).

ctypes wrapper around oracle/liboracle.so, the plain-C restatement of the reference's algorithms for the
hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product package elprep_amd never does.  PARITY UNPINNED (no reference tests/fixtures for this path; the
reference is not buildable here) — see DESIGN.md.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional, Sequence

import numpy as np

from elprep_amd.batch import Batch, Header, CBatch, CHeader

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None

NCTR = 7
NQUAL = 94
NCTX = 16
CTR_NAMES = ["UnpairedReadsExamined", "ReadPairsExamined", "SecondaryOrSupplementaryReads", "UnmappedReads",
             "UnpairedReadDuplicates", "ReadPairDuplicates", "ReadPairOpticalDuplicates"]


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_sort.c", "orc_markdup.c", "orc_bqsr.c", "orc_bam.c", "orc_gomath.c", "orc.h")]
    if force or not os.path.exists(path) or any(os.path.getmtime(s) > os.path.getmtime(path) for s in srcs if os.path.exists(s)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return path


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        L = _LIB
        L.orc_mod_flag.restype = C.c_uint16
        L.orc_mod_flag.argtypes = [C.c_uint16]
        L.orc_phred_score.restype = C.c_int32
        L.orc_unclipped_position.restype = C.c_int32
        L.orc_unclipped_position.argtypes = [C.c_int32, C.c_uint16, C.c_void_p, C.c_uint32]
        L.orc_bqsr_finalize.restype = C.c_void_p
        L.orc_bqsr_report.restype = C.c_void_p
        L.orc_go_log10.restype = C.c_double
        L.orc_go_log10.argtypes = [C.c_double]
        L.orc_go_pow10.restype = C.c_double
        L.orc_go_pow10.argtypes = [C.c_double]
        L.orc_go_log.restype = C.c_double
        L.orc_go_log.argtypes = [C.c_double]
        L.orc_go_exp.restype = C.c_double
        L.orc_go_exp.argtypes = [C.c_double]
        L.orc_go_lgamma.restype = C.c_double
        L.orc_go_lgamma.argtypes = [C.c_double]
        L.orc_bayesian_estimate.restype = C.c_uint8
        L.orc_bayesian_estimate.argtypes = [C.c_int64, C.c_int64, C.c_double]
        L.orc_estimate_library_size.restype = C.c_int64
        L.orc_estimate_library_size.argtypes = [C.c_int64, C.c_int64]
        L.orc_flatten.restype = C.c_size_t
        L.orc_num_sorted.restype = C.c_uint64
        L.orc_bam_encode.restype = C.c_size_t
        L.orc_bqsr_recal_qual.restype = C.c_uint8
    return _LIB


def _p(a: Optional[np.ndarray]):
    return C.c_void_p(a.ctypes.data) if a is not None and a.size else C.c_void_p(0)


def _check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc} (reference would panic)")


# ---------------- sort ----------------
def mod_flag(flag: int) -> int:
    return int(lib().orc_mod_flag(flag))


def sort_coordinate(b: Batch, flags: Optional[np.ndarray] = None) -> np.ndarray:
    """flags: the FLAG column the sort sees.  In `elprep filter` the sort is the Finalize step of the phase-1 pipeline
    (sam/filter-pipeline.go:116), i.e. it runs behind the filters: with --mark-duplicates the comparator's modFlag(FLAG) tie-break
    (sam/sam-types.go:447-452) sees the duplicate bits.  Pass the flags after MarkDuplicates to get that order."""
    perm = np.empty(b.n, dtype=np.uint32)
    if flags is not None:
        cols = {name: getattr(b, name) for name in b.__dataclass_fields__}
        cols["flag"] = np.ascontiguousarray(flags, dtype=np.uint16)
        b = Batch(**cols)
    s = b.as_struct()
    _check(lib().orc_sort_coordinate(C.byref(s), _p(perm)), "sort_coordinate")
    return perm


def num_sorted(b: Batch) -> int:
    """records without the sr tag: perm[:num_sorted] of sort_coordinate is the sorted output (RemoveOptionalReads drops the rest)"""
    s = b.as_struct()
    return int(lib().orc_num_sorted(C.byref(s)))


def coordinate_less(b: Batch, i: int, j: int) -> bool:
    s = b.as_struct()
    return bool(lib().orc_coordinate_less(C.byref(s), C.c_uint64(i), C.c_uint64(j)))


# ---------------- mark duplicates ----------------
def phred_score(qual: np.ndarray) -> int:
    q = np.ascontiguousarray(qual, dtype=np.uint8)
    inv = C.c_int(0)
    r = lib().orc_phred_score(_p(q), C.c_uint32(q.size), C.byref(inv))
    if inv.value:
        raise ValueError("Invalid QUAL character")
    return int(r)


def unclipped_position(pos: int, flag: int, cigar: np.ndarray) -> int:
    c = np.ascontiguousarray(cigar, dtype=np.uint32)
    return int(lib().orc_unclipped_position(pos, flag, _p(c), c.size))


def mark_duplicates(b: Batch, h: Header, with_adapted: bool = False):
    flags = np.empty(b.n, dtype=np.uint16)
    upos = np.zeros(b.n, dtype=np.int32) if with_adapted else None
    score = np.zeros(b.n, dtype=np.int32) if with_adapted else None
    s, hs = b.as_struct(), h.as_struct()
    _check(lib().orc_mark_duplicates(C.byref(s), C.byref(hs), _p(flags), _p(upos), _p(score)), "mark_duplicates")
    return (flags, upos, score) if with_adapted else flags


def dup_metrics(b: Batch, h: Header, perm: Optional[np.ndarray], pixel_dist: int = 100, hist_len: int = 0):
    flags = np.empty(b.n, dtype=np.uint16)
    ctr = np.zeros((h.n_lib + 1, NCTR), dtype=np.int64)
    hist = np.zeros((h.n_lib + 1, 3, hist_len), dtype=np.int64) if hist_len else None
    pp = None if perm is None else np.ascontiguousarray(perm, dtype=np.uint32)
    s, hs = b.as_struct(), h.as_struct()
    _check(lib().orc_dup_metrics(C.byref(s), C.byref(hs), _p(pp), C.c_int(pixel_dist), _p(flags), _p(ctr), _p(hist), C.c_int(hist_len)),
           "dup_metrics")
    return flags, ctr, hist


def tile_info(qname: bytes):
    t, x, y = C.c_int64(), C.c_int64(), C.c_int64()
    buf = np.frombuffer(qname, dtype=np.uint8)
    lib().orc_tile_info(_p(buf), C.c_uint32(len(qname)), C.byref(t), C.byref(x), C.byref(y))
    return t.value, x.value, y.value


def estimate_library_size(n_pairs: int, n_unique: int) -> int:
    return int(lib().orc_estimate_library_size(n_pairs, n_unique))


# ---------------- intervals ----------------
def _iv(arr) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(arr, dtype=np.int32).reshape(-1, 2))


def flatten(intervals) -> np.ndarray:
    iv = _iv(intervals).copy()
    n = lib().orc_flatten(_p(iv), C.c_size_t(iv.shape[0]))
    return iv[:n]


def sort_by_start(intervals) -> np.ndarray:
    iv = _iv(intervals).copy()
    lib().orc_sort_by_start(_p(iv), C.c_size_t(iv.shape[0]))
    return iv


def overlap(intervals, start: int, end: int) -> bool:
    iv = _iv(intervals)
    return bool(lib().orc_overlap(_p(iv), C.c_size_t(iv.shape[0]), C.c_int32(start), C.c_int32(end)))


def intersect(intervals, start: int, end: int) -> np.ndarray:
    iv = _iv(intervals)
    lo, hi = C.c_size_t(), C.c_size_t()
    lib().orc_intersect(_p(iv), C.c_size_t(iv.shape[0]), C.c_int32(start), C.c_int32(end), C.byref(lo), C.byref(hi))
    return iv[lo.value:hi.value]


# ---------------- BQSR ----------------
class BqsrRef:
    """Reference bases + flattened known sites per refid (fasta.MappedFasta + BaseRecalibrator.knownIntervals)."""

    def __init__(self, ref_seqs: Sequence[np.ndarray], sites: Sequence[np.ndarray]):
        self.ref_seqs = [np.ascontiguousarray(r, dtype=np.uint8) for r in ref_seqs]
        self.sites = [_iv(s) for s in sites]
        n = len(self.ref_seqs)
        self._seq_ptrs = (C.c_void_p * n)(*[r.ctypes.data if r.size else 0 for r in self.ref_seqs])
        self._seq_len = np.asarray([r.size for r in self.ref_seqs], dtype=np.int64)
        self._site_ptrs = (C.c_void_p * n)(*[s.ctypes.data if s.size else 0 for s in self.sites])
        self._n_sites = np.asarray([s.shape[0] for s in self.sites], dtype=np.int64)

        class S(C.Structure):
            _fields_ = [("ref_seq", C.c_void_p), ("ref_seq_len", C.c_void_p), ("sites", C.c_void_p), ("n_sites", C.c_void_p)]

        self.struct = S(C.cast(self._seq_ptrs, C.c_void_p), self._seq_len.ctypes.data, C.cast(self._site_ptrs, C.c_void_p),
                        self._n_sites.ctypes.data)


def bqsr_gather(b: Batch, h: Header, ref: BqsrRef, flags: Optional[np.ndarray] = None, max_cycle: int = 500):
    ncyc = 2 * max_cycle + 1
    qt = np.zeros((h.n_cov, NQUAL, 2), dtype=np.int64)
    ct = np.zeros((h.n_cov, NQUAL, ncyc, 2), dtype=np.int64)
    xt = np.zeros((h.n_cov, NQUAL, NCTX, 2), dtype=np.int64)
    f = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint16)
    s, hs = b.as_struct(), h.as_struct()
    _check(lib().orc_bqsr_gather(C.byref(s), C.byref(hs), C.byref(ref.struct), _p(f), C.c_int(max_cycle), _p(qt), _p(ct), _p(xt)),
           "bqsr_gather")
    return qt, ct, xt


def recalibrate_aln(b: Batch, h: Header, i: int, flags: Optional[np.ndarray] = None) -> bool:
    f = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint16)
    s, hs = b.as_struct(), h.as_struct()
    return bool(lib().orc_recalibrate_aln(C.byref(s), C.byref(hs), _p(f), C.c_uint64(i)))


def clip_for_bqsr(b: Batch, i: int):
    """-> (a, b, new_pos, cigar ops uint32[]) after hardClipAdaptorSequence + hardClipSoftClippedBases"""
    a, e, npos = C.c_int(), C.c_int(), C.c_int32()
    cap = int(b.cigar_off[i + 1] - b.cigar_off[i]) + 8
    out = np.zeros(cap, dtype=np.uint32)
    s = b.as_struct()
    n = lib().orc_clip_for_bqsr(C.byref(s), C.c_uint64(i), C.byref(a), C.byref(e), C.byref(npos), _p(out), C.c_int(cap))
    if n < 0:
        raise RuntimeError("reference would panic in hardClipByReferenceCoordinates")
    return a.value, e.value, npos.value, out[:n]


def read_coordinate_for_reference_coordinate(cigar: np.ndarray, soft_start: int, ref_index: int, right_tail: bool):
    c = np.ascontiguousarray(cigar, dtype=np.uint32)
    ok = C.c_int()
    r = lib().orc_read_coordinate_for_reference_coordinate(_p(c), C.c_uint32(c.size), C.c_int(soft_start), C.c_int(ref_index),
                                                           C.c_int(1 if right_tail else 0), C.byref(ok))
    return int(r), bool(ok.value)


def context_with(bases: bytes) -> np.ndarray:
    buf = np.frombuffer(bases, dtype=np.uint8)
    out = np.zeros(max(len(bases), 1), dtype=np.int32)
    lib().orc_context_with(_p(buf), C.c_int(len(bases)), _p(out))
    return out[:len(bases)]


def cycle(flag: int, l_seq: int, index: int) -> int:
    return int(lib().orc_cycle(C.c_uint16(flag), C.c_int(l_seq), C.c_int(index)))


class BqsrFinal:
    """FinalizeBQSRTables result (float64 host math)."""

    def __init__(self, qt, ct, xt, max_cycle: int = 500):
        self.n_cov = qt.shape[0]
        self.max_cycle = max_cycle
        self.qt, self.ct, self.xt = (np.ascontiguousarray(t, dtype=np.int64) for t in (qt, ct, xt))
        self.h = C.c_void_p(lib().orc_bqsr_finalize(C.c_int(self.n_cov), C.c_int(max_cycle), _p(self.qt), _p(self.ct), _p(self.xt)))

    def __del__(self):
        try:
            lib().orc_bqsr_final_free(self.h)
        except Exception:
            pass

    def empirical(self):
        ncyc = 2 * self.max_cycle + 1
        q = np.zeros((self.n_cov, NQUAL), dtype=np.uint8)
        c = np.zeros((self.n_cov, NQUAL, ncyc), dtype=np.uint8)
        x = np.zeros((self.n_cov, NQUAL, NCTX), dtype=np.uint8)
        lib().orc_bqsr_final_empirical(self.h, _p(q), _p(c), _p(x))
        return q, c, x

    def combined(self):
        rep = np.zeros(self.n_cov, dtype=np.float64)
        emp = np.zeros(self.n_cov, dtype=np.uint8)
        obs = np.zeros(self.n_cov, dtype=np.int64)
        mism = np.zeros(self.n_cov, dtype=np.int64)
        present = np.zeros(self.n_cov, dtype=np.uint8)
        lib().orc_bqsr_final_combined(self.h, _p(rep), _p(emp), _p(obs), _p(mism), _p(present))
        return rep, emp, obs, mism, present

    def quantize(self, levels: int):
        counts = np.zeros(94, dtype=np.int64)
        scores = np.zeros(94, dtype=np.uint8)
        lib().orc_bqsr_quantize(self.h, C.c_int(levels), _p(counts), _p(scores))
        return counts, scores

    def recal_qual(self, cov: int, qual: int, cyc: int, ctx_key: int, quantized: np.ndarray, static_q: Optional[np.ndarray] = None) -> int:
        q = np.ascontiguousarray(quantized, dtype=np.uint8)
        s = None if static_q is None else np.ascontiguousarray(static_q, dtype=np.uint8)
        return int(lib().orc_bqsr_recal_qual(self.h, C.c_int(cov), C.c_int(qual), C.c_int(cyc), C.c_int(ctx_key), _p(q), _p(s)))

    def report(self, cov_names: Sequence[str], prefix: str = "GATK") -> str:
        arr = (C.c_char_p * len(cov_names))(*[n.encode() for n in cov_names])
        p = lib().orc_bqsr_report(self.h, arr, prefix.encode())
        s = C.string_at(p).decode()
        lib().orc_free(C.c_void_p(p))
        return s

    def apply(self, b: Batch, h: Header, quantize_levels: int = 0, sqq: Sequence[int] = ()) -> np.ndarray:
        out = np.empty_like(b.qual)
        sq = np.asarray(list(sqq), dtype=np.uint8)
        s, hs = b.as_struct(), h.as_struct()
        _check(lib().orc_bqsr_apply(C.byref(s), C.byref(hs), self.h, C.c_int(quantize_levels), _p(sq), C.c_int(sq.size),
                                    C.c_int(self.max_cycle), _p(out)), "bqsr_apply")
        return out


def static_quantized_scores(quals: Sequence[int]) -> np.ndarray:
    q = np.asarray(list(quals), dtype=np.uint8)
    out = np.zeros(254, dtype=np.uint8)
    lib().orc_static_quantized_scores(_p(q), C.c_int(q.size), _p(out))
    return out


def go_log10(x: float) -> float:
    return float(lib().orc_go_log10(x))


def go_pow10(y: float) -> float:
    return float(lib().orc_go_pow10(y))


def go_log(x: float) -> float:
    """math.Log as Go computes it on amd64 (orc_gomath.c)"""
    return float(lib().orc_go_log(x))


def go_exp(x: float) -> float:
    """Go's math.Exp, the pure-Go function (oracle/orc_gomath.c)"""
    return float(lib().orc_go_exp(x))


def go_lgamma(x: float) -> float:
    return float(lib().orc_go_lgamma(x))


def gomath_selfcheck() -> int:
    return int(lib().orc_gomath_selfcheck())


def bayesian_estimate(obs: int, mism: int, prior: float) -> int:
    return int(lib().orc_bayesian_estimate(obs, mism, prior))


def contig_groups(ref_len: np.ndarray, contig_group_size: int = 0):
    rl = np.ascontiguousarray(ref_len, dtype=np.int32)
    out = np.zeros(rl.size, dtype=np.int32)
    n = lib().orc_contig_groups(_p(rl), C.c_int(rl.size), C.c_int(contig_group_size), _p(out))
    return int(n), out


# ---------------- BAM records ----------------
def bam_encode(b: Batch, rg_ids: Sequence[str], order: Optional[np.ndarray] = None, flags: Optional[np.ndarray] = None,
               qual: Optional[np.ndarray] = None, normalize_tags: bool = False, out: Optional[np.ndarray] = None) -> np.ndarray:
    """formatBamAlignment over the records `order` (default: all, input order); see orc_bam.c for the optional fields.
    out: a uint8 array to write into (e.g. over pinned memory); default: a fresh array."""
    arr = (C.c_char_p * max(len(rg_ids), 1))(*[s.encode() for s in rg_ids])
    o = None if order is None else np.ascontiguousarray(order, dtype=np.uint32)
    f = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint16)
    q = None if qual is None else np.ascontiguousarray(qual, dtype=np.uint8)
    s = b.as_struct()
    args = (C.byref(s), arr, _p(o), C.c_uint64(0 if o is None else o.size), _p(f), _p(q), C.c_int(1 if normalize_tags else 0))
    n = lib().orc_bam_encode(*args, C.c_void_p(0))
    if out is None:
        out = np.empty(n, dtype=np.uint8)
    assert out.size >= n
    lib().orc_bam_encode(*args, C.c_void_p(out.ctypes.data))
    return out[:n]


# ---------------- all host cores (bench.py's CPU baseline; same results as the sequential functions) ----------------
def sort_coordinate_mt(b: Batch, flags: Optional[np.ndarray] = None, n_threads: int = 0) -> np.ndarray:
    perm = np.empty(b.n, dtype=np.uint32)
    if flags is not None:
        cols = {name: getattr(b, name) for name in b.__dataclass_fields__}
        cols["flag"] = np.ascontiguousarray(flags, dtype=np.uint16)
        b = Batch(**cols)
    s = b.as_struct()
    _check(lib().orc_sort_coordinate_mt(C.byref(s), _p(perm), C.c_int(n_threads)), "sort_coordinate_mt")
    return perm


def dup_metrics_mt(b: Batch, h: Header, perm: Optional[np.ndarray], pixel_dist: int = 100, n_threads: int = 0):
    flags = np.empty(b.n, dtype=np.uint16)
    ctr = np.zeros((h.n_lib + 1, NCTR), dtype=np.int64)
    pp = None if perm is None else np.ascontiguousarray(perm, dtype=np.uint32)
    s, hs = b.as_struct(), h.as_struct()
    _check(lib().orc_dup_metrics_mt(C.byref(s), C.byref(hs), _p(pp), C.c_int(pixel_dist), _p(flags), _p(ctr), C.c_int(n_threads)), "dup_metrics_mt")
    return flags, ctr


def mark_duplicates_mt(b: Batch, h: Header, n_threads: int = 0) -> np.ndarray:
    """MarkDuplicates alone (the phase-1 filter, no metrics pass) on all cores: the flags of mark_duplicates()"""
    flags = np.empty(b.n, dtype=np.uint16)
    s, hs = b.as_struct(), h.as_struct()
    _check(lib().orc_dup_metrics_mt(C.byref(s), C.byref(hs), C.c_void_p(0), C.c_int(0), _p(flags), C.c_void_p(0), C.c_int(n_threads)), "mark_duplicates_mt")
    return flags


def bqsr_gather_mt(b: Batch, h: Header, ref: BqsrRef, flags: Optional[np.ndarray] = None, max_cycle: int = 500, n_threads: int = 0):
    ncyc = 2 * max_cycle + 1
    qt = np.zeros((h.n_cov, NQUAL, 2), dtype=np.int64)
    ct = np.zeros((h.n_cov, NQUAL, ncyc, 2), dtype=np.int64)
    xt = np.zeros((h.n_cov, NQUAL, NCTX, 2), dtype=np.int64)
    f = None if flags is None else np.ascontiguousarray(flags, dtype=np.uint16)
    s, hs = b.as_struct(), h.as_struct()
    _check(lib().orc_bqsr_gather_mt(C.byref(s), C.byref(hs), C.byref(ref.struct), _p(f), C.c_int(max_cycle), _p(qt), _p(ct), _p(xt), C.c_int(n_threads)),
           "bqsr_gather_mt")
    return qt, ct, xt


def bqsr_apply_mt(fin: "BqsrFinal", b: Batch, h: Header, quantize_levels: int = 0, sqq: Sequence[int] = (), n_threads: int = 0) -> np.ndarray:
    out = np.empty_like(b.qual)
    sq = np.asarray(list(sqq), dtype=np.uint8)
    s, hs = b.as_struct(), h.as_struct()
    _check(lib().orc_bqsr_apply_mt(C.byref(s), C.byref(hs), fin.h, C.c_int(quantize_levels), _p(sq), C.c_int(sq.size), C.c_int(fin.max_cycle), _p(out),
                                   C.c_int(n_threads)), "bqsr_apply_mt")
    return out


def bam_offsets(b: Batch, rg_ids: Sequence[str], normalize_tags: bool = False) -> np.ndarray:
    """byte offsets (n + 1) of the records of bam_encode(b, rg_ids) - what a BAM reader knows about the stream it hands over"""
    arr = (C.c_char_p * max(len(rg_ids), 1))(*[s.encode() for s in rg_ids])
    off = np.empty(b.n + 1, dtype=np.uint64)
    s = b.as_struct()
    lib().orc_bam_offsets(C.byref(s), arr, C.c_void_p(0), C.c_uint64(0), C.c_int(1 if normalize_tags else 0), _p(off))
    return off
