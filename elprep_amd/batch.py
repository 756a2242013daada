"""Column-wise (SoA) alignment batches and the header facts the hot path needs.

`Batch` is the host-side staging format handed to the C ABI (`elp_batch` in include/elprep_hip.h): the
restaged form of the reference's AoS `sam.Alignment` (sam/sam-types.go:289-331) produced by record
batching (sam/filter-pipeline.go:282-296).  `Header` carries what the reference's filters read from
`sam.Header`: @SQ lengths (AddREFID, filters/simple-filters.go:208-231; alignmentAgreesWithHeader,
filters/utils.go:130-139) and the @RG -> LB / PU maps (filters/mark-duplicates.go:413-423,
filters/bqsr.go:35-51).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

NIL16 = 0xFFFF

# SAM FLAG bits (sam/sam-types.go:484-520)
MULTIPLE, PROPER, UNMAPPED, NEXT_UNMAPPED = 0x1, 0x2, 0x4, 0x8
REVERSED, NEXT_REVERSED, FIRST, LAST = 0x10, 0x20, 0x40, 0x80
SECONDARY, QCFAILED, DUPLICATE, SUPPLEMENTARY = 0x100, 0x200, 0x400, 0x800

CIGAR_OPS = "MIDNSHP=X"  # BAM op codes (sam/bam-files.go:289)
NIBBLE_TO_BASE = "=ACMGRSVTWYHKDBN"  # sam/sam-types.go:228


class CBatch(C.Structure):
    """Binary layout shared by elp_batch (include/elprep_hip.h) and orc_batch (oracle/orc.h)."""

    _fields_ = [
        ("n", C.c_uint64),
        ("refid", C.c_void_p), ("pos", C.c_void_p), ("next_refid", C.c_void_p), ("pnext", C.c_void_p), ("tlen", C.c_void_p),
        ("flag", C.c_void_p), ("mapq", C.c_void_p), ("rgid", C.c_void_p), ("has_sr", C.c_void_p), ("l_seq", C.c_void_p),
        ("qname_off", C.c_void_p), ("qname", C.c_void_p),
        ("cigar_off", C.c_void_p), ("cigar", C.c_void_p),
        ("seq_off", C.c_void_p), ("seq4", C.c_void_p),
        ("qual_off", C.c_void_p), ("qual", C.c_void_p),
        ("split", C.c_void_p),
    ]


class CHeader(C.Structure):
    """Binary layout shared by elp_header and orc_header."""

    _fields_ = [
        ("n_ref", C.c_int32), ("ref_len", C.c_void_p),
        ("n_rg", C.c_int32), ("rg_lib", C.c_void_p), ("rg_cov", C.c_void_p),
        ("n_lib", C.c_int32), ("n_cov", C.c_int32),
    ]


_COLS = [
    ("refid", np.int32), ("pos", np.int32), ("next_refid", np.int32), ("pnext", np.int32), ("tlen", np.int32),
    ("flag", np.uint16), ("mapq", np.uint8), ("rgid", np.uint16), ("has_sr", np.uint8), ("l_seq", np.uint32),
    ("qname_off", np.uint64), ("qname", np.uint8), ("cigar_off", np.uint64), ("cigar", np.uint32),
    ("seq_off", np.uint64), ("seq4", np.uint8), ("qual_off", np.uint64), ("qual", np.uint8),
    ("split", np.uint16),
]
_FIXED = ("refid", "pos", "next_refid", "pnext", "tlen", "flag", "mapq", "rgid", "has_sr", "l_seq", "split")


def _ptr(a: np.ndarray) -> int:
    return a.ctypes.data if a.size else 0


@dataclass
class Batch:
    """n records, column-wise.  Offsets have n+1 entries; POS/PNEXT are 1-based as in SAM."""

    refid: np.ndarray
    pos: np.ndarray
    next_refid: np.ndarray
    pnext: np.ndarray
    tlen: np.ndarray
    flag: np.ndarray
    mapq: np.ndarray
    rgid: np.ndarray
    has_sr: np.ndarray
    l_seq: np.ndarray
    qname_off: np.ndarray
    qname: np.ndarray
    cigar_off: np.ndarray
    cigar: np.ndarray
    seq_off: np.ndarray
    seq4: np.ndarray
    qual_off: np.ndarray
    qual: np.ndarray
    split: Optional[np.ndarray] = None  # id of the `elprep split` file a record belongs to (default: all 0 = one filter run)

    def __post_init__(self):
        if self.split is None:
            self.split = np.zeros(np.asarray(self.refid).shape[0], dtype=np.uint16)
        for name, dt in _COLS:
            a = np.ascontiguousarray(getattr(self, name), dtype=dt)
            setattr(self, name, a)
        n = self.n
        for name in _FIXED[1:]:
            if getattr(self, name).shape[0] != n:
                raise ValueError(f"column {name} has {getattr(self, name).shape[0]} rows, expected {n}")
        for name in ("qname_off", "cigar_off", "seq_off", "qual_off"):
            if getattr(self, name).shape[0] != n + 1:
                raise ValueError(f"offset column {name} must have n+1 entries")

    @property
    def n(self) -> int:
        return int(self.refid.shape[0])

    def as_struct(self) -> CBatch:
        s = CBatch()
        s.n = self.n
        for name, _ in _COLS:
            setattr(s, name, _ptr(getattr(self, name)))
        return s

    # ---- convenience accessors (tests, host harness) ----
    def qname_of(self, i: int) -> bytes:
        return self.qname[int(self.qname_off[i]):int(self.qname_off[i + 1])].tobytes()

    def cigar_of(self, i: int) -> str:
        ops = self.cigar[int(self.cigar_off[i]):int(self.cigar_off[i + 1])]
        return "".join(f"{int(c) >> 4}{CIGAR_OPS[int(c) & 0xF]}" for c in ops) or "*"

    def qual_of(self, i: int) -> np.ndarray:
        return self.qual[int(self.qual_off[i]):int(self.qual_off[i + 1])]

    def seq_of(self, i: int) -> str:
        raw = self.seq4[int(self.seq_off[i]):int(self.seq_off[i + 1])]
        out = []
        for k in range(int(self.l_seq[i])):
            b = int(raw[k >> 1])
            out.append(NIBBLE_TO_BASE[(b & 0xF) if (k & 1) else (b >> 4)])
        return "".join(out)

    def take(self, idx) -> "Batch":
        """Gather records `idx` (any order) into a new batch (payload permutation: host work in the reference design)."""
        idx = np.asarray(idx, dtype=np.int64)
        cols = {name: getattr(self, name)[idx] for name in _FIXED}
        for off, dat in (("qname_off", "qname"), ("cigar_off", "cigar"), ("seq_off", "seq4"), ("qual_off", "qual")):
            o = getattr(self, off).astype(np.int64)
            lens = (o[1:] - o[:-1])[idx]
            new_off = np.zeros(len(idx) + 1, dtype=np.uint64)
            np.cumsum(lens, out=new_off[1:])
            src_start = o[:-1][idx]
            total = int(new_off[-1])
            # flat gather indices
            if total:
                rep = np.repeat(src_start - new_off[:-1].astype(np.int64), lens)
                gather = rep + np.arange(total, dtype=np.int64)
                cols[dat] = getattr(self, dat)[gather]
            else:
                cols[dat] = getattr(self, dat)[:0]
            cols[off] = new_off
        return Batch(**cols)

    @staticmethod
    def concat(parts: List["Batch"]) -> "Batch":
        cols = {}
        for name in _FIXED + ("qname", "cigar", "seq4", "qual"):
            cols[name] = np.concatenate([getattr(p, name) for p in parts])
        for off in ("qname_off", "cigar_off", "seq_off", "qual_off"):
            acc = [np.zeros(1, dtype=np.uint64)]
            base = np.uint64(0)
            for p in parts:
                o = getattr(p, off)
                acc.append(o[1:] + base)
                base = base + o[-1]
            cols[off] = np.concatenate(acc)
        return Batch(**cols)


@dataclass
class Header:
    """@SQ lengths and the @RG dictionaries of one run."""

    ref_len: np.ndarray                      # int32[n_ref], @SQ LN in @SQ order
    rg_lib: np.ndarray                       # uint16[n_rg]: dense library id of the RG's LB, NIL16 if none
    rg_cov: np.ndarray                       # uint16[n_rg]: dense id of the BQSR read-group covariate string (PU, else ID)
    n_lib: int = 0
    n_cov: int = 0
    ref_names: List[str] = field(default_factory=list)
    rg_ids: List[str] = field(default_factory=list)
    lib_names: List[str] = field(default_factory=list)
    cov_names: List[str] = field(default_factory=list)

    def __post_init__(self):
        self.ref_len = np.ascontiguousarray(self.ref_len, dtype=np.int32)
        self.rg_lib = np.ascontiguousarray(self.rg_lib, dtype=np.uint16)
        self.rg_cov = np.ascontiguousarray(self.rg_cov, dtype=np.uint16)
        if not self.n_lib:
            v = self.rg_lib[self.rg_lib != NIL16]
            self.n_lib = int(v.max()) + 1 if v.size else 0
        if not self.n_cov:
            self.n_cov = int(self.rg_cov.max()) + 1 if self.rg_cov.size else 0
        if not self.cov_names:
            self.cov_names = [f"cov{i}" for i in range(self.n_cov)]
        if not self.lib_names:
            self.lib_names = [f"lib{i}" for i in range(self.n_lib)]

    @property
    def n_ref(self) -> int:
        return int(self.ref_len.shape[0])

    @property
    def n_rg(self) -> int:
        return int(self.rg_lib.shape[0])

    def as_struct(self) -> CHeader:
        s = CHeader()
        s.n_ref = self.n_ref
        s.ref_len = _ptr(self.ref_len)
        s.n_rg = self.n_rg
        s.rg_lib = _ptr(self.rg_lib)
        s.rg_cov = _ptr(self.rg_cov)
        s.n_lib = self.n_lib
        s.n_cov = self.n_cov
        return s

    @staticmethod
    def from_read_groups(ref_names, ref_len, read_groups) -> "Header":
        """read_groups: list of dicts with ID and optional LB / PU (sam.Header.RG)."""
        libs: List[str] = []
        covs: List[str] = []
        rg_lib, rg_cov, ids = [], [], []
        for rg in read_groups:
            ids.append(rg["ID"])
            lb = rg.get("LB")
            if lb is None:
                rg_lib.append(NIL16)
            else:
                if lb not in libs:
                    libs.append(lb)
                rg_lib.append(libs.index(lb))
            cov = rg.get("PU", rg["ID"])
            if cov not in covs:
                covs.append(cov)
            rg_cov.append(covs.index(cov))
        return Header(ref_len=np.asarray(ref_len, dtype=np.int32), rg_lib=np.asarray(rg_lib, dtype=np.uint16),
                      rg_cov=np.asarray(rg_cov, dtype=np.uint16), n_lib=len(libs), n_cov=len(covs),
                      ref_names=list(ref_names), rg_ids=ids, lib_names=libs, cov_names=covs)


def parse_cigar(s: str) -> np.ndarray:
    """'5S95M' -> BAM-encoded uint32 ops (no merging of adjacent equal ops: BAM-parse behaviour, sam/bam-files.go:357-365)."""
    if s in ("*", ""):
        return np.zeros(0, dtype=np.uint32)
    out, num = [], ""
    for ch in s:
        if ch.isdigit():
            num += ch
        else:
            out.append((int(num) << 4) | CIGAR_OPS.index(ch.upper()))
            num = ""
    return np.asarray(out, dtype=np.uint32)


def pack_seq(seq: str) -> np.ndarray:
    n = len(seq)
    out = np.zeros((n + 1) // 2, dtype=np.uint8)
    for i, ch in enumerate(seq):
        nib = NIBBLE_TO_BASE.find(ch.upper())
        if nib < 0:
            nib = 15
        out[i >> 1] |= nib if (i & 1) else (nib << 4)
    return out


def batch_from_records(records) -> Batch:
    """Build a Batch from dict records {qname, flag, refid, pos, mapq, cigar, next_refid, pnext, tlen, seq, qual, rgid, has_sr}.
    Small-case helper for tests and the host harness."""
    n = len(records)
    cols = {k: np.zeros(n, dtype=dt) for k, dt in _COLS[:10]}
    cols["split"] = np.zeros(n, dtype=np.uint16)
    qn, cg, sq, ql = [], [], [], []
    qo, co, so, lo = [0], [0], [0], [0]
    for i, r in enumerate(records):
        cols["refid"][i] = r.get("refid", -1)
        cols["pos"][i] = r.get("pos", 0)
        cols["next_refid"][i] = r.get("next_refid", -1)
        cols["pnext"][i] = r.get("pnext", 0)
        cols["tlen"][i] = r.get("tlen", 0)
        cols["flag"][i] = r.get("flag", 0)
        cols["mapq"][i] = r.get("mapq", 0)
        rg = r.get("rgid", NIL16)
        cols["rgid"][i] = NIL16 if rg is None else rg
        cols["has_sr"][i] = 1 if r.get("has_sr") else 0
        cols["split"][i] = r.get("split", 0)
        name = r.get("qname", b"")
        name = name.encode() if isinstance(name, str) else name
        qn.append(np.frombuffer(name, dtype=np.uint8))
        qo.append(qo[-1] + len(name))
        c = r.get("cigar", "*")
        c = parse_cigar(c) if isinstance(c, str) else np.asarray(c, dtype=np.uint32)
        cg.append(c)
        co.append(co[-1] + len(c))
        seq = r.get("seq", "")
        cols["l_seq"][i] = len(seq)
        p = pack_seq(seq)
        sq.append(p)
        so.append(so[-1] + len(p))
        q = np.asarray(r.get("qual", []), dtype=np.uint8)
        ql.append(q)
        lo.append(lo[-1] + len(q))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs and sum(len(x) for x in xs) else np.zeros(0, dtype=dt)
    return Batch(qname_off=np.asarray(qo, dtype=np.uint64), qname=cat(qn, np.uint8),
                 cigar_off=np.asarray(co, dtype=np.uint64), cigar=cat(cg, np.uint32),
                 seq_off=np.asarray(so, dtype=np.uint64), seq4=cat(sq, np.uint8),
                 qual_off=np.asarray(lo, dtype=np.uint64), qual=cat(ql, np.uint8), **cols)
