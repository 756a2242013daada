"""Synthetic 150 bp paired-end data generator (test/bench infrastructure; see synth.c and SURVEY.md §8(d))."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from elprep_amd.batch import Batch, Header

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None

# hg38 primary assembly contig lengths chr1..chr22, chrX, chrY
HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
        135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
        46709983, 50818468, 156040895, 57227415]
HG38_NAMES = [f"chr{i}" for i in range(1, 23)] + ["chrX", "chrY"]

BASE_SEED = 0xE1F5EED0


class _Cfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_ref", C.c_int32), ("ref_len", C.c_void_p), ("read_len", C.c_int32),
                ("p_dup", C.c_double), ("p_optical", C.c_double), ("p_unmapped_pair", C.c_double), ("p_mate_unmapped", C.c_double),
                ("p_supp", C.c_double), ("p_sec", C.c_double), ("p_spread", C.c_double), ("p_frag", C.c_double), ("n_lanes", C.c_int32),
                ("qual_mode", C.c_int32), ("home_lo", C.c_int32), ("home_hi", C.c_int32), ("ref_seed", C.c_uint64)]


class _Sizes(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in ("n_records", "qname_bytes", "cigar_ops", "seq_bytes", "qual_bytes")]


class _Out(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("refid", "pos", "next_refid", "pnext", "tlen", "flag", "mapq", "rgid", "has_sr", "l_seq",
                                          "qname_off", "qname", "cigar_off", "cigar", "seq_off", "seq4", "qual_off", "qual")]


def build(force: bool = False) -> str:
    path = os.path.join(_HERE, "libelprep_synth.so")
    srcs = [os.path.join(_HERE, f) for f in ("synth.c", "bam_writer.c")]
    if force or not os.path.exists(path) or any(os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(path) for src in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return path


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.synth_known_sites.restype = C.c_int64
        _LIB.synth_bam_write.restype = C.c_uint64
    return _LIB


@dataclass
class SynthConfig:
    ref_len: List[int]
    ref_names: List[str] = field(default_factory=list)
    seed: int = BASE_SEED
    read_len: int = 150
    p_dup: float = 0.10
    p_optical: float = 0.15
    p_unmapped_pair: float = 0.01
    p_mate_unmapped: float = 0.01
    p_supp: float = 0.005
    p_sec: float = 0.003
    p_spread: float = 0.02
    p_frag: float = 0.0
    n_lanes: int = 4
    qual_mode: int = 0  # 0 = binned qualities (7 values), 1 = full range (~40 values)
    home_lo: int = 0   # fragments start on contigs [home_lo, home_hi) (0, 0 = all): the reads of one contig group
    home_hi: int = 0
    ref_seed: int = 0  # seed of the reference and the known sites (0 = seed): shards with seeds of their own are reads of one genome

    def __post_init__(self):
        self._ref_len = np.asarray(self.ref_len, dtype=np.int32)
        if not self.ref_names:
            self.ref_names = [f"chr{i + 1}" for i in range(len(self.ref_len))]

    def cstruct(self) -> _Cfg:
        return _Cfg(self.seed, len(self.ref_len), self._ref_len.ctypes.data, self.read_len, self.p_dup, self.p_optical,
                    self.p_unmapped_pair, self.p_mate_unmapped, self.p_supp, self.p_sec, self.p_spread, self.p_frag, self.n_lanes,
                    self.qual_mode, self.home_lo, self.home_hi, self.ref_seed)

    def header(self) -> Header:
        half = (self.n_lanes + 1) // 2
        rgs = [{"ID": f"rg{l}", "LB": "lib1" if (l - 1) < half else "lib2", "PU": f"FC1.{l}", "SM": "s1", "PL": "illumina"}
               for l in range(1, self.n_lanes + 1)]
        return Header.from_read_groups(self.ref_names, self.ref_len, rgs)


def config(name: str, seed_index: int = 0) -> SynthConfig:
    """Named genomes of BASELINE.md §3: 'c1' (1 contig 5 Mbp), 'c2' (hg38 / 12), 'c4' (hg38), 'tiny' (3 small contigs)."""
    if name == "c1":
        return SynthConfig(ref_len=[5_000_000], ref_names=["chr1"], seed=BASE_SEED + 1)
    if name in ("c2", "c3"):
        return SynthConfig(ref_len=[l // 12 for l in HG38], ref_names=HG38_NAMES, seed=BASE_SEED + (2 if name == "c2" else 3))
    if name in ("c4", "c5"):
        return SynthConfig(ref_len=list(HG38), ref_names=HG38_NAMES, seed=BASE_SEED + (4 if name == "c4" else 5))
    if name == "tiny":
        return SynthConfig(ref_len=[60_000, 45_000, 30_000], ref_names=["chrA", "chrB", "chrC"], seed=BASE_SEED + 100 + seed_index)
    raise KeyError(name)


def generate(cfg: SynthConfig, pair_lo: int, pair_hi: int) -> Batch:
    c = cfg.cstruct()
    z = _Sizes()
    lib().synth_plan(C.byref(c), C.c_uint64(pair_lo), C.c_uint64(pair_hi), C.byref(z))
    n = z.n_records
    a = dict(
        refid=np.empty(n, np.int32), pos=np.empty(n, np.int32), next_refid=np.empty(n, np.int32), pnext=np.empty(n, np.int32),
        tlen=np.empty(n, np.int32), flag=np.empty(n, np.uint16), mapq=np.empty(n, np.uint8), rgid=np.empty(n, np.uint16),
        has_sr=np.empty(n, np.uint8), l_seq=np.empty(n, np.uint32),
        qname_off=np.empty(n + 1, np.uint64), qname=np.empty(z.qname_bytes, np.uint8),
        cigar_off=np.empty(n + 1, np.uint64), cigar=np.empty(z.cigar_ops, np.uint32),
        seq_off=np.empty(n + 1, np.uint64), seq4=np.empty(z.seq_bytes, np.uint8),
        qual_off=np.empty(n + 1, np.uint64), qual=np.empty(z.qual_bytes, np.uint8))
    o = _Out(*[a[k].ctypes.data if a[k].size else 0 for k, _ in _Out._fields_])
    rc = lib().synth_fill(C.byref(c), C.c_uint64(pair_lo), C.c_uint64(pair_hi), C.byref(o))
    if rc != 0:
        raise MemoryError("synth_fill failed")
    return Batch(**a)


def reference(cfg: SynthConfig, refid: int) -> np.ndarray:
    out = np.empty(cfg.ref_len[refid], dtype=np.uint8)
    c = cfg.cstruct()
    lib().synth_reference(C.byref(c), C.c_int(refid), C.c_void_p(out.ctypes.data))
    return out


def known_sites_raw(cfg: SynthConfig, refid: int) -> np.ndarray:
    """Raw (unsorted, possibly overlapping) known-site intervals [n,2], 1-based inclusive, as an .elsites file would list them."""
    c = cfg.cstruct()
    n = lib().synth_known_sites(C.byref(c), C.c_int(refid), C.c_void_p(0), C.c_void_p(0), C.c_int64(0))
    s = np.empty(max(n, 1), dtype=np.int32)
    e = np.empty(max(n, 1), dtype=np.int32)
    lib().synth_known_sites(C.byref(c), C.c_int(refid), C.c_void_p(s.ctypes.data), C.c_void_p(e.ctypes.data), C.c_int64(n))
    return np.stack([s[:n], e[:n]], axis=1)


def bam_records(b: Batch, rg_ids, out: Optional[np.ndarray] = None):
    """The batch as uncompressed BAM alignment records (what a BAM reader hands over after inflating BGZF blocks) + the byte offsets
    of the records (n + 1).  out: uint8 array to write into (e.g. over page-locked memory); size query: bam_records_size()."""
    arr = (C.c_char_p * max(len(rg_ids), 1))(*[s.encode() for s in rg_ids])
    st = b.as_struct()
    off = np.empty(b.n + 1, dtype=np.uint64)
    n = int(lib().synth_bam_write(C.byref(st), arr, C.c_void_p(0), C.c_void_p(off.ctypes.data)))
    if out is None:
        out = np.empty(n, dtype=np.uint8)
    assert out.size >= n
    lib().synth_bam_write(C.byref(st), arr, C.c_void_p(out.ctypes.data), C.c_void_p(0))
    return out[:n], off


def bam_records_size(b: Batch, rg_ids) -> int:
    arr = (C.c_char_p * max(len(rg_ids), 1))(*[s.encode() for s in rg_ids])
    st = b.as_struct()
    return int(lib().synth_bam_write(C.byref(st), arr, C.c_void_p(0), C.c_void_p(0)))
