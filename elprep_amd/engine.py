"""Python face of the two C ABIs (harness for tests and bench.py; the product is the shared objects).

`Engine` wraps one elp_ctx (one GPU).  Method names follow the reference operators they stand in for:

    Engine.sort_coordinate   ~ sam.By(sam.CoordinateLess).ParallelStableSort        (sam/sam-types.go:639)
    Engine.mark_duplicates   ~ filters.MarkDuplicates(alsoOpticals)                 (filters/mark-duplicates.go:406)
    Engine.dup_metrics       ~ filters.MarkOpticalDuplicates(reads, pairs, dist)    (filters/mark-optical-duplicates.go:469)
    Engine.recalibrate       ~ (*BaseRecalibrator).Recalibrate(reads, maxCycle)     (filters/bqsr.go:467)
    BqsrTables.finalize      ~ (*BaseRecalibratorTables).FinalizeBQSRTables()       (filters/bqsr.go:677)
    Engine.apply_bqsr        ~ (*BaseRecalibratorTables).ApplyBQSR(...)             (filters/bqsr.go:936)
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .batch import Batch, Header

NCTR, NQUAL, NCTX = 7, 94, 16


class ElpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[elp {code}] {msg}")
        self.code = code


def _vp(a: Optional[np.ndarray]):
    return C.c_void_p(a.ctypes.data) if a is not None and a.size else C.c_void_p(0)


class Engine:
    """One GPU context holding the staged column store."""

    def __init__(self, header: Header, device: int = 0, flat_abi: bool = False, tuning: Optional[Dict[str, int]] = None):
        """flat_abi: use the entry points that take every array as its own argument (what a cgo binding calls) instead of the
        struct forms - same library code behind both.  tuning: elp_set_tuning pairs; the harness also takes them from the environment
        variable ELP_TUNE="key=value,key=value" (A/B sessions and tests; the library itself reads no environment)"""
        self.L = _lib.hip()
        h = C.c_void_p()
        rc = self.L.elp_create(device, C.byref(h))
        if rc != 0 or not h.value:
            raise ElpError(rc, "elp_create failed: no usable gfx950 device (there is no CPU fallback)")
        self.h = h
        self.header = header
        self.flat_abi = flat_abi
        if flat_abi:
            rl = np.ascontiguousarray(header.ref_len, dtype=np.int32)
            lib_ = np.ascontiguousarray(header.rg_lib, dtype=np.uint16)
            cov = np.ascontiguousarray(header.rg_cov, dtype=np.uint16)
            self._check(self.L.elp_set_header_columns(self.h, header.n_ref, _vp(rl), header.n_rg, _vp(lib_), _vp(cov), header.n_lib, header.n_cov))
        else:
            hs = header.as_struct()
            self._check(self.L.elp_set_header(self.h, C.byref(hs)))
        import os
        for kv in filter(None, os.environ.get("ELP_TUNE", "").split(",")):
            k, v = kv.split("=")
            self.set_tuning(k.strip(), int(v))
        for k, v in (tuning or {}).items():
            self.set_tuning(k, v)

    def set_tuning(self, key: str, value: int):
        """elp_set_tuning: pin a kernel choice of this context (include/elprep_hip.h lists the keys)"""
        self._check(self.L.elp_set_tuning(self.h, key.encode(), int(value)))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            bad = self.L.elp_debug_check_guards()  # (ELP_DEBUG_GUARD=1: a kernel wrote past the end of a device buffer; 0 otherwise)
            self.L.elp_destroy(self.h)
            self.h = C.c_void_p()
        else:
            bad = 0
        for ptr in getattr(self, "_pinned", []):  # (before the guard's report: an exception must not leave page-locked memory behind)
            self.L.elp_pinned_free(ptr)
        self._pinned = []
        if bad:
            raise RuntimeError(f"ELP_DEBUG_GUARD: {bad} device buffer(s) were written past their end (sizes on stderr)")

    def pinned_zeros(self, shape, dtype) -> np.ndarray:
        """a zeroed array in page-locked host memory (elp_pinned_alloc; freed by close(), which __del__ also calls: the array dies with the
        engine - copy what must outlive it): the buffers the tables and the LUT travel through, so that their copies run at the PCIe rate
        instead of through the runtime's staging of pageable memory"""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        ptr = self.L.elp_pinned_alloc(max(nbytes, 8))
        if not ptr:
            raise MemoryError("elp_pinned_alloc(%d)" % nbytes)
        if not hasattr(self, "_pinned"):
            self._pinned = []
        self._pinned.append(ptr)
        a = np.frombuffer((C.c_uint8 * max(nbytes, 8)).from_address(ptr), dtype=np.uint8)[:nbytes].view(dtype).reshape(shape)
        a[...] = 0
        return a

    def pinned_release(self, a: np.ndarray):
        """frees an array pinned_zeros handed out (nobody may touch it afterwards)"""
        ptr = a.ctypes.data
        if ptr in getattr(self, "_pinned", []):
            self._pinned.remove(ptr)
            self.L.elp_pinned_free(ptr)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            raise ElpError(rc, (self.L.elp_last_error(self.h) or b"").decode())

    # ---- staging
    @property
    def n(self) -> int:
        return int(self.L.elp_num_records(self.h))

    @property
    def n_sorted(self) -> int:
        """records that survive RemoveOptionalReads (no sr tag): the first n_sorted entries of the permutation are the output"""
        return int(self.L.elp_num_sorted(self.h))

    def reserve(self, n, qname_bytes, cigar_ops, seq_bytes, qual_bytes):
        self._check(self.L.elp_reserve(self.h, n, qname_bytes, cigar_ops, seq_bytes, qual_bytes))

    def stage(self, b: Batch):
        if self.flat_abi:
            return self.stage_columns(b)
        s = b.as_struct()
        self._check(self.L.elp_stage(self.h, C.byref(s)))

    _STAGE_COLS = ("refid", "pos", "next_refid", "pnext", "tlen", "flag", "mapq", "rgid", "has_sr", "l_seq", "qname_off", "qname", "cigar_off", "cigar",
                   "seq_off", "seq4", "qual_off", "qual", "split")

    def stage_columns(self, b: Batch):
        """elp_stage_columns: every column as an argument of its own (the form a cgo binding uses with Go slices)"""
        self._check(self.L.elp_stage_columns(self.h, b.n, *[_vp(getattr(b, k)) for k in self._STAGE_COLS]))

    def stage_pointers(self, n: int, ptrs: dict):
        """elp_stage_columns on raw addresses (columns that live in page-locked memory: bench.py's PCIe-inclusive measurement)"""
        self._check(self.L.elp_stage_columns(self.h, n, *[C.c_void_p(ptrs.get(k, 0)) for k in self._STAGE_COLS]))

    def reset(self):
        self._check(self.L.elp_reset(self.h))

    def snapshot(self):
        self._check(self.L.elp_snapshot(self.h))

    def rollback(self):
        self._check(self.L.elp_rollback(self.h))

    def sync(self):
        self._check(self.L.elp_sync(self.h))

    # ---- BAM in / BAM out (include/elprep_hip.h: sam/bam-files.go on the device)
    def set_read_group_ids(self, ids: Sequence[str]):
        if self.flat_abi:
            enc = [s.encode() for s in ids]
            cat = np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8)
            off = np.cumsum([0] + [len(e) for e in enc]).astype(np.uint32)
            return self._check(self.L.elp_set_read_group_ids_flat(self.h, _vp(cat), _vp(off)))
        arr = (C.c_char_p * max(len(ids), 1))(*[s.encode() for s in ids])
        self._check(self.L.elp_set_read_group_ids(self.h, C.cast(arr, C.c_void_p)))

    def stage_bam(self, data: np.ndarray, split_id: int = 0, rec_off: Optional[np.ndarray] = None):
        """data: uint8 array of whole inflated BAM alignment records (page-locked memory is read in place by the DMA engine);
        rec_off: uint64 offsets of the records' block_size fields + the total (optional: else the block_size chain is walked)"""
        d = np.ascontiguousarray(data, dtype=np.uint8)
        if rec_off is None:
            self._check(self.L.elp_stage_bam(self.h, _vp(d), d.size, C.c_void_p(0), 0, split_id))
        else:
            ro = np.ascontiguousarray(rec_off, dtype=np.uint64)
            self._check(self.L.elp_stage_bam(self.h, _vp(d), d.size, _vp(ro), ro.size - 1, split_id))

    def emit_sorted_bam(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        n = C.c_uint64()
        if out is None:
            self._check(self.L.elp_emit_sorted_bam(self.h, C.c_void_p(0), 0, C.byref(n)))
            out = np.empty(int(n.value), dtype=np.uint8)
        self._check(self.L.elp_emit_sorted_bam(self.h, _vp(out), out.size, C.byref(n)))
        return out[:int(n.value)]

    def emit_sorted_bgzf(self) -> np.ndarray:
        """the sorted records as BGZF blocks (compressed on the device, CRC-32 on the device): elp_emit_sorted_bgzf; the size query gives an
        upper bound (the stored form), the call the actual size"""
        n = C.c_uint64()
        self._check(self.L.elp_emit_sorted_bgzf(self.h, C.c_void_p(0), 0, C.byref(n)))
        out = np.empty(int(n.value), dtype=np.uint8)
        self._check(self.L.elp_emit_sorted_bgzf(self.h, _vp(out), out.size, C.byref(n)))
        return out[:int(n.value)]

    def stage_bgzf(self, bgzf: np.ndarray, first_record: int = 0, split_id: int = 0):
        """whole BGZF blocks of a BAM file: inflated, checked and cut into records on the device (elp_stage_bgzf)"""
        bgzf = np.ascontiguousarray(bgzf, dtype=np.uint8)
        self._check(self.L.elp_stage_bgzf(self.h, _vp(bgzf), bgzf.size, first_record, split_id))

    def emit_merged_bam(self, spread: "Engine") -> np.ndarray:
        """this context's (group splits) and `spread`'s sorted outputs as one BAM record stream in the merge's order (elp_emit_merged_bam)"""
        n = C.c_uint64()
        self._check(self.L.elp_emit_merged_bam(self.h, spread.h, C.c_void_p(0), 0, C.byref(n)))
        out = np.empty(int(n.value), dtype=np.uint8)
        self._check(self.L.elp_emit_merged_bam(self.h, spread.h, _vp(out), out.size, C.byref(n)))
        return out[:int(n.value)]

    # ---- fused predicates, split / merge bookkeeping (include/elprep_hip.h)
    def filter_records(self, remove_unmapped=False, remove_unmapped_strict=False, min_mapq=0, remove_non_exact=False, remove_duplicates=False,
                       regions=None) -> int:
        """filters/simple-filters.go predicates in one pass; regions: per refid an int32 [k][2] array (sorted, flattened) or None.
        -> number of records rejected by this call"""
        class P(C.Structure):
            _fields_ = [(k, C.c_int) for k in ("remove_unmapped", "remove_unmapped_strict", "min_mapq", "remove_non_exact", "remove_duplicates", "use_regions")] + \
                       [("regions", C.c_void_p), ("n_regions", C.c_void_p)]
        p = P(int(remove_unmapped), int(remove_unmapped_strict), int(min_mapq), int(remove_non_exact), int(remove_duplicates), 0, None, None)
        if self.flat_abi:
            n = C.c_uint64()
            reg, off = C.c_void_p(0), C.c_void_p(0)
            if regions is not None:
                arrs = [np.asarray(r, dtype=np.int32).reshape(-1, 2) for r in regions]
                cat = np.ascontiguousarray(np.concatenate(arrs, axis=0)) if arrs else np.zeros((0, 2), np.int32)
                if cat.size == 0:
                    cat = np.zeros((1, 2), np.int32)  # a non-NULL pointer switches the region test on
                offs = np.cumsum([0] + [a.shape[0] for a in arrs]).astype(np.int64)
                reg, off = C.c_void_p(cat.ctypes.data), C.c_void_p(offs.ctypes.data)
            self._check(self.L.elp_filter_records_flat(self.h, int(remove_unmapped), int(remove_unmapped_strict), int(min_mapq), int(remove_non_exact),
                                                       int(remove_duplicates), reg, off, C.byref(n)))
            return int(n.value)
        keep = []
        if regions is not None:
            arrs = [np.ascontiguousarray(np.asarray(r, dtype=np.int32).reshape(-1, 2)) for r in regions]
            ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data if a.size else 0 for a in arrs])
            cnts = np.asarray([a.shape[0] for a in arrs], dtype=np.int64)
            keep = [arrs, ptrs, cnts]
            p.use_regions, p.regions, p.n_regions = 1, C.cast(ptrs, C.c_void_p), cnts.ctypes.data
        n = C.c_uint64()
        self._check(self.L.elp_filter_records(self.h, C.byref(p), C.byref(n)))
        del keep
        return int(n.value)

    def copy_records_from(self, src: "Engine", idx: np.ndarray, new_split: Optional[int] = None, tag_sr=False):
        """appends src's records idx (staging indices) to this context, device to device (elp_copy_records).  tag_sr: False / True (all
        copies tagged sr) / 2 (the copies whose index has bit 31 set)"""
        ix = np.ascontiguousarray(idx, dtype=np.uint32)
        self._check(self.L.elp_copy_records(self.h, src.h, _vp(ix), ix.size, -1 if new_split is None else int(new_split), int(tag_sr)))

    def group_share(self, member: "Engine"):
        """this context uses the device group of `member` (same process and device): elp_group_share"""
        self._check(self.L.elp_group_share(self.h, member.h))
        self._group_owner = member  # (the borrowed communicator lives in `member`: it must not be collected while this context can use it)
        if getattr(member, "_p2p_cb", None) is not None:
            self._p2p_cb = member._p2p_cb  # (keeps the callback object alive as long as either context)

    def group_set_p2p(self, sendrecv):
        """point-to-point messages of a transport group: sendrecv(send_peer, send: bytes or None, recv_peer, recv_bytes) -> bytes or None
        (elp_group_set_p2p)"""
        def cb(_user, sp, sbuf, sbytes, rp, rbuf, rbytes):
            try:
                out = C.string_at(sbuf, sbytes) if sp >= 0 and sbytes else None
                got = sendrecv(sp, out, rp, rbytes)
                if rp >= 0 and rbytes:
                    if got is None or len(got) != rbytes:
                        return 2
                    C.memmove(rbuf, got, rbytes)
                return 0
            except Exception:  # noqa: BLE001 - reported through the C ABI's return code
                return 1
        self._p2p_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t)(cb)
        self._check(self.L.elp_group_set_p2p(self.h, C.cast(self._p2p_cb, C.c_void_p), C.c_void_p(0)))

    def exchange_records(self, send_peer: int, idx: Optional[np.ndarray], dst: Optional["Engine"], recv_peer: int, new_split: Optional[int] = None,
                         tag_sr=False):
        """one step of the split phase's all-to-all (elp_exchange_records): this context's records idx go to rank send_peer of the device
        group, what rank recv_peer sends in its matching call is appended to dst"""
        ix = np.ascontiguousarray(idx if idx is not None else np.zeros(0), dtype=np.uint32)
        self._check(self.L.elp_exchange_records(self.h, send_peer, _vp(ix), ix.size, -1 if new_split is None else int(new_split), int(tag_sr),
                                                dst.h if dst is not None else C.c_void_p(0), recv_peer))

    def clean_sam(self) -> int:
        """filters.CleanSam on the staged records (elp_clean_sam); returns the number of records whose CIGAR was rewritten"""
        n = C.c_uint64()
        self._check(self.L.elp_clean_sam(self.h, C.byref(n)))
        return int(n.value)

    def split_classify(self, group_of_ref: np.ndarray, n_groups: int):
        g = np.ascontiguousarray(group_of_ref, dtype=np.int32)
        split = np.empty(self.n, dtype=np.uint16)
        spread = np.empty(self.n, dtype=np.uint8)
        counts = np.zeros(n_groups + 2, dtype=np.uint64)
        self._check(self.L.elp_split_classify(self.h, _vp(g), n_groups, _vp(split), _vp(spread), _vp(counts)))
        return split, spread, counts

    def merge_spread(self, spread: "Engine") -> np.ndarray:
        slots = np.empty(spread.n_sorted, dtype=np.uint64)
        self._check(self.L.elp_merge_spread(self.h, spread.h, _vp(slots)))
        return slots

    # ---- operators
    def sort_ahead(self, on: bool = True):
        """elp_sort_ahead: a coordinate sort will follow mark duplicates - its key passes are queued on the sort lane from inside
        elp_mark_duplicates, as soon as the keys exist"""
        self._check(self.L.elp_sort_ahead(self.h, 1 if on else 0))

    def sort_coordinate(self, fetch: bool = True) -> Optional[np.ndarray]:
        self._check(self.L.elp_sort_coordinate(self.h))
        return self.permutation() if fetch else None

    def permutation(self) -> np.ndarray:
        perm = np.empty(self.n, dtype=np.uint32)
        self._check(self.L.elp_get_permutation(self.h, _vp(perm)))
        return perm

    def mark_duplicates(self, also_opticals: bool = False, fetch: bool = True) -> Optional[np.ndarray]:
        self._check(self.L.elp_mark_duplicates(self.h, 1 if also_opticals else 0))
        return self.flags() if fetch else None

    def flags(self) -> np.ndarray:
        f = np.empty(self.n, dtype=np.uint16)
        self._check(self.L.elp_get_flags(self.h, _vp(f)))
        return f

    def adapted(self) -> Tuple[np.ndarray, np.ndarray]:
        up = np.empty(self.n, dtype=np.int32)
        sc = np.empty(self.n, dtype=np.int32)
        self._check(self.L.elp_get_adapted(self.h, _vp(up), _vp(sc)))
        return up, sc

    def dup_metrics(self, pixel_dist: int = 100, hist_len: int = 0):
        """DuplicationMetrics counters [n_lib + 1][7]; with hist_len >= 2 also the three set-size histograms
        [n_lib + 1][3][hist_len] (duplicates / non-optical / optical, the last bin takes everything beyond it)."""
        ctr = np.zeros((self.header.n_lib + 1, NCTR), dtype=np.int64)
        if hist_len:
            hist = np.zeros((self.header.n_lib + 1, 3, hist_len), dtype=np.int64)
            self._check(self.L.elp_dup_metrics_hist(self.h, pixel_dist, _vp(ctr), _vp(hist), hist_len))
            return ctr, hist
        self._check(self.L.elp_dup_metrics(self.h, pixel_dist, _vp(ctr)))
        return ctr

    def set_reference(self, refid: int, bases: np.ndarray):
        b = np.ascontiguousarray(bases, dtype=np.uint8)
        self._check(self.L.elp_bqsr_set_reference(self.h, refid, _vp(b), b.size))

    def set_known_sites(self, refid: int, intervals: np.ndarray):
        iv = np.ascontiguousarray(np.asarray(intervals, dtype=np.int32).reshape(-1, 2))
        self._check(self.L.elp_bqsr_set_known_sites(self.h, refid, _vp(iv), iv.shape[0]))

    def _table_buffers(self, max_cycle: int, reuse: bool):
        """the three count tables' host arrays.  reuse=True: the engine's own page-locked arrays (elp_pinned_alloc), overwritten by the next
        call and FREED by close() - also when the engine is collected: a caller that keeps the tables past the engine copies them
        (np.array(qt)); buffers of another shape that these replace are freed here."""
        ncyc = 2 * max_cycle + 1
        nc = self.header.n_cov
        bufs = getattr(self, "_tables", None) if reuse else None
        if bufs is None or bufs[0] != (nc, max_cycle):
            if bufs is not None:
                for a in bufs[1:]:
                    self.pinned_release(a)
            mk = self.pinned_zeros if reuse else (lambda shape, dtype: np.zeros(shape, dtype=dtype))  # buffers that stay: page-locked
            bufs = ((nc, max_cycle), mk((nc, NQUAL, 2), np.int64), mk((nc, NQUAL, ncyc, 2), np.int64), mk((nc, NQUAL, NCTX, 2), np.int64))
            if reuse:
                self._tables = bufs
        return bufs

    def recalibrate(self, max_cycle: int = 500, reuse: bool = False):
        """BaseRecalibrator tables.  reuse=True hands out the engine's own arrays (overwritten by the next call): a caller that
        merges or finalizes right away saves the page faults of 6 MB of fresh memory per call."""
        bufs = self._table_buffers(max_cycle, reuse)
        _, qt, ct, xt = bufs
        self._check(self.L.elp_bqsr_gather(self.h, max_cycle, _vp(qt), _vp(ct), _vp(xt)))
        return qt, ct, xt

    # ---- device-resident tables and the device group (include/elprep_hip.h: the `sfm` merge phase's sums)
    def recalibrate_device(self, max_cycle: int = 500):
        """Recalibrate, tables stay in HBM (for tables_add / tables_allreduce); tables_fetch copies them out."""
        self._max_cycle = max_cycle
        self._check(self.L.elp_bqsr_gather_device(self.h, max_cycle))

    def tables_add(self, other: "Engine"):
        """device tables of this context += those of `other` (another split of the same rank, same GPU)"""
        self._check(self.L.elp_bqsr_tables_add(self.h, other.h))

    def tables_allreduce(self, counters: Optional[np.ndarray] = None) -> Optional[np.ndarray]:
        """ONE RCCL all-reduce (sum, int64) of the device tables over the context's group, in place in HBM; `counters` (int64,
        e.g. the duplication counters) ride behind them and come back summed.  A group of one: no-op."""
        if counters is None:
            self._check(self.L.elp_bqsr_tables_allreduce(self.h, C.c_void_p(0), 0))
            return None
        c = np.ascontiguousarray(counters, dtype=np.int64).copy()
        self._check(self.L.elp_bqsr_tables_allreduce(self.h, _vp(c), c.size))
        return c.reshape(np.shape(counters))

    def tables_fetch(self, reuse: bool = False):
        bufs = self._table_buffers(self._max_cycle, reuse)
        _, qt, ct, xt = bufs
        self._check(self.L.elp_bqsr_tables_fetch(self.h, _vp(qt), _vp(ct), _vp(xt)))
        return qt, ct, xt

    def group_init(self, rank: int, world: int, uid: Optional[bytes]):
        buf = (C.c_uint8 * 128).from_buffer_copy(uid) if uid is not None else None
        self._check(self.L.elp_group_init(self.h, rank, world, buf))

    def group_init_transport(self, rank: int, world: int, allreduce):
        """the group over the caller's transport: allreduce(values: np.ndarray[int64]) sums over the ranks in place (elp_group_init_transport)"""
        def cb(_user, ptr, n):
            try:
                allreduce(np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int64)), shape=(n,)))
                return 0
            except Exception:  # noqa: BLE001 - reported through the C ABI's return code
                return 1
        self._xport_cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t)(cb)  # kept alive with the engine
        self._check(self.L.elp_group_init_transport(self.h, rank, world, C.cast(self._xport_cb, C.c_void_p), C.c_void_p(0)))

    def allreduce_i64(self, a: np.ndarray) -> np.ndarray:
        out = np.ascontiguousarray(a, dtype=np.int64).copy()
        self._check(self.L.elp_allreduce_i64(self.h, _vp(out), out.size))
        return out.reshape(a.shape)

    def quals_counted(self) -> list:
        """the qualities that had table slots in the gather that made the device tables (elp_bqsr_quals_counted), ascending"""
        bits = np.zeros(2, np.uint64)
        self._check(self.L.elp_bqsr_quals_counted(self.h, _vp(bits)))
        return [q for q in range(NQUAL) if (int(bits[q >> 6]) >> (q & 63)) & 1]

    def tables_fetch_rows(self, quals, reuse: bool = False):
        """the device tables' rows of `quals` only (elp_bqsr_tables_fetch_rows): (q_rows, c_rows, x_rows), or None if another quality has
        observations (tables that were summed with another context's: fetch the dense ones).  reuse: the engine's own page-locked arrays"""
        max_cycle = self._max_cycle
        qs = np.ascontiguousarray(list(quals), dtype=np.uint8)
        nc, nq, ncyc = self.header.n_cov, int(qs.size), 2 * max_cycle + 1
        if reuse:
            #  the engine's own page-locked arrays hold room for EVERY quality's rows and are handed out as views of their leading
            # part: the number of qualities changes from step to step, and arrays that were freed and re-made whenever it did left views a
            # caller still held pointing at released memory; they are only re-made when the header's covariates or --max-cycle change
            bufs = getattr(self, "_row_tables", None)
            if bufs is None or bufs[0] != (nc, max_cycle):
                if bufs is not None:
                    for a in bufs[1:]:
                        self.pinned_release(a)
                bufs = ((nc, max_cycle), self.pinned_zeros((nc * NQUAL * 2,), np.int64), self.pinned_zeros((nc * NQUAL * ncyc * 2,), np.int64),
                        self.pinned_zeros((nc * NQUAL * NCTX * 2,), np.int64))
                self._row_tables = bufs
            qr, cr, xr = bufs[1][:nc * nq * 2].reshape(nc, nq, 2), bufs[2][:nc * nq * ncyc * 2].reshape(nc, nq, ncyc, 2), bufs[3][:nc * nq * NCTX * 2].reshape(nc, nq, NCTX, 2)
        else:
            qr, cr, xr = np.zeros((nc, nq, 2), np.int64), np.zeros((nc, nq, ncyc, 2), np.int64), np.zeros((nc, nq, NCTX, 2), np.int64)
        rc = self.L.elp_bqsr_tables_fetch_rows(self.h, _vp(qs), nq, _vp(qr), _vp(cr), _vp(xr))
        if rc == 1:
            return None
        self._check(rc)
        return qr, cr, xr

    def lut_upload_rows(self, quals, rows: np.ndarray, defaults: np.ndarray, cov_present: np.ndarray, max_cycle: int = 500):
        """lut_upload for the LUT in rows form (BqsrTables.build_lut_rows): expanded into the dense LUT on the device"""
        qs = np.ascontiguousarray(list(quals), dtype=np.uint8)
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        d = np.ascontiguousarray(defaults, dtype=np.uint8)
        cp = np.ascontiguousarray(cov_present, dtype=np.uint8)
        assert rows.size == self.header.n_cov * qs.size * (2 * max_cycle + 1) * 17 and d.size == self.header.n_cov * NQUAL
        self._check(self.L.elp_bqsr_lut_upload_rows(self.h, max_cycle, _vp(qs), int(qs.size), _vp(rows), _vp(d), _vp(cp)))

    def lut_upload(self, lut: np.ndarray, cov_present: np.ndarray, max_cycle: int = 500):
        """the LUT's way to the device ahead of apply_bqsr(None, None, ...): callable from the thread that built it while other calls run"""
        lut = np.ascontiguousarray(lut, dtype=np.uint8)
        cp = np.ascontiguousarray(cov_present, dtype=np.uint8)
        assert lut.size == self.header.n_cov * NQUAL * (2 * max_cycle + 1) * 17
        self._check(self.L.elp_bqsr_lut_upload(self.h, max_cycle, _vp(lut), _vp(cp)))

    def apply_bqsr(self, lut: Optional[np.ndarray], cov_present: Optional[np.ndarray], max_cycle: int = 500, fetch: bool = True) -> Optional[np.ndarray]:
        if lut is None:  # uploaded ahead (lut_upload)
            self._check(self.L.elp_bqsr_apply(self.h, max_cycle, C.c_void_p(0), C.c_void_p(0)))
            return self.qual() if fetch else None
        lut = np.ascontiguousarray(lut, dtype=np.uint8)
        cp = np.ascontiguousarray(cov_present, dtype=np.uint8)
        assert lut.size == self.header.n_cov * NQUAL * (2 * max_cycle + 1) * 17
        self._check(self.L.elp_bqsr_apply(self.h, max_cycle, _vp(lut), _vp(cp)))
        return self.qual() if fetch else None

    def qual(self) -> np.ndarray:
        q = np.empty(int(self.L.elp_num_qual_bytes(self.h)), dtype=np.uint8)
        self._check(self.L.elp_get_qual(self.h, _vp(q)))
        return q

    # ---- measurement
    def profile_enable(self, on: bool = True):
        self._check(self.L.elp_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._check(self.L.elp_profile_reset(self.h))

    def profile(self) -> Dict[str, Tuple[int, float]]:
        """-> {kernel name: (launches, total ms)} measured with HIP events on the ctx stream"""
        out = {}
        n = self.L.elp_profile_count(self.h)
        if n < 0:
            self._check(n)
        for i in range(n):
            name, cnt, ms = C.c_char_p(), C.c_uint64(), C.c_double()
            self._check(self.L.elp_profile_get(self.h, i, C.byref(name), C.byref(cnt), C.byref(ms)))
            out[name.value.decode()] = (int(cnt.value), float(ms.value))
        return out


def group_unique_id() -> bytes:
    """ncclUniqueId made by rank 0; the host passes the 128 bytes to the other ranks (any channel)."""
    buf = (C.c_uint8 * 128)()
    rc = _lib.hip().elp_group_unique_id(buf)
    if rc != 0:
        raise ElpError(rc, "elp_group_unique_id failed (RCCL not loadable?)")
    return bytes(buf)


class BqsrTables:
    """BaseRecalibratorTables on the host (float64): merge, finalize, LUT, report."""

    def __init__(self, qt: np.ndarray, ct: np.ndarray, xt: np.ndarray, max_cycle: int = 500):
        self.H = _lib.host()
        self.n_cov = int(qt.shape[0])
        self.max_cycle = max_cycle
        q, c, x = (np.ascontiguousarray(t, dtype=np.int64) for t in (qt, ct, xt))
        self.t = C.c_void_p(self.H.elp_bqsr_tables_new(self.n_cov, max_cycle, _vp(q), _vp(c), _vp(x)))
        if not self.t.value:
            raise RuntimeError("elp_bqsr_tables_new failed")

    @classmethod
    def from_rows(cls, n_cov: int, quals: Sequence[int], q_rows: np.ndarray, c_rows: np.ndarray, x_rows: np.ndarray, max_cycle: int = 500) -> "BqsrTables":
        """the tables from the rows of the qualities that can hold anything (Engine.tables_fetch_rows): q_rows [n_cov][len(quals)][2],
        c_rows [n_cov][len(quals)][2*max_cycle+1][2], x_rows [n_cov][len(quals)][16][2]; every other row is empty"""
        self = cls.__new__(cls)
        self.H = _lib.host()
        self.n_cov, self.max_cycle = int(n_cov), max_cycle
        qs = np.ascontiguousarray(list(quals), dtype=np.uint8)
        q, c, x = (np.ascontiguousarray(t, dtype=np.int64) for t in (q_rows, c_rows, x_rows))
        self.t = C.c_void_p(self.H.elp_bqsr_tables_new_rows(self.n_cov, max_cycle, _vp(qs), qs.size, _vp(q), _vp(c), _vp(x)))
        if not self.t.value:
            raise RuntimeError("elp_bqsr_tables_new_rows failed")
        return self

    def build_lut_rows(self, quals: Sequence[int], quantize_levels: int = 0, sqq: Sequence[int] = (), out=None):
        """the LUT in rows form: (rows [n_cov][len(quals)][2*max_cycle+1][17], defaults [n_cov][94], present [n_cov]) - what
        Engine.lut_upload_rows expands on the device into build_lut's dense LUT.  out: the triple of an earlier call to fill again."""
        ncyc = 2 * self.max_cycle + 1
        qs = np.ascontiguousarray(list(quals), dtype=np.uint8)
        if out is not None and out[0].shape == (self.n_cov, qs.size, ncyc, 17):
            rows, defaults, present = out
        else:
            rows = np.zeros((self.n_cov, qs.size, ncyc, 17), np.uint8)
            defaults = np.zeros((self.n_cov, NQUAL), np.uint8)
            present = np.zeros(self.n_cov, np.uint8)
        s = np.asarray(list(sqq), dtype=np.uint8)
        rc = self.H.elp_bqsr_tables_build_lut_rows(self.t, quantize_levels, _vp(s), s.size, _vp(qs), qs.size, _vp(rows), _vp(defaults), _vp(present))
        if rc != 0:
            raise RuntimeError("elp_bqsr_tables_build_lut_rows: %d (a quality outside `quals` has table entries)" % rc)
        return rows, defaults, present

    def __del__(self):
        try:
            if self.t.value:
                self.H.elp_bqsr_tables_free(self.t)
        except Exception:
            pass

    def merge(self, qt, ct, xt):
        q, c, x = (np.ascontiguousarray(t, dtype=np.int64) for t in (qt, ct, xt))
        assert self.H.elp_bqsr_tables_merge(self.t, _vp(q), _vp(c), _vp(x)) == 0

    def finalize(self):
        assert self.H.elp_bqsr_tables_finalize(self.t) == 0
        return self

    def empirical(self):
        ncyc = 2 * self.max_cycle + 1
        q = np.zeros((self.n_cov, NQUAL), np.uint8)
        c = np.zeros((self.n_cov, NQUAL, ncyc), np.uint8)
        x = np.zeros((self.n_cov, NQUAL, NCTX), np.uint8)
        assert self.H.elp_bqsr_tables_empirical(self.t, _vp(q), _vp(c), _vp(x)) == 0
        return q, c, x

    def combined(self):
        rep = np.zeros(self.n_cov, np.float64)
        emp = np.zeros(self.n_cov, np.uint8)
        obs = np.zeros(self.n_cov, np.int64)
        mism = np.zeros(self.n_cov, np.int64)
        present = np.zeros(self.n_cov, np.uint8)
        assert self.H.elp_bqsr_tables_combined(self.t, _vp(rep), _vp(emp), _vp(obs), _vp(mism), _vp(present)) == 0
        return rep, emp, obs, mism, present

    def quantize(self, levels: int):
        counts = np.zeros(94, np.int64)
        scores = np.zeros(94, np.uint8)
        assert self.H.elp_bqsr_tables_quantize(self.t, levels, _vp(counts), _vp(scores)) == 0
        return counts, scores

    def build_lut(self, quantize_levels: int = 0, sqq: Sequence[int] = (), out=None):
        """out: a (lut, present) pair of an earlier call to fill again (rows of covariates that are not present keep their bytes)."""
        ncyc = 2 * self.max_cycle + 1
        if out is not None and out[0].shape == (self.n_cov, NQUAL, ncyc, 17):
            lut, present = out
            present[:] = 0
        else:
            lut = np.zeros((self.n_cov, NQUAL, ncyc, 17), np.uint8)
            present = np.zeros(self.n_cov, np.uint8)
        s = np.asarray(list(sqq), dtype=np.uint8)
        assert self.H.elp_bqsr_tables_build_lut(self.t, quantize_levels, _vp(s), s.size, _vp(lut), _vp(present)) == 0
        return lut, present

    def report(self, cov_names: Sequence[str], prefix: str = "GATK") -> str:
        arr = (C.c_char_p * len(cov_names))(*[n.encode() for n in cov_names])
        p = self.H.elp_bqsr_tables_report(self.t, C.cast(arr, C.c_void_p), prefix.encode())
        s = C.string_at(p).decode()
        self.H.elp_host_free(C.c_void_p(p))
        return s


def dup_derived(ctr_row: np.ndarray) -> Tuple[float, int]:
    H = _lib.host()
    row = np.ascontiguousarray(ctr_row, dtype=np.int64)
    pct, ls = C.c_double(), C.c_int64()
    assert H.elp_dup_derived(_vp(row), C.byref(pct), C.byref(ls)) == 0
    return pct.value, ls.value


def dup_metrics_report(counters: np.ndarray, lib_names: Sequence[str], command_line: str = "", hist: Optional[np.ndarray] = None) -> str:
    """PrintDuplicatesMetrics; with `hist` (Engine.dup_metrics(..., hist_len)) also the "## HISTOGRAM" block."""
    H = _lib.host()
    ctr = np.ascontiguousarray(counters, dtype=np.int64)
    arr = (C.c_char_p * max(len(lib_names), 1))(*[n.encode() for n in lib_names])
    if hist is not None:
        hh = np.ascontiguousarray(hist, dtype=np.int64)
        p = H.elp_dup_metrics_report_hist(_vp(ctr), _vp(hh), int(hh.shape[2]), len(lib_names), C.cast(arr, C.c_void_p), command_line.encode())
    else:
        p = H.elp_dup_metrics_report(_vp(ctr), len(lib_names), C.cast(arr, C.c_void_p), command_line.encode())
    s = C.string_at(p).decode()
    H.elp_host_free(C.c_void_p(p))
    return s
