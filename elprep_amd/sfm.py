"""`elprep sfm` on GPUs: contig-group partition, one GPU context per split, one all-reduce.

Host-side mirror (Python, as the rest of the harness; the reference's is Go) of

    computeContigGroups / SplitFilePerChromosome   sam/split-merge.go:178-311
    the per-split `filter --bqsr-tables-only` runs and the table / metrics combination of phase 2
                                                   cmd/sfm.go:129-805, filters/print-bqsr.go:310-329,
                                                   filters/mark-optical-duplicates.go:711-731

Splits are numbered 0 = "unmapped" (RNAME '*'), 1..G = contig groups in @SQ order, G+1 = "spread" (reads whose mate maps
to another group; they are ALSO kept, tagged sr:i:1, in their own group split, where they only knock out fragments and are
ignored by BQSR).  A split is processed by exactly one rank, as the reference processes a split file by one `filter`
process; group splits owned by the same rank share one GPU context: every record carries the id of its split file
(elp_batch.split), which is part of every duplicate-marking key, so fragments, mates and pairs of different splits never meet
(two tagged copies of a spread pair whose groups land on one rank would otherwise pair up there).  The spread split always
gets its own context: it is coordinate-sorted by itself and merged into the groups' output afterwards.

The only data-path collectives are (1) the routing of the few records (spread mates, supplementary alignments, unmapped
pairs: ~3 %) that a rank produced for a split it does not own — point-to-point sends of packed batches — and (2) ONE
all-reduce (sum, int64) of the BQSR count tables concatenated with the duplication counters.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import time

import numpy as np

from .batch import Batch, Header, _COLS

NCTR = 7


# ------------------------------------------------------------------------------------------------ partition
def contig_groups(ref_len: Sequence[int], contig_group_size: int = 0) -> Tuple[np.ndarray, int]:
    """computeContigGroups (sam/split-merge.go:178-213): greedy in @SQ order, a new group starts when the running sum would
    exceed the target (default: the longest contig).  Returns (group index 1..G per refid, G)."""
    ref_len = [int(x) for x in ref_len]
    if contig_group_size <= 0:
        contig_group_size = max(ref_len) if ref_len else 0
        if contig_group_size <= 0:
            raise ValueError("no valid contig group size")  # log.Panic in the reference
    group_of_ref = np.zeros(len(ref_len), dtype=np.int32)
    cur, size = 1, 0
    for r, ln in enumerate(ref_len):
        if size > 0 and size + ln > contig_group_size:
            cur += 1
            size = 0
        group_of_ref[r] = cur
        size += ln
    return group_of_ref, (cur if ref_len else 0)


def group_ranges(group_of_ref: np.ndarray, n_groups: int) -> List[Tuple[int, int]]:
    """contig index range [lo, hi) of every group 1..G (groups are contiguous in @SQ order by construction)."""
    out = []
    for g in range(1, n_groups + 1):
        idx = np.nonzero(group_of_ref == g)[0]
        out.append((int(idx[0]), int(idx[-1]) + 1))
    return out


def split_records(b: Batch, group_of_ref: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """SplitFilePerChromosome's routing rule (sam/split-merge.go:280-293) on BAM-style fields.
    Returns (split of RNAME per record: 0 unmapped / 1..G, spread flag per record).  RNEXT == "=" is next_refid == refid
    (sam/bam-files.go:344-346); contigToGroup["*"] = "unmapped", so a mapped read with RNEXT '*' is spread."""
    g = np.where(b.refid >= 0, group_of_ref[np.clip(b.refid, 0, None)], 0).astype(np.int32) if b.n else np.zeros(0, np.int32)
    gn = np.where(b.next_refid >= 0, group_of_ref[np.clip(b.next_refid, 0, None)], 0).astype(np.int32) if b.n else np.zeros(0, np.int32)
    spread = (b.next_refid != b.refid) & (b.refid >= 0) & (gn != g)
    return g, spread


def assign_splits(weights: Sequence[float], world: int) -> np.ndarray:
    """owner rank of every split (0 .. G+1): longest-processing-time greedy on the expected read counts.  Any assignment is
    semantically free (the reference runs the splits one after the other)."""
    order = np.argsort(-np.asarray(weights, dtype=np.float64), kind="stable")
    load = np.zeros(world)
    owner = np.zeros(len(weights), dtype=np.int32)
    for s in order:
        r = int(np.argmin(load))
        owner[s] = r
        load[r] += weights[s]
    return owner


# ------------------------------------------------------------------------------------------------ packed batches
def pack_batch(b: Batch) -> np.ndarray:
    """Batch -> one uint8 buffer (header of 1 + len(_COLS) int64 lengths, then the raw columns)."""
    meta = np.zeros(1 + len(_COLS), dtype=np.int64)
    meta[0] = b.n
    parts = []
    for k, (name, dt) in enumerate(_COLS):
        a = np.ascontiguousarray(getattr(b, name), dtype=dt)
        meta[1 + k] = a.size
        parts.append(a.view(np.uint8).reshape(-1))
    return np.concatenate([meta.view(np.uint8)] + parts)


def unpack_batch(buf: np.ndarray) -> Batch:
    nmeta = 1 + len(_COLS)
    meta = buf[:8 * nmeta].view(np.int64)
    at = 8 * nmeta
    cols = {}
    for k, (name, dt) in enumerate(_COLS):
        nbytes = int(meta[1 + k]) * np.dtype(dt).itemsize
        cols[name] = buf[at:at + nbytes].view(dt).copy()
        at += nbytes
    return Batch(**cols)


def empty_batch() -> Batch:
    cols = {name: np.zeros(1 if name.endswith("_off") else 0, dtype=dt) for name, dt in _COLS}
    return Batch(**cols)


def with_sr(b: Batch, sr: np.ndarray, split: Optional[np.ndarray] = None) -> Batch:
    """copy of b whose has_sr column is OR-ed with `sr` (aln.TAGS.Set(sr, 1), sam/split-merge.go:291); `split` (optional) becomes the
    split-id column: the split file of every record, which keeps the duplicate-marking keys of the splits that share a GPU context
    apart (the reference runs one `filter` process per split file)"""
    cols = {name: getattr(b, name) for name, _ in _COLS}
    cols["has_sr"] = (b.has_sr | sr.astype(np.uint8)).astype(np.uint8)
    if split is not None:
        cols["split"] = split.astype(np.uint16)
    return Batch(**cols)


# ------------------------------------------------------------------------------------------------ exchange
class Comm:
    """torch.distributed wrapper: backend 'nccl' (= RCCL over xGMI on the GPU box) or 'gloo' (CPU tests)."""

    def __init__(self, device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.on = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank() if self.on else 0
        self.world = dist.get_world_size() if self.on else 1
        self.device = device if device is not None else torch.device("cpu")

    def allreduce_i64(self, a: np.ndarray) -> np.ndarray:
        """elementwise sum over all ranks of an int64 array: LoadAndCombineBQSRTables + LoadAndCombineDuplicateMetrics"""
        if not self.on or self.world == 1:
            return a
        t = self.torch.from_numpy(np.array(a, dtype=np.int64, copy=True)).to(self.device)  # never reduce in place into the caller's array
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy().reshape(a.shape)

    def exchange(self, outgoing: List[Optional[np.ndarray]]) -> List[np.ndarray]:
        """all-to-all of byte buffers by point-to-point messages (sizes first); outgoing[self.rank] is returned as is."""
        torch, dist = self.torch, self.dist
        world = self.world
        out = [np.zeros(0, np.uint8) if o is None else o for o in outgoing]
        if not self.on or world == 1:
            return out
        sizes = torch.tensor([o.size for o in out], dtype=torch.int64, device=self.device)
        all_sizes = [torch.zeros(world, dtype=torch.int64, device=self.device) for _ in range(world)]
        dist.all_gather(all_sizes, sizes)
        incoming = [int(all_sizes[src][self.rank].item()) for src in range(world)]
        recv = [torch.empty(incoming[src], dtype=torch.uint8, device=self.device) if src != self.rank else None for src in range(world)]
        send = [torch.from_numpy(out[dst]).to(self.device) if dst != self.rank else None for dst in range(world)]
        ops = []
        for peer in range(world):
            if peer == self.rank:
                continue
            if out[peer].size:
                ops.append(dist.P2POp(dist.isend, send[peer], peer))
            if incoming[peer]:
                ops.append(dist.P2POp(dist.irecv, recv[peer], peer))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        return [out[src] if src == self.rank else recv[src].cpu().numpy() for src in range(world)]


    def sendrecv(self, send_peer: int, data, recv_peer: int, recv_bytes: int):
        """one message each way (either may be absent): the send-receive callback of the C ABI's device group (elp_group_set_p2p) over
        torch.distributed - what ranks that share a GPU, and the gloo tests, route their records through"""
        torch, dist = self.torch, self.dist
        ops, recv = [], None
        if send_peer >= 0 and data is not None:
            ops.append(dist.P2POp(dist.isend, torch.frombuffer(bytearray(data), dtype=torch.uint8).to(self.device), send_peer))
        if recv_peer >= 0 and recv_bytes:
            recv = torch.empty(recv_bytes, dtype=torch.uint8, device=self.device)
            ops.append(dist.P2POp(dist.irecv, recv, recv_peer))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        if recv is None:
            return None
        if self.device.type == "cuda":
            torch.cuda.synchronize()
        return recv.cpu().numpy().tobytes()


@dataclass
class RankSplits:
    local: Batch    # records of the group splits (and the unmapped split) this rank owns, spread copies tagged has_sr
    spread: Batch   # the spread split, if this rank owns it (else empty)


def route(b: Batch, group_of_ref: np.ndarray, n_groups: int, owner: np.ndarray, comm: Comm) -> RankSplits:
    """Send every record this rank holds to the owner of its split(s) and collect what this rank owns."""
    g, spread = split_records(b, group_of_ref)
    tagged = with_sr(b, spread, g)
    spread_owner = int(owner[n_groups + 1])
    dest = owner[g]
    out_local, out_spread = [], []
    for r in range(comm.world):
        idx = np.nonzero(dest == r)[0]
        out_local.append(pack_batch(tagged.take(idx)) if idx.size else None)
        if r == spread_owner:
            idx = np.nonzero(spread)[0]
            out_spread.append(pack_batch(b.take(idx)) if idx.size else None)
        else:
            out_spread.append(None)
    got_local = comm.exchange(out_local)
    got_spread = comm.exchange(out_spread)
    lp = [unpack_batch(x) for x in got_local if x.size]
    sp = [unpack_batch(x) for x in got_spread if x.size]
    return RankSplits(Batch.concat(lp) if lp else empty_batch(), Batch.concat(sp) if sp else empty_batch())


def route_device(reader, b: Batch, group_of_ref: np.ndarray, n_groups: int, owner: np.ndarray, rank: int, world: int, dst_local, dst_spread,
                 stage=None) -> Tuple[int, int]:
    """The split phase through the C ABI (SplitFilePerChromosome, sam/split-merge.go:230-311): `b` is staged into the rank's READER
    context, classified on the device (elp_split_classify: split id and spread flag per record; the ids are written to the split-id
    column and travel with the records), and every record goes to the context of the rank that owns its split - device to device:
    elp_copy_records inside the rank, elp_exchange_records between ranks (a ring of world - 1 steps; RCCL send / receive over xGMI when
    every rank has its own GPU, the group's send-receive callback otherwise).  A group file's records arrive in input order, the copies
    of reads that also go to the spread file tagged sr among them (bit 31 of the index, tag_sr = 2); the spread file's records arrive in
    the spread owner's second context with split id 0.  The host sees three small arrays per batch (split ids, spread flags, counts) and
    builds index lists from them - no record passes through it.  `reader` belongs to the device group (elp_group_init /
    elp_group_share / elp_group_set_p2p).  Every rank calls this the same number of times (empty batches where it has nothing).
    stage(reader, b): how the batch gets into the reader (default: the column batch; a BAM reader hands over inflated records:
    elp_stage_bam - the raw records then travel with the columns and the destination can emit BAM).
    -> (records that arrived in dst_local, records that arrived in dst_spread)"""
    reader.reset()
    if b.n:
        (stage or (lambda e, x: e.stage(x)))(reader, b)
    if b.n:
        split, spread, _ = reader.split_classify(group_of_ref, n_groups)
        spread = spread.astype(bool)
    else:
        split, spread = np.zeros(0, np.uint16), np.zeros(0, bool)
    dest = owner[split] if b.n else np.zeros(0, np.int32)
    spread_owner = int(owner[n_groups + 1])
    n0, n1 = dst_local.n, dst_spread.n
    for s in range(world):
        sp, rp = (rank + s) % world, (rank - s + world) % world
        idx = np.nonzero(dest == sp)[0].astype(np.uint32)
        idx |= (spread[idx].astype(np.uint32) << np.uint32(31))
        sidx = np.nonzero(spread)[0].astype(np.uint32) if sp == spread_owner else np.zeros(0, np.uint32)
        if s == 0:
            if idx.size:
                dst_local.copy_records_from(reader, idx, tag_sr=2)
            if sidx.size:
                dst_spread.copy_records_from(reader, sidx, new_split=0)
        else:
            reader.exchange_records(sp, idx, dst_local, rp, tag_sr=2)
            # the spread file: only its owner receives, everybody sends to it in the step that reaches it
            send_to = sp if sp == spread_owner else -1
            recv_from = rp if rank == spread_owner else -1
            if send_to >= 0 or recv_from >= 0:
                reader.exchange_records(send_to, sidx, dst_spread if recv_from >= 0 else None, recv_from, new_split=0)
    return dst_local.n - n0, dst_spread.n - n1


def emit_merged_device(groups, spread, part, group_of_ref: np.ndarray, n_groups: int, owner: np.ndarray, rank: int, world: int) -> np.ndarray:
    """The merge phase across ranks through the C ABI (MergeSortedFilesSplitPerChromosome, sam/split-merge.go:410-576): the spread split's
    owner sends every rank the spread reads of that rank's contig groups - in the spread file's coordinate order, with the FLAG and QUAL
    columns as the path left them (elp_exchange_records; elp_copy_records for its own) - into the rank's `part` context; every rank then
    emits the merge of its group splits' sorted output and those reads as one stream of BAM records (elp_emit_merged_bam: the spread
    reads behind the group reads of their position, the rank's unmapped split last).  The output file is the ranks' streams cut at
    their group boundaries in @SQ order + the unmapped split.  `groups` and `spread` are coordinate-sorted and hold the inflated BAM
    records (elp_stage_bam); `spread` is empty on every rank but the owner; `part` is an empty context of the same device group."""
    spread_owner = int(owner[n_groups + 1])
    part.reset()
    if rank == spread_owner and spread.n:
        split, _, _ = spread.split_classify(group_of_ref, n_groups)   # the contig group of every spread read
        order = spread.permutation()[:spread.n_sorted]
        dest = owner[split[order]]
    else:
        order, dest = np.zeros(0, np.uint32), np.zeros(0, np.int32)
    for s in range(world):
        sp, rp = (rank + s) % world, (rank - s + world) % world
        idx = order[dest == sp].astype(np.uint32) if rank == spread_owner else np.zeros(0, np.uint32)
        if s == 0:
            if idx.size:
                part.copy_records_from(spread, idx, new_split=0)
        else:
            send_to = sp if rank == spread_owner else -1
            recv_from = rp if rp == spread_owner else -1
            if send_to >= 0 or recv_from >= 0:
                (spread if rank == spread_owner else part).exchange_records(send_to, idx, part if recv_from >= 0 else None, recv_from, new_split=0)
    part.sort_coordinate(fetch=False)
    return groups.emit_merged_bam(part)


# ------------------------------------------------------------------------------------------------ output order
def merge_order(group_refid: np.ndarray, group_pos: np.ndarray, spread_refid: np.ndarray, spread_pos: np.ndarray) -> np.ndarray:
    """MergeSortedFilesSplitPerChromosome (sam/split-merge.go:410-576): the group splits, coordinate-sorted and concatenated in
    group order (= @SQ order, so the concatenation is sorted by (refid, POS)), are streamed out and every read of the
    coordinate-sorted spread split is inserted in front of the first group read that is strictly greater by (refid, POS) — i.e.
    behind all group reads of the same position; spread reads left over follow, then the unmapped split (not part of this
    function).  Returns for every output slot a code: i >= 0 = the i-th group read, -(j + 1) = the j-th spread read."""
    ng, ns = int(group_refid.shape[0]), int(spread_refid.shape[0])
    kg = (group_refid.astype(np.int64) << 32) | group_pos.astype(np.int64)
    ks = (spread_refid.astype(np.int64) << 32) | spread_pos.astype(np.int64)
    # spread read j goes behind every group read with key <= its key: number of group reads in front of it
    before = np.searchsorted(kg, ks, side="right")
    out = np.empty(ng + ns, dtype=np.int64)
    slot_s = before + np.arange(ns, dtype=np.int64)          # spread reads keep their own order
    is_spread = np.zeros(ng + ns, dtype=bool)
    is_spread[slot_s] = True
    out[slot_s] = -(np.arange(ns, dtype=np.int64) + 1)
    out[~is_spread] = np.arange(ng, dtype=np.int64)
    return out


def sorted_output(b: Batch, perm: np.ndarray, flags: np.ndarray, qual: np.ndarray) -> Batch:
    """The records of a split as the output phase writes them: coordinate order (the device's permutation), duplicate flags and
    recalibrated qualities in place of the staged ones.  Payload permutation is host work (it sits next to the BAM encoder)."""
    out = Batch(**{f: getattr(b, f) for f in ("refid", "pos", "next_refid", "pnext", "tlen", "mapq", "rgid", "has_sr", "l_seq", "qname_off", "qname",
                                                  "cigar_off", "cigar", "seq_off", "seq4", "qual_off", "split")},
                flag=np.asarray(flags, dtype=b.flag.dtype), qual=np.asarray(qual, dtype=b.qual.dtype))
    return out.take(perm)


def merge_splits(groups: List[Batch], spread: Batch, unmapped: Batch) -> Batch:
    """(Test infrastructure since round 5: the product merges on the device - elp_emit_merged_bam on one GPU, emit_merged_device across
    ranks.)  MergeSortedFilesSplitPerChromosome on payloads (sam/split-merge.go:410-576): `groups` are the coordinate-sorted group
    splits in group order, `spread` the coordinate-sorted spread split, `unmapped` the unmapped split; the result is the one
    coordinate-sorted output the reference's merge phase writes (group reads with the spread reads inserted by merge_order,
    then the unmapped split)."""
    cat = Batch.concat(groups) if groups else spread.take(np.zeros(0, dtype=np.int64))
    code = merge_order(cat.refid, cat.pos, spread.refid, spread.pos)
    both = Batch.concat([cat, spread])
    idx = np.where(code >= 0, code, cat.n + (-code - 1))
    return Batch.concat([both.take(idx), unmapped])


# ------------------------------------------------------------------------------------------------ per-rank driver
class SfmRank:
    """The splits of one rank on one GPU: context 0 = its group splits, context 1 = the spread split (if owned).

    collective = "cabi": the tables stay in HBM, the two contexts are summed on the device and ONE ncclAllReduce over the device
    group of the C ABI (elp_group_init / elp_bqsr_tables_allreduce: RCCL over xGMI, no host hop) merges tables + duplication
    counters of all ranks; the group id is made by rank 0 and handed round with one torch.distributed broadcast (a Go host would
    use a file next to its split files).  collective = "torch": the tables go through the host and torch.distributed - what the CPU
    tests (gloo) and ranks that share one GPU use.  Default: "cabi" when every rank has its own GPU (nccl backend)."""

    def __init__(self, header: Header, device_ordinal: int, comm: Comm, collective: Optional[str] = None):
        import os
        from .engine import Engine, group_unique_id
        self.header, self.comm = header, comm
        self.engines = [Engine(header, device_ordinal), Engine(header, device_ordinal)]
        self.n = [0, 0]
        self.allreduce_s = []  # wall time of every all-reduce call (bench.py reports the timed steps' mean)
        self._side = None      # host thread for the spread split's context (step)
        self.reader = None     # the split phase's reader context (route)
        self._device_ordinal = device_ordinal
        if collective is None:
            collective = os.environ.get("ELP_SFM_COLLECTIVE", "cabi" if (comm.world > 1 and comm.device.type == "cuda") else "torch")
        self.collective = collective if comm.world > 1 else "none"
        if self.collective == "cabi":
            # the group id is made by rank 0 and handed round once; if any rank cannot join (no RCCL to load, ...) ALL ranks fall back
            # to the torch.distributed collective - the decision itself is a collective, so nobody is left waiting in a group
            ok = 1
            try:
                uid = group_unique_id()  # every rank: also the check that this process can load RCCL at all (rank 0's id is used)
                box = [uid if comm.rank == 0 else None]
            except Exception:
                box, ok = [None], 0
            comm.dist.broadcast_object_list(box, src=0)
            flag = comm.torch.tensor([ok if (ok and box[0] is not None) else 0], dtype=comm.torch.int64, device=comm.device)
            comm.dist.all_reduce(flag, op=comm.dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                try:
                    self.engines[0].group_init(comm.rank, comm.world, box[0])
                except Exception:
                    ok = 0
                flag = comm.torch.tensor([ok], dtype=comm.torch.int64, device=comm.device)
                comm.dist.all_reduce(flag, op=comm.dist.ReduceOp.MIN)
            if int(flag.item()) != 1:
                self.collective = "torch"

    def stage(self, which: int, b: Batch):
        if b.n:
            self.engines[which].stage(b)
            self.n[which] += b.n

    def route(self, b: Batch, group_of_ref: np.ndarray, n_groups: int, owner: np.ndarray, stage=None):
        """the split phase for one batch of input records through the C ABI (route_device): staged into the rank's reader context,
        classified on the device, delivered device to device to the contexts of the ranks that own the splits.  Every rank calls it the
        same number of times.  stage(reader, batch): how the batch enters the reader (route_device; a BAM reader: elp_stage_bam)."""
        from .engine import Engine
        if self.reader is None:
            self.reader = Engine(self.header, self._device_ordinal)
            if self.header.rg_ids:
                self.reader.set_read_group_ids(self.header.rg_ids)
            if self.comm.world > 1:
                if self.collective == "cabi":
                    self.reader.group_share(self.engines[0])  # RCCL send / receive on the communicator the tables are reduced on
                else:
                    self.reader.group_init_transport(self.comm.rank, self.comm.world, lambda v: None)
                    self.reader.group_set_p2p(self.comm.sendrecv)
        nl, ns = route_device(self.reader, b, group_of_ref, n_groups, owner, self.comm.rank, self.comm.world, self.engines[0], self.engines[1], stage)
        self.n[0] += nl
        self.n[1] += ns

    @property
    def n_reads(self) -> int:
        """reads of this rank's splits (the sr-tagged copies in the group splits are not reads of their own)"""
        return self.engines[0].n_sorted + self.engines[1].n

    def set_reference(self, refid: int, bases: np.ndarray):
        for e in self.engines:
            e.set_reference(refid, bases)

    def set_known_sites(self, refid: int, iv: np.ndarray):
        for e in self.engines:
            e.set_known_sites(refid, iv)

    def snapshot(self):
        for e in self.engines:
            e.snapshot()

    def rollback(self):
        for e in self.engines:
            e.rollback()

    def sync(self):
        for e in self.engines:
            e.sync()

    def gather(self, max_cycle: int, pixel_dist: int = 100):
        """mark duplicates + sort + duplication metrics + BQSR tables of every split of this rank, then THE all-reduce."""
        if self.collective == "torch":
            tot = None
            for e in self.engines:
                e.mark_duplicates(True, fetch=False)
                e.sort_coordinate(fetch=False)  # the sort is the Finalize step behind the filters (sam/filter-pipeline.go:116)
                ctr = e.dup_metrics(pixel_dist)
                qt, ct, xt = e.recalibrate(max_cycle)
                flat = np.concatenate([qt.ravel(), ct.ravel(), xt.ravel(), ctr.ravel()])
                tot = flat if tot is None else tot + flat
                shapes = (qt.shape, ct.shape, xt.shape, ctr.shape)
            t0 = time.perf_counter()
            tot = self.comm.allreduce_i64(tot)
            self.allreduce_s.append(time.perf_counter() - t0)
            out, at = [], 0
            for shp in shapes:
                n = int(np.prod(shp))
                out.append(tot[at:at + n].reshape(shp))
                at += n
            return tuple(out)  # (qual table, cycle table, context table, duplication counters): identical on every rank
        ctr = None
        for e in self.engines:
            e.mark_duplicates(True, fetch=False)
            e.sort_coordinate(fetch=False)
            c7 = e.dup_metrics(pixel_dist)
            ctr = c7 if ctr is None else ctr + c7
            e.recalibrate_device(max_cycle)  # tables stay in HBM
        e0 = self.engines[0]
        e0.tables_add(self.engines[1])       # this rank's splits, summed on the device
        e0.sync()                            # (so that the time below is the collective - and the wait for the slowest rank - alone)
        t0 = time.perf_counter()
        ctr = e0.tables_allreduce(ctr)       # RCCL, in place on the tables in HBM; the counters ride along
        self.allreduce_s.append(time.perf_counter() - t0)
        qt, ct, xt = e0.tables_fetch(reuse=True)
        return qt, ct, xt, ctr

    def step(self, max_cycle: int, pixel_dist: int, host_pool, finalize, finalize_rows=None):
        """One pass of the path over this rank's splits with the host's float64 finalisation hidden behind the sorts, as the one-context
        filter step has it: mark duplicates, duplication metrics (order-independent sums: they do not need the sort) and the BQSR count of
        every split, THE all-reduce (tables + counters), then the tables' way to the host, FinalizeBQSRTables and the LUT's upload to both
        contexts on a host thread while the GPU sorts the splits, then ApplyBQSR.  `finalize(qt, ct, xt) -> (lut, present)`;
        `finalize_rows(quals, q_rows, c_rows, x_rows) -> (rows, defaults, present)` (optional): the tables and the LUT in ROWS form - only the
        rows of the qualities this rank's contexts counted cross PCIe (Engine.tables_fetch_rows / lut_upload_rows); if another rank
        counted a quality this one did not, the all-reduced tables hold a row outside that set and the dense forms are taken.
        Returns the all-reduced duplication counters."""
        if self.collective == "torch":
            qt, ct, xt, ctr = self.gather(max_cycle, pixel_dist)
            lut, present = finalize(qt, ct, xt)
            self.apply(lut, present, max_cycle)
            return ctr
        # the spread split's context is small (a few per cent of the reads): its chain of launches and read-backs runs from a second host
        # thread on its own stream, under the kernels of the group splits' context, instead of in front of them
        e0, e1 = self.engines
        if self._side is None:
            from concurrent.futures import ThreadPoolExecutor
            self._side = ThreadPoolExecutor(1)

        def count(e, pool=None, sort_pool=None):
            e.mark_duplicates(True, fetch=False)
            st = None
            if pool is not None:
                # round 6: behind mark duplicates the coordinate sort, the metrics pass and the BQSR count need nothing of each other; the
                # library runs the first two on side lanes of the context (streams and scratch of their own), a host thread each drives them
                # under the gather's kernels
                st = sort_pool.submit(e.sort_coordinate, False)
                mx = pool.submit(e.dup_metrics, pixel_dist)
                e.recalibrate_device(max_cycle)  # tables stay in HBM
                c7 = mx.result()
            else:
                c7 = e.dup_metrics(pixel_dist)
                e.recalibrate_device(max_cycle)
            e.sync()
            return c7, st
        if getattr(self, "_mx_pool", None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._mx_pool, self._sort_pool = ThreadPoolExecutor(1), ThreadPoolExecutor(1)
        side = self._side.submit(count, e1) if self.n[1] else None
        ctr, sort0 = count(e0, self._mx_pool, self._sort_pool)
        if side is not None:
            ctr = ctr + side.result()[0]
            e0.tables_add(e1)
            e0.sync()                        # (so that the time below is the collective - and the wait for the slowest rank - alone)
        t0 = time.perf_counter()
        ctr = e0.tables_allreduce(ctr)
        self.allreduce_s.append(time.perf_counter() - t0)

        def host_side():
            if finalize_rows is not None:
                quals = sorted(set(e0.quals_counted()) | (set(e1.quals_counted()) if self.n[1] else set()))
                got = e0.tables_fetch_rows(quals, reuse=True)
                if got is not None:
                    rows, defaults, present = finalize_rows(quals, *got)
                    e0.lut_upload_rows(quals, rows, defaults, present, max_cycle)
                    if self.n[1]:
                        e1.lut_upload_rows(quals, rows, defaults, present, max_cycle)
                    return rows, present
            lut, present = finalize(*e0.tables_fetch(reuse=True))
            e0.lut_upload(lut, present, max_cycle)
            if self.n[1]:
                e1.lut_upload(lut, present, max_cycle)
            return lut, present
        fin = host_pool.submit(host_side)
        side = self._side.submit(lambda: e1.sort_coordinate(fetch=False)) if self.n[1] else None
        if side is not None:
            side.result()
        fin.result()
        side = self._side.submit(lambda: e1.apply_bqsr(None, None, max_cycle, fetch=False)) if self.n[1] else None
        e0.apply_bqsr(None, None, max_cycle, fetch=False)
        if side is not None:
            side.result()
        sort0.result()  # (the group splits' sort, running on its side lane since mark duplicates)
        return ctr

    def apply(self, lut: np.ndarray, present: np.ndarray, max_cycle: int):
        for e in self.engines:
            e.apply_bqsr(lut, present, max_cycle, fetch=False)

    def emit_merged(self, group_of_ref: np.ndarray, n_groups: int, owner: np.ndarray) -> np.ndarray:
        """the merge phase of this rank (emit_merged_device): the BAM records of its contig groups' output with the spread reads of those
        groups inserted - every context of the rank that sends or receives joins the device group first (the communicator the tables were
        reduced on, or the send-receive callback).  The records must have been staged from BAM bytes (route(..., stage=...))."""
        from .engine import Engine
        if getattr(self, "_part", None) is None:
            self._part = Engine(self.header, self._device_ordinal)
            if self.header.rg_ids:
                self._part.set_read_group_ids(self.header.rg_ids)
            if self.comm.world > 1:
                for e in (self._part, self.engines[1]):
                    if self.collective == "cabi":
                        e.group_share(self.engines[0])
                    else:
                        e.group_init_transport(self.comm.rank, self.comm.world, lambda v: None)
                        e.group_set_p2p(self.comm.sendrecv)
        return emit_merged_device(self.engines[0], self.engines[1], self._part, group_of_ref, n_groups, owner, self.comm.rank, self.comm.world)

    def close(self):
        if self._side is not None:
            self._side.shutdown()
            self._side = None
        if getattr(self, "_mx_pool", None) is not None:
            self._mx_pool.shutdown()
            self._sort_pool.shutdown()
            self._mx_pool = self._sort_pool = None
        if self.reader is not None:
            self.reader.close()
            self.reader = None
        if getattr(self, "_part", None) is not None:
            self._part.close()
            self._part = None
        for e in self.engines:
            e.close()
