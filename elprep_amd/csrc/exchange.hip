// exchange.hip — records from one context's column store to another's without a trip through the host.
//
// Reference: `elprep split` / `sfm` phase 1 (sam/split-merge.go:280-293: SplitFilePerChromosome writes a record to the file of its
// contig group - and, if its mate lies in another group, to the spread file as well, leaving an sr:i:1 tagged copy in the group file).
// A Go host is ONE process that drives all GPUs of a node through cgo, one context per GPU (or several per GPU): the split phase is then
// elp_split_classify on the context that decoded the records and elp_copy_records into the contexts that own the splits - column slices
// gathered on the source GPU and copied device to device (hipMemcpyPeerAsync over xGMI between GPUs).  One process per GPU (bench.py
// --gpus N, torch.distributed) exchanges through the harness instead (sfm.route).
//
// A record = its eleven fixed columns + its slices of QNAME, CIGAR, SEQ (already code nibbles), QUAL and - if both contexts hold the
// inflated BAM records (elp_stage_bam) - of those.  The mutable columns (FLAG, QUAL) travel as they are now.
#include <algorithm>

#include "common.hpp"

namespace elp {

constexpr int XV = 5;  // variable-length columns: QNAME bytes, CIGAR operations, SEQ bytes, QUAL bytes, raw BAM bytes
struct XSrc {
  uint64_t n_src;
  const int32_t *refid, *pos, *next_refid, *pnext, *tlen;
  const uint16_t *flag, *rgid, *split;
  const uint8_t *mapq, *has_sr;
  const uint32_t *l_seq;
  const uint64_t *off[XV];  // offset columns (n_src + 1); off[4] may be null
};
struct XFixed {
  int32_t *refid, *pos, *next_refid, *pnext, *tlen;
  uint16_t *flag, *rgid, *split;
  uint8_t *mapq, *has_sr;
  uint32_t *l_seq;
};
// counters: 0 max QNAME length, 1 max L_SEQ, 2 max POS (as uint32), 3 max split id, 4 records with state != 0, 5 with state 2, 6 bad index
enum { XC_QNAME, XC_LSEQ, XC_POS, XC_SPLIT, XC_NSR, XC_NFILT, XC_BAD, XC_N };

// lengths of the selected records' slices, the fixed columns, the limits the staging bookkeeping needs
__global__ __launch_bounds__(256) void k_x_fixed(uint64_t n, const uint32_t *__restrict__ idx, XSrc s, XFixed d, uint32_t *__restrict__ len /* [XV][n] */,
                                                 int new_split, int tag_sr, uint32_t *ctr) {
  __shared__ uint32_t acc[XC_N];
  if (threadIdx.x < XC_N) acc[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) {
    const uint32_t iw = idx[k];
    const bool tag_me = tag_sr == 1 || (tag_sr == 2 && (iw >> 31));  // tag_sr 2: bit 31 of an index = this copy is tagged
    const uint64_t i = tag_sr == 2 ? (iw & 0x7FFFFFFFu) : iw;
    if (i >= s.n_src) {
      atomicAdd(&acc[XC_BAD], 1u);
      for (int v = 0; v < XV; v++) len[(size_t)v * n + k] = 0;
    } else {
      uint8_t state = s.has_sr[i];
      if (tag_me && state == 0) state = 1;
      const uint16_t sp = new_split >= 0 ? (uint16_t)new_split : s.split[i];
      const int32_t p = s.pos[i];
      const uint32_t ls = s.l_seq[i];
      d.refid[k] = s.refid[i]; d.pos[k] = p; d.next_refid[k] = s.next_refid[i]; d.pnext[k] = s.pnext[i]; d.tlen[k] = s.tlen[i];
      d.flag[k] = s.flag[i]; d.rgid[k] = s.rgid[i]; d.split[k] = sp; d.mapq[k] = s.mapq[i]; d.has_sr[k] = state; d.l_seq[k] = ls;
      uint32_t ql = 0;
      for (int v = 0; v < XV; v++) {
        const uint32_t l = s.off[v] ? (uint32_t)(s.off[v][i + 1] - s.off[v][i]) : 0u;
        len[(size_t)v * n + k] = l;
        if (v == 0) ql = l;
      }
      atomicMax(&acc[XC_QNAME], ql);
      atomicMax(&acc[XC_LSEQ], ls);
      atomicMax(&acc[XC_POS], (uint32_t)p);
      atomicMax(&acc[XC_SPLIT], (uint32_t)sp);
      if (state) atomicAdd(&acc[XC_NSR], 1u);
      if (state == 2) atomicAdd(&acc[XC_NFILT], 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x < XC_N && acc[threadIdx.x]) {
    if (threadIdx.x <= XC_SPLIT) atomicMax(&ctr[threadIdx.x], acc[threadIdx.x]);
    else atomicAdd(&ctr[threadIdx.x], acc[threadIdx.x]);
  }
}

// one wave per record and column: the record's slice, 64 elements per step
template <class T>
__global__ __launch_bounds__(256) void k_x_var(uint64_t n, const uint32_t *__restrict__ idx, const uint64_t *__restrict__ src_off, const T *__restrict__ src,
                                               const uint32_t *__restrict__ dst_off, T *__restrict__ dst, uint64_t n_src, uint32_t idx_mask) {
  const uint64_t k = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (k >= n) return;
  const uint64_t i = idx[k] & idx_mask;
  if (i >= n_src) return;
  const uint64_t a = src_off[i], e = src_off[i + 1];
  const uint32_t o = dst_off[k];
  for (uint64_t j = a + (threadIdx.x & 63u); j < e; j += 64) dst[o + (j - a)] = src[j];
}

__global__ __launch_bounds__(256) void k_x_offsets(uint64_t n_plus_1, const uint32_t *__restrict__ excl, uint32_t total, uint64_t base, uint64_t *__restrict__ out) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_plus_1) out[k] = base + (k + 1 == n_plus_1 ? (uint64_t)total : (uint64_t)excl[k]);
}

static int peer_copy(elp_ctx *dst, void *to, elp_ctx *src, const void *from, size_t bytes) {
  if (!bytes) return 0;
  if (dst->device == src->device) ELP_HIP(src, hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, src->stream));
  else ELP_HIP(src, hipMemcpyPeerAsync(to, dst->device, from, src->device, bytes, src->stream));
  return 0;
}

}  // namespace elp

using namespace elp;

// ---- a piece of records, gathered on the source GPU: the fixed columns (eleven arrays, each n16 elements apart), the five
// variable-length pools (64-byte aligned sections) and the exclusive scans of the slice lengths (XV x (n + 1)); all in src's scratch
struct Gathered {
  uint64_t n = 0;
  uint32_t total[XV] = {0, 0, 0, 0, 0};
  uint32_t hc[XC_N] = {0};
  bool raw = false;
  uint8_t *fx = nullptr;   // fixed block, gathered_fixed_bytes(n)
  uint8_t *pool = nullptr; // gathered_pool_bytes(total)
  uint32_t *excl = nullptr;
};
static size_t n16_of(uint64_t n) { return (size_t)((n + 15) & ~(uint64_t)15); }
static size_t gathered_fixed_bytes(uint64_t n) { return n16_of(n) * 32; }  // 5 x 4 + 4 + 3 x 2 + 2 x 1 bytes per record
static size_t pool_section(size_t bytes) { return (bytes + 63) & ~(size_t)63; }
static size_t gathered_pool_bytes(const uint32_t *total) {
  return pool_section(total[0]) + pool_section((size_t)total[1] * 4) + pool_section(total[2]) + pool_section(total[3]) + pool_section(total[4]);
}
static XFixed fixed_view(uint8_t *fx, uint64_t n) {
  const size_t n16 = n16_of(n);
  XFixed F;
  uint8_t *p = fx;
  F.refid = (int32_t *)p; p += n16 * 4; F.pos = (int32_t *)p; p += n16 * 4; F.next_refid = (int32_t *)p; p += n16 * 4; F.pnext = (int32_t *)p; p += n16 * 4;
  F.tlen = (int32_t *)p; p += n16 * 4; F.l_seq = (uint32_t *)p; p += n16 * 4; F.flag = (uint16_t *)p; p += n16 * 2; F.rgid = (uint16_t *)p; p += n16 * 2;
  F.split = (uint16_t *)p; p += n16 * 2; F.mapq = p; p += n16; F.has_sr = p;
  return F;
}
static void pool_view(uint8_t *pool, const uint32_t *total, uint8_t **pv) {
  const size_t sz[XV] = {total[0], (size_t)total[1] * 4, total[2], total[3], total[4]};
  uint8_t *p = pool;
  for (int v = 0; v < XV; v++) { pv[v] = p; p += pool_section(sz[v]); }
}

// source GPU: lengths + fixed columns, scans, slices.  err_ctx = the context an error is reported on
static int gather_piece(elp_ctx *err_ctx, elp_ctx *src, const uint32_t *idx, uint64_t n, int new_split, int tag_sr, bool raw, Gathered *G) {
  if (tag_sr == 2 && src->n > 0x7FFFFFFFull) return set_error(err_ctx, ELP_ERR_UNSUPPORTED, "per-record sr tags (tag_sr = 2) need a source context of at most 2^31 records");
  ELP_HIP(src, hipSetDevice(src->device));
  hipStream_t ss = src->stream;
  uint32_t *w;  // idx | len[XV][n] | excl[XV][n + 1] | counters
  ELP_TRY(scratch(src, 6, n + (size_t)XV * n + (size_t)XV * (n + 1) + 64, &w));
  uint32_t *d_idx = w, *len = w + n, *excl = len + (size_t)XV * n, *ctr = excl + (size_t)XV * (n + 1);
  ELP_HIP(src, hipMemcpyAsync(d_idx, idx, n * 4, hipMemcpyHostToDevice, ss));
  ELP_HIP(src, hipMemsetAsync(ctr, 0, XC_N * 4, ss));
  uint8_t *fx;
  ELP_TRY(scratch(src, 5, gathered_fixed_bytes(n) + 256, &fx));
  const XFixed F = fixed_view(fx, n);
  XSrc S{src->n, src->refid.p, src->pos.p, src->next_refid.p, src->pnext.p, src->tlen.p, src->flag.p, src->rgid.p, src->split.p, src->mapq.p, src->has_sr.p,
         src->l_seq.p, {src->qname_off.p, src->cigar_off.p, src->seq_off.p, src->qual_off.p, raw ? src->raw_off.p : nullptr}};
  hipLaunchKernelGGL(k_x_fixed, dim3(blocks_for(n, 256)), dim3(256), 0, ss, n, (const uint32_t *)d_idx, S, F, len, new_split, tag_sr, ctr);
  ELP_HIP(src, hipGetLastError());
  for (int v = 0; v < XV; v++) {
    G->total[v] = 0;
    if (v == 4 && !raw) continue;
    ELP_TRY(exclusive_scan_u32(src, len + (size_t)v * n, excl + (size_t)v * (n + 1), n, &G->total[v]));  // (a slice total beyond 2^32: see the check below)
  }
  ELP_HIP(src, hipMemcpyAsync(G->hc, ctr, sizeof G->hc, hipMemcpyDeviceToHost, ss));
  ELP_HIP(src, elp::stream_wait(ss));
  if (G->hc[XC_BAD]) return set_error(err_ctx, ELP_ERR_ARG, "elp_copy_records: %u indices are not records of the source context", G->hc[XC_BAD]);
  if (G->hc[XC_QNAME] > elp_ctx::MAX_QNAME) return set_error(err_ctx, ELP_ERR_UNSUPPORTED, "QNAME of %u bytes (limit %u)", G->hc[XC_QNAME], elp_ctx::MAX_QNAME);
  // (the scans are 32-bit: a call moves at most 4 GiB of any one column - ~25 M reads of 150 bases; callers move larger sets in pieces)
  if ((uint64_t)n * (uint64_t)std::max<uint32_t>(G->hc[XC_LSEQ], 1) >= 0xFFFFFFFFull || (raw && (uint64_t)n * src->max_raw_rec >= 0xFFFFFFFFull))
    return set_error(err_ctx, ELP_ERR_UNSUPPORTED, "elp_copy_records: more than 4 GiB of one column in one call: move the records in pieces");
  uint8_t *pool;
  ELP_TRY(scratch(src, 4, gathered_pool_bytes(G->total) + 64, &pool));
  uint8_t *pv[XV];
  pool_view(pool, G->total, pv);
  const unsigned vgrid = blocks_for(n * 64, 256);
  const uint32_t idx_mask = tag_sr == 2 ? 0x7FFFFFFFu : 0xFFFFFFFFu;
  hipLaunchKernelGGL(k_x_var<uint8_t>, dim3(vgrid), dim3(256), 0, ss, n, (const uint32_t *)d_idx, (const uint64_t *)src->qname_off.p, (const uint8_t *)src->qname.p,
                     (const uint32_t *)excl, pv[0], src->n, idx_mask);
  hipLaunchKernelGGL(k_x_var<uint32_t>, dim3(vgrid), dim3(256), 0, ss, n, (const uint32_t *)d_idx, (const uint64_t *)src->cigar_off.p, (const uint32_t *)src->cigar.p,
                     (const uint32_t *)(excl + (n + 1)), (uint32_t *)pv[1], src->n, idx_mask);
  hipLaunchKernelGGL(k_x_var<uint8_t>, dim3(vgrid), dim3(256), 0, ss, n, (const uint32_t *)d_idx, (const uint64_t *)src->seq_off.p, (const uint8_t *)src->seq4.p,
                     (const uint32_t *)(excl + 2 * (n + 1)), pv[2], src->n, idx_mask);
  hipLaunchKernelGGL(k_x_var<uint8_t>, dim3(vgrid), dim3(256), 0, ss, n, (const uint32_t *)d_idx, (const uint64_t *)src->qual_off.p, (const uint8_t *)src->qual.p,
                     (const uint32_t *)(excl + 3 * (n + 1)), pv[3], src->n, idx_mask);
  if (raw)
    hipLaunchKernelGGL(k_x_var<uint8_t>, dim3(vgrid), dim3(256), 0, ss, n, (const uint32_t *)d_idx, (const uint64_t *)src->raw_off.p, (const uint8_t *)src->raw.p,
                       (const uint32_t *)(excl + 4 * (n + 1)), pv[4], src->n, idx_mask);
  ELP_HIP(src, hipGetLastError());
  G->n = n; G->raw = raw; G->fx = fx; G->pool = pool; G->excl = excl;
  return 0;
}

// destination: room, then the copies - from `from`'s device and on its stream (the gathering context, or dst itself for a piece that
// arrived through the group) -, then the offset columns and the commit (as elp_stage does)
static int append_piece(elp_ctx *dst, elp_ctx *from, const Gathered &G, uint64_t src_max_raw_rec) {
  const uint64_t n = G.n;
  const bool raw = G.raw;
  const uint32_t *total = G.total, *hc = G.hc;
  ELP_HIP(dst, hipSetDevice(dst->device));
  ELP_TRY(stage_reserve(dst, dst->n + n, dst->qname_bytes + total[0], dst->cigar_ops + total[1], dst->seq_bytes + total[2], dst->qual_bytes + total[3]));
  if (raw) {
    const bool keep = dst->raw_n > 0;
    ELP_TRY(ensure(dst, dst->raw, dst->raw_bytes + total[4] + 64, keep, dst->raw_bytes));
    ELP_TRY(ensure(dst, dst->raw_off, dst->n + n + 1, keep, dst->raw_n + 1));
  }
  uint32_t *dx;  // the exclusive scans, on the destination GPU
  ELP_TRY(ensure(dst, dst->stage_tmp, ((size_t)XV * (n + 1) + 1) / 2 + 8));
  dx = reinterpret_cast<uint32_t *>(dst->stage_tmp.p);
  ELP_HIP(dst, elp::stream_wait(dst->stream));  // (the destination's columns may just have moved to larger allocations on its stream)
  ELP_HIP(from, hipSetDevice(from->device));
  const uint64_t at = dst->n;
  const XFixed F = fixed_view(G.fx, n);
  uint8_t *pv[XV];
  pool_view(G.pool, total, pv);
#define XCOPY(field, T) ELP_TRY(peer_copy(dst, dst->field.p + at, from, F.field, n * sizeof(T)))
  XCOPY(refid, int32_t); XCOPY(pos, int32_t); XCOPY(next_refid, int32_t); XCOPY(pnext, int32_t); XCOPY(tlen, int32_t); XCOPY(l_seq, uint32_t);
  XCOPY(flag, uint16_t); XCOPY(rgid, uint16_t); XCOPY(split, uint16_t); XCOPY(mapq, uint8_t); XCOPY(has_sr, uint8_t);
#undef XCOPY
  ELP_TRY(peer_copy(dst, dst->qname.p + dst->qname_bytes, from, pv[0], total[0]));
  ELP_TRY(peer_copy(dst, dst->cigar.p + dst->cigar_ops, from, pv[1], (size_t)total[1] * 4));
  ELP_TRY(peer_copy(dst, dst->seq4.p + dst->seq_bytes, from, pv[2], total[2]));
  ELP_TRY(peer_copy(dst, dst->qual.p + dst->qual_bytes, from, pv[3], total[3]));
  if (raw) ELP_TRY(peer_copy(dst, dst->raw.p + dst->raw_bytes, from, pv[4], total[4]));
  ELP_TRY(peer_copy(dst, dx, from, G.excl, (size_t)XV * (n + 1) * 4));
  ELP_HIP(from, elp::stream_wait(from->stream));
  ELP_HIP(dst, hipSetDevice(dst->device));
  {
    struct { uint64_t *out; uint64_t base; int v; } oc[XV] = {{dst->qname_off.p + at, dst->qname_bytes, 0}, {dst->cigar_off.p + at, dst->cigar_ops, 1},
                                                              {dst->seq_off.p + at, dst->seq_bytes, 2}, {dst->qual_off.p + at, dst->qual_bytes, 3},
                                                              {raw ? dst->raw_off.p + at : nullptr, dst->raw_bytes, 4}};
    for (auto &o : oc) {
      if (!o.out) continue;
      hipLaunchKernelGGL(k_x_offsets, dim3(blocks_for(n + 1, 256)), dim3(256), 0, dst->stream, n + 1, (const uint32_t *)(dx + (size_t)o.v * (n + 1)), total[o.v], o.base, o.out);
    }
    ELP_HIP(dst, hipGetLastError());
  }
  ELP_HIP(dst, elp::stream_wait(dst->stream));
  dst->n += n; dst->qname_bytes += total[0]; dst->cigar_ops += total[1]; dst->seq_bytes += total[2]; dst->qual_bytes += total[3];
  if (raw) { dst->raw_n = dst->n; dst->raw_bytes += total[4]; dst->max_raw_rec = std::max(dst->max_raw_rec, src_max_raw_rec); }
  dst->n_sr += hc[XC_NSR];
  dst->n_filtered += hc[XC_NFILT];
  dst->max_split = std::max(dst->max_split, hc[XC_SPLIT]);
  dst->max_qname_len = std::max(dst->max_qname_len, hc[XC_QNAME]);
  dst->max_l_seq = std::max(dst->max_l_seq, hc[XC_LSEQ]);
  dst->max_pos = std::max(dst->max_pos, hc[XC_POS]);
  dst->adapted = dst->sorted = dst->marked = false;
  dst->have_qual_present = false;
  dst->have_snapshot = false;
  dst->flat_index_n = 0;
  dst->uniform_n = ~0ull;
  return 0;
}

static int copy_piece(elp_ctx *dst, elp_ctx *src, const uint32_t *idx, uint64_t n, int new_split, int tag_sr);

// the call moves its records in pieces whose columns stay below 4 GiB each (the gather's offsets are scanned in 32 bits)
extern "C" int elp_copy_records(elp_ctx *dst, elp_ctx *src, const uint32_t *idx, uint64_t n, int new_split, int tag_sr) {
  if (!dst || !src || dst == src || (!idx && n)) return set_error(dst, ELP_ERR_ARG, "elp_copy_records: bad arguments");
  const uint64_t widest = std::max<uint64_t>({(uint64_t)src->max_l_seq, (uint64_t)elp_ctx::MAX_QNAME, src->max_raw_rec, 1024});
  const uint64_t piece = std::max<uint64_t>(1, std::min<uint64_t>(1u << 22, 0xF0000000ull / widest));
  for (uint64_t at = 0; at < n; at += piece) {
    const int rc = copy_piece(dst, src, idx + at, std::min<uint64_t>(piece, n - at), new_split, tag_sr);
    if (rc) return rc;  // (pieces already appended stay: the call reports how far it got through dst's record count)
  }
  return 0;
}

static int copy_piece(elp_ctx *dst, elp_ctx *src, const uint32_t *idx, uint64_t n, int new_split, int tag_sr) {
  if (!dst || !src || dst == src || (!idx && n)) return set_error(dst, ELP_ERR_ARG, "elp_copy_records: bad arguments");
  if (!dst->have_header || !src->have_header || dst->n_ref != src->n_ref || dst->n_rg != src->n_rg || dst->h_ref_len != src->h_ref_len)
    return set_error(dst, ELP_ERR_ARG, "elp_copy_records: the two contexts need the same header");
  if (new_split > 0xFFFF) return set_error(dst, ELP_ERR_ARG, "elp_copy_records: split id %d", new_split);
  if (n == 0) return 0;
  std::lock(dst->stage_mu, src->stage_mu);
  std::lock_guard<std::mutex> g1(dst->stage_mu, std::adopt_lock), g2(src->stage_mu, std::adopt_lock);
  if (dst->n + n > 0xFFFFFFF0ull) return set_error(dst, ELP_ERR_UNSUPPORTED, "more than 2^32-16 records per context");
  // the inflated BAM records travel if both sides hold them for every record they have (else the destination could not emit)
  const bool src_raw = src->raw_n == src->n && src->n > 0, dst_raw = dst->raw_n == dst->n && (dst->n > 0 || src_raw);
  if ((src->raw_n && !src_raw) || (dst->raw_n && !dst_raw) || (dst->n > 0 && (dst->raw_n > 0) != src_raw))
    return set_error(dst, ELP_ERR_UNSUPPORTED, "elp_copy_records: either both contexts hold the inflated BAM records of all their reads (elp_stage_bam) or neither does");
  Gathered G;
  ELP_TRY(gather_piece(dst, src, idx, n, new_split, tag_sr, src_raw, &G));
  return append_piece(dst, src, G, src->max_raw_rec);
}

// ------------------------------------------------------------------ between PROCESSES: one step of the split phase's all-to-all
// elp_exchange_records: the records `idx` of `src` go to rank send_peer of the device group, the records rank recv_peer selected for this
// rank in ITS matching call are appended to `dst` (either side may be absent: peer -1).  The records move in pieces whose columns stay
// below 4 GiB each (as elp_copy_records: the gather's offsets are scanned in 32 bits); a piece is gathered on the source GPU as for
// elp_copy_records.  Per piece and direction: a 160-byte header (counts, totals, limits, the sender's status, the number of pieces), an
// 8-byte verdict back from the receiver (its destination-side checks), then - only if both are clean - the three payload blocks: fixed
// columns, pools, scans, device to device: ncclSend / ncclRecv inside one ncclGroupStart / ncclGroupEnd per message (RCCL over xGMI), or
// the group's own send-receive callback through page-locked memory (elp_group_set_p2p: a host with its own communicator, and the tests
// on one GPU).  No rank returns while its peer still waits in a matching message: a failure on either side of a direction
// travels in the header or the verdict, both sides then skip that direction's payload and its remaining pieces, and report the error.
// That covers failures of the DATA (a bad index, limits, records the destination cannot take).  A failure of the machinery in mid-protocol -
// a HIP call, the transport itself - returns at once: the message the peer waits for would have to travel through what just failed, so
// such an error is FATAL FOR THE GROUP: the caller tears the group down (elp_group_init again, or the process), as a host
// would after a failed ncclSend; RCCL's own watchdog / the transport's time-out is what releases the peer.
extern "C" int elp_exchange_records(elp_ctx *src, int send_peer, const uint32_t *idx, uint64_t n, int new_split, int tag_sr, elp_ctx *dst, int recv_peer) {
  // the context that belongs to the device group: the one of the two that has a communicator / a transport
  elp_ctx *g = (src && (src->comm || src->p2p)) ? src : ((dst && (dst->comm || dst->p2p)) ? dst : (src ? src : dst));
  if (!g || (send_peer >= 0 && (!src || (!idx && n))) || (recv_peer >= 0 && !dst)) return set_error(g, ELP_ERR_ARG, "elp_exchange_records: bad arguments");
  if (send_peer < 0) n = 0;
  if (new_split > 0xFFFF) return set_error(g, ELP_ERR_ARG, "elp_exchange_records: split id %d", new_split);
  constexpr int HDR = 20;  // 0 records, 1-5 slice totals, 6-12 limits / counts (XC_*), 13 raw kind, 14 largest raw record, 15 magic, 16 status, 17 pieces, 18 piece
  constexpr uint64_t MAGIC = 0x454c505845434847ull;
  uint64_t piece = 1;
  uint64_t out_pieces = 0;
  if (send_peer >= 0) {
    const uint64_t widest = std::max<uint64_t>({(uint64_t)src->max_l_seq, (uint64_t)elp_ctx::MAX_QNAME, src->max_raw_rec, 1024});
    piece = std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)(src->tune.exchange_piece > 0 ? src->tune.exchange_piece : (1 << 22)), 0xF0000000ull / widest));
    out_pieces = std::max<uint64_t>(1, (n + piece - 1) / piece);  // (a rank without records still sends one header: the receiver learns there is nothing)
  }
  uint64_t *d_hdr;  // out | in | verdict out | verdict in, on the group context's device
  ELP_HIP(g, hipSetDevice(g->device));
  ELP_TRY(ensure(g, g->xchg_hdr, 2 * HDR + 16));
  d_hdr = reinterpret_cast<uint64_t *>(g->xchg_hdr.p);
  uint64_t *d_ack = d_hdr + 2 * HDR;
  int first_err = 0;         // the call's result: the first failure of either direction (the error text is that one's)
  std::string first_text;
  auto note = [&](int st) { if (st && !first_err) { first_err = st; first_text = g->err; } };
  bool out_on = send_peer >= 0, in_on = recv_peer >= 0;
  uint64_t in_pieces = 1;    // known from the first header
  int in_refuse = 0;         // a piece arrived but could not be appended: the next header is answered with this verdict
  for (uint64_t j = 0; (out_on && j < out_pieces) || (in_on && j < in_pieces); j++) {
    const bool out_now = out_on && j < out_pieces, in_now = in_on && j < in_pieces;
    // ---- what goes out: piece j, gathered on the source GPU
    Gathered G;
    uint64_t h_out[HDR] = {0}, h_in[HDR] = {0};
    uint64_t n_j = 0;
    if (out_now) {
      std::lock_guard<std::mutex> lk(src->stage_mu);
      const bool raw = src->raw_n == src->n && src->n > 0;
      int st = 0;
      n_j = n > j * piece ? std::min<uint64_t>(piece, n - j * piece) : 0;
      if (src->raw_n && !raw) st = set_error(g, ELP_ERR_UNSUPPORTED, "elp_exchange_records: some but not all records of the source hold their inflated BAM bytes");
      else if (n_j) st = gather_piece(g, src, idx + j * piece, n_j, new_split, tag_sr, raw, &G);
      if (st) { n_j = 0; note(st); }
      h_out[0] = n_j;
      for (int v = 0; v < XV; v++) h_out[1 + v] = G.total[v];
      for (int k = 0; k < XC_N; k++) h_out[6 + k] = G.hc[k];
      h_out[13] = n_j ? (G.raw ? 1 : 0) : 2;  // 2: no records, either kind of destination is fine
      h_out[14] = src->max_raw_rec;
      h_out[15] = MAGIC;
      h_out[16] = (uint64_t)(uint32_t)(-st);
      h_out[17] = out_pieces;
      h_out[18] = j;
    }
    // ---- headers
    elp_ctx *sc = send_peer >= 0 ? src : dst;
    ELP_HIP(g, hipSetDevice(g->device));
    ELP_HIP(g, hipMemcpyAsync(d_hdr, h_out, sizeof h_out, hipMemcpyHostToDevice, g->stream));
    ELP_HIP(g, elp::stream_wait(sc->stream));  // (the gather ran on the source's stream)
    ELP_HIP(g, elp::stream_wait(g->stream));
    ELP_TRY(group_sendrecv(g, out_now ? send_peer : -1, d_hdr, sizeof h_out, in_now ? recv_peer : -1, d_hdr + HDR, sizeof h_in));
    ELP_HIP(g, hipMemcpyAsync(h_in, d_hdr + HDR, sizeof h_in, hipMemcpyDeviceToHost, g->stream));
    ELP_HIP(g, elp::stream_wait(g->stream));
    // ---- the receiver's checks and buffers, then its verdict back to the sender
    Gathered R;
    uint64_t verdict_out = 0, verdict_in = 0;
    if (in_now) {
      int st = 0;
      if (in_refuse) st = in_refuse;
      else if (h_in[15] != MAGIC || h_in[18] != j) st = set_error(g, ELP_ERR_DATA, "elp_exchange_records: rank %d did not send the header of piece %llu (calls out of step?)", recv_peer, (unsigned long long)j);
      else if (h_in[16]) st = set_error(g, -(int)(uint32_t)h_in[16], "elp_exchange_records: rank %d failed on its side of the exchange (status %d)", recv_peer, -(int)(uint32_t)h_in[16]);
      else {
        if (j == 0) in_pieces = std::max<uint64_t>(1, h_in[17]);
        R.n = h_in[0];
        for (int v = 0; v < XV; v++) R.total[v] = (uint32_t)h_in[1 + v];
        for (int k = 0; k < XC_N; k++) R.hc[k] = (uint32_t)h_in[6 + k];
        R.raw = h_in[13] == 1;
        if (R.n) {
          std::lock_guard<std::mutex> lk(dst->stage_mu);
          const bool dst_raw = dst->raw_n == dst->n && dst->n > 0;
          if (dst->n + R.n > 0xFFFFFFF0ull) st = set_error(g, ELP_ERR_UNSUPPORTED, "more than 2^32-16 records per context");
          else if ((dst->raw_n && !dst_raw) || (dst->n > 0 && dst_raw != R.raw))
            st = set_error(g, ELP_ERR_UNSUPPORTED, "elp_exchange_records: the records that arrive and the destination's differ in whether they hold inflated BAM bytes");
          else if (hipSetDevice(dst->device) != hipSuccess) st = set_error(g, ELP_ERR_HIP, "hipSetDevice failed");
          else if ((st = scratch(dst, 1, gathered_fixed_bytes(R.n) + 256, &R.fx)) == 0 && (st = scratch(dst, 2, gathered_pool_bytes(R.total) + 64, &R.pool)) == 0 &&
                   (st = scratch(dst, 3, (size_t)XV * (R.n + 1) + 16, &R.excl)) == 0)
            (void)elp::stream_wait(dst->stream);
        }
      }
      if (st) { R.n = 0; in_on = false; note(st); verdict_out = (uint64_t)(uint32_t)(-st); }
    }
    // the verdict travels against the records: to the rank I receive from, from the rank I send to
    ELP_HIP(g, hipSetDevice(g->device));
    ELP_HIP(g, hipMemcpyAsync(d_ack, &verdict_out, 8, hipMemcpyHostToDevice, g->stream));
    ELP_HIP(g, elp::stream_wait(g->stream));
    ELP_TRY(group_sendrecv(g, in_now ? recv_peer : -1, d_ack, 8, out_now ? send_peer : -1, d_ack + 1, 8));
    if (out_now) {
      ELP_HIP(g, hipMemcpyAsync(&verdict_in, d_ack + 1, 8, hipMemcpyDeviceToHost, g->stream));
      ELP_HIP(g, elp::stream_wait(g->stream));
      if (verdict_in) {
        note(set_error(g, -(int)(uint32_t)verdict_in, "elp_exchange_records: rank %d refused the records (status %d)", send_peer, -(int)(uint32_t)verdict_in));
      }
      if (verdict_in || h_out[16]) { out_on = false; n_j = 0; }
    }
    // ---- payload: three blocks each way (a direction without records, or one that failed, sends / receives none: both ends know)
    const int sp = (out_now && n_j) ? send_peer : -1, rp = (in_now && R.n) ? recv_peer : -1;
    if (sp >= 0 || rp >= 0) {
      ELP_TRY(group_sendrecv(g, sp, G.fx, n_j ? gathered_fixed_bytes(n_j) : 0, rp, R.fx, R.n ? gathered_fixed_bytes(R.n) : 0));
      ELP_TRY(group_sendrecv(g, sp, G.pool, n_j ? gathered_pool_bytes(G.total) : 0, rp, R.pool, R.n ? gathered_pool_bytes(R.total) : 0));
      ELP_TRY(group_sendrecv(g, sp, G.excl, n_j ? (size_t)XV * (n_j + 1) * 4 : 0, rp, R.excl, R.n ? (size_t)XV * (R.n + 1) * 4 : 0));
      ELP_HIP(g, elp::stream_wait(g->stream));
    }
    if (R.n) {
      std::lock_guard<std::mutex> lk(dst->stage_mu);
      const int st = append_piece(dst, dst, R, h_in[14]);
      // (the records of this piece have arrived; a failure to append them ends this direction with the next header's verdict)
      if (st) { note(st); in_refuse = st; }
    }
  }
  if (first_err) g->err = first_text;
  return first_err;
}
