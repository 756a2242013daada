// filter.hip — the per-record predicates next to the hot path, evaluated on the column store, and the `elprep split` / `merge`
// bookkeeping done in HBM.
//
// (1) Fused predicates.  Reference: filters/simple-filters.go — RemoveUnmappedReads :73-75, RemoveUnmappedReadsStrict :79-83,
//     RemoveNonExactMappingReads :90-99, RemoveDuplicateReads :136-138, RemoveNonOverlappingReads :310-328 (+ intervals.Overlap,
//     intervals/intervals.go:146-160), RemoveMappingQualityLessThan :332-347.  In `elprep filter` they stand in FRONT of AddREFID
//     and MarkDuplicates in filters1 (cmd/filter.go:696-803): a read they reject never reaches duplicate marking, the sort, the
//     metrics or BQSR.  Here one kernel evaluates the selected predicates for every staged record and marks the rejected ones in the
//     record-state column (2 = filtered; 1 = sr-tagged copy, which still takes part in duplicate marking): every later stage skips
//     them, the sort puts them behind the output (elp_num_sorted).
// (2) `elprep split`: SplitFilePerChromosome's routing rule (sam/split-merge.go:280-293) per record — split of RNAME and the
//     "spread" test — and the per-split record counts, so that a host can partition a staged file across GPUs without touching the
//     payload on the CPU.
// (3) `elprep merge`: MergeSortedFilesSplitPerChromosome (sam/split-merge.go:410-576) as a rank computation: where every record of
//     the coordinate-sorted spread split goes among the concatenated, coordinate-sorted group splits (behind all group reads of
//     its position).
#include "common.hpp"

namespace elp {

struct FilterCols {
  uint64_t n;
  const int32_t *refid, *pos;
  const uint16_t *flag;
  const uint8_t *mapq;
  const uint64_t *cigar_off;
  const uint32_t *cigar;
  uint8_t *state;  // has_sr column: 0 live, 1 sr-tagged, 2 filtered
  int32_t n_ref;
  int32_t *const *regions;   // per refid: [n][2] Start, End of intervals.FromBed, sorted by Start and flattened
  const int64_t *n_regions;
};

// intervals.Overlap, intervals/intervals.go:146-160 (binary search, same comparisons)
__device__ inline bool overlap(const int32_t *iv, int64_t n, int32_t start, int32_t end) {
  int64_t left = 0, right = n - 1;
  while (left <= right) {
    const int64_t mid = (left + right) / 2;
    const int32_t is = iv[2 * mid], ie = iv[2 * mid + 1];
    if (is > end - 1) right = mid - 1;
    else if (ie <= start - 1) left = mid + 1;
    else return true;
  }
  return false;
}

__global__ __launch_bounds__(256) void k_filter_records(FilterCols m, elp_predicates p, unsigned long long *n_dropped) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool drop = false, tagged = false;
  if (i < m.n && m.state[i] != 2) {
    const uint16_t f = m.flag[i];
    const int32_t r = m.refid[i], ps = m.pos[i];
    if (p.remove_unmapped && (f & F_UNMAPPED)) drop = true;
    if (p.remove_unmapped_strict && ((f & F_UNMAPPED) || ps == 0 || r < 0)) drop = true;
    if (p.min_mapq > 0 && (p.min_mapq > 255 || (int)m.mapq[i] < p.min_mapq)) drop = true;
    if (p.remove_duplicates && (f & F_DUPLICATE)) drop = true;
    if (!drop && (p.remove_non_exact || p.use_regions)) {
      int32_t ref_len = 0, read_len = 0;
      bool exact = true;
      for (uint64_t k = m.cigar_off[i]; k < m.cigar_off[i + 1]; k++) {
        const uint32_t c = m.cigar[k], op = c & 0xF;
        const int32_t ln = (int32_t)(c >> 4);
        if (op != OP_M && op != OP_S) exact = false;  // nonExactMappingOperator: I D N H P X =
        if (op_consumes_ref(op)) ref_len += ln;
        if (op_consumes_read(op)) read_len += ln;
      }
      if (p.remove_non_exact && !exact) drop = true;
      if (!drop && p.use_regions) {
        int32_t a_end = ps;  // :318-324
        if (!(f & F_UNMAPPED) && read_len > 0) a_end = ps + ref_len - 1;
        drop = !(r >= 0 && r < m.n_ref && overlap(m.regions[r], m.n_regions[r], ps, a_end));
      }
    }
    if (drop) {
      tagged = m.state[i] == 1;  // a tagged copy that is rejected was already counted among the records that leave the output
      drop = !tagged;
      m.state[i] = 2;
    }
  }
  const unsigned long long b = __ballot(drop), t = __ballot(tagged);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_dropped, (unsigned long long)__popcll(b));
  if ((threadIdx.x & 63) == 0 && t) atomicAdd(n_dropped + 1, (unsigned long long)__popcll(t));  // rejected tagged copies
}

// ---- split: routing rule of SplitFilePerChromosome
__global__ __launch_bounds__(256) void k_split_classify(uint64_t n, const int32_t *__restrict__ refid, const int32_t *__restrict__ next_refid,
                                                        const int32_t *__restrict__ group_of_ref, int32_t n_ref, int32_t n_groups,
                                                        uint16_t *__restrict__ split_out, uint8_t *__restrict__ spread_out,
                                                        unsigned long long *__restrict__ counts /* [n_groups + 2]: unmapped, groups, spread */,
                                                        uint16_t *__restrict__ split_col /* the context's split-id column */) {
  extern __shared__ unsigned int lds_cnt[];
  for (int k = threadIdx.x; k < n_groups + 2; k += blockDim.x) lds_cnt[k] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    const int32_t r = refid[i], nr = next_refid[i];
    const int32_t g = (r >= 0 && r < n_ref) ? group_of_ref[r] : 0;       // contigToGroup["*"] = "unmapped" (:254)
    const int32_t gn = (nr >= 0 && nr < n_ref) ? group_of_ref[nr] : 0;
    // :286: RNEXT != "=" (BAM: next_refid != refid, sam/bam-files.go:344-346), RNAME != "*", and the mate's group differs
    const bool spread = nr != r && r >= 0 && gn != g;
    split_out[i] = (uint16_t)g;
    split_col[i] = (uint16_t)g;  // the record's split file: travels with it (elp_copy_records / elp_exchange_records keep it)
    spread_out[i] = spread ? 1 : 0;
    atomicAdd(&lds_cnt[g], 1u);
    if (spread) atomicAdd(&lds_cnt[n_groups + 1], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n_groups + 2; k += blockDim.x)
    if (lds_cnt[k]) atomicAdd(&counts[k], (unsigned long long)lds_cnt[k]);
}

// ---- merge: spread read j (key ks[j]) goes behind every group read with key <= ks[j]
__global__ __launch_bounds__(256) void k_merge_rank(uint64_t ng, const uint64_t *__restrict__ kg, uint64_t ns, const uint64_t *__restrict__ ks,
                                                    uint64_t *__restrict__ slot_of_spread) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ns) return;
  const uint64_t key = ks[j];
  uint64_t lo = 0, hi = ng;  // first group read with key > ks[j]
  while (lo < hi) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (kg[mid] <= key) lo = mid + 1; else hi = mid;
  }
  slot_of_spread[j] = lo + j;  // spread reads keep their own order
}
__global__ __launch_bounds__(256) void k_perm_keys(uint64_t n_out, const uint32_t *__restrict__ perm, const int32_t *__restrict__ refid,
                                                   const int32_t *__restrict__ pos, uint64_t *__restrict__ keys) {
  const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_out) return;
  const uint32_t i = perm[k];
  keys[k] = ((uint64_t)(uint32_t)refid[i] << 32) | (uint64_t)(uint32_t)pos[i];  // (refid, POS) as the merge loop compares them (:417-434)
}


// MergeSortedFilesSplitPerChromosome's order as ranks, on the device: slots[j] = output slot of the j-th record of `spread`'s sorted output
// among `groups`' sorted output (scratch slot 5 of `groups`; valid until that slot is reused).  Queued on groups->stream.
int merge_spread_slots(elp_ctx *groups, elp_ctx *spread, uint64_t **slots_out) {
  if (!groups->sorted || !spread->sorted) return set_error(groups, ELP_ERR_ARG, "elp_merge_spread: both contexts must be coordinate-sorted");
  if (groups->device != spread->device) return set_error(groups, ELP_ERR_ARG, "elp_merge_spread: contexts on different devices");
  ELP_HIP(groups, hipSetDevice(groups->device));
  // (a sort defers the read of its radix passes' look-back timeout bit: nothing is merged from a permutation that was flagged wrong)
  ELP_TRY(radix_check(groups));
  if (radix_check(spread) != 0) return set_error(groups, ELP_ERR_HIP, "elp_merge_spread: the spread context's sort failed: %s", spread->err.c_str());
  // the mapped part of the groups' output (the unmapped split is appended behind the merge, :560-576)
  const uint64_t ns = spread->n - spread->n_sr;
  uint64_t *kg, *ks, *slots;
  ELP_TRY(scratch(groups, 5, (groups->n + 8) + 2 * (ns + 8), &kg));
  ks = kg + groups->n + 8;
  slots = ks + ns + 8;
  const uint64_t ng_all = groups->n - groups->n_sr;
  ELP_HIP(groups, elp::stream_wait(spread->stream));
  if (ng_all)
    ELP_LAUNCH(groups, "merge_keys", k_perm_keys, dim3(blocks_for(ng_all, 256)), dim3(256), 0, ng_all, (const uint32_t *)groups->perm.p,
               (const int32_t *)groups->refid.p, (const int32_t *)groups->pos.p, kg);
  if (ns) {
    ELP_LAUNCH(groups, "merge_keys", k_perm_keys, dim3(blocks_for(ns, 256)), dim3(256), 0, ns, (const uint32_t *)spread->perm.p,
               (const int32_t *)spread->refid.p, (const int32_t *)spread->pos.p, ks);
    // unmapped reads (refid -1 -> 0xFFFFFFFF........) sort behind every contig: the search covers them without a special case
    ELP_LAUNCH(groups, "merge_rank", k_merge_rank, dim3(blocks_for(ns, 256)), dim3(256), 0, ng_all, (const uint64_t *)kg, ns, (const uint64_t *)ks, slots);
  }
  *slots_out = slots;
  return 0;
}
// ---------------- CleanSam (filters/simple-filters.go:292-306): MAPQ 0 for unmapped reads; an alignment that ends behind its reference
// sequence is soft-clipped there by softClipEndOfRead (filters/utils.go:102-119) - restated with its arithmetic as it stands: the
// running position accumulates (`pos += endPos`, :116) and the clip's length is ReadLengthFromCigar + clipFrom (:112); a drop-in must
// write what the reference writes.  The rewritten CIGAR has at most one operation more than the old one, so the CIGAR column is rebuilt
// (new lengths, a scan, a copy pass) - only if a record is affected at all.
struct CleanCols {
  uint64_t n;
  const int32_t *refid, *pos, *ref_len;
  const uint16_t *flag;
  uint8_t *mapq;
  const uint8_t *state;
  const uint64_t *cigar_off;
  const uint32_t *cigar;
  int32_t n_ref;
};
// the new CIGAR of record i (out may be null: count only); returns the number of operations, -1: the reference panics, -2: a length the
// 28-bit BAM field cannot hold; *changed = the record is rewritten
__device__ inline int clean_cigar(const CleanCols &m, uint64_t i, uint32_t *out, bool *changed) {
  const uint64_t c0 = m.cigar_off[i], c1 = m.cigar_off[i + 1];
  const int nop = (int)(c1 - c0);
  *changed = false;
  const uint16_t f = m.flag[i];
  const int32_t r = m.refid[i];
  bool clip = false;
  int32_t clip_from = 0, read_len = 0;
  if (!(f & F_UNMAPPED)) {
    int32_t ref_span = 0;
    for (int k = 0; k < nop; k++) {
      const uint32_t op = m.cigar[c0 + k], o = op & 15u;
      const int32_t l = (int32_t)(op >> 4);
      if (op_consumes_ref(o)) ref_span += l;
      if (op_consumes_read(o)) read_len += l;
    }
    // referenceSequenceTable[aln.RNAME] (simple-filters.go:300): a Go map - RNAME '*' or a name the header does not have gives length 0, so
    // a read with the mapped flag and no reference (End() > 0) is soft-clipped from its first base on
    const int32_t length = (r >= 0 && r < m.n_ref) ? m.ref_len[r] : 0, end = m.pos[i] + ref_span - 1;  // Alignment.End, sam/sam-types.go:769-775
    if (end > length) { clip = true; clip_from = length - m.pos[i] + 1; }
  }
  if (!clip) {
    if (out) for (int k = 0; k < nop; k++) out[k] = m.cigar[c0 + k];
    return nop;
  }
  *changed = true;
  int32_t pos = 0;
  clip_from--;
  int n_new = 0;
  for (int k = 0; k < nop; k++) {
    const uint32_t op = m.cigar[c0 + k], o = op & 15u;
    const int32_t end_pos = pos + (op_consumes_read(o) ? (int32_t)(op >> 4) : 0);
    if (end_pos < clip_from) {
      if (out) out[n_new] = op;
      n_new++;
    } else {
      int32_t clipped = read_len + clip_from;
      const int32_t rel = clip_from - pos;
      if (op_consumes_read(o)) {
        if (op_consumes_ref(o)) {
          if (rel > 0) {
            if (rel >= (1 << 28)) return -2;
            if (out) out[n_new] = ((uint32_t)rel << 4) | o;
            n_new++;
          }
        } else {
          clipped += rel;
        }
      } else if (rel != 0) {
        return -1;  // "Unexpected non-0 relative clipping position in CleanSam." (filters/utils.go:93)
      }
      if (clipped < 0 || clipped >= (1 << 28)) return -2;
      if (out) out[n_new] = ((uint32_t)clipped << 4) | OP_S;
      n_new++;
      break;
    }
    pos += end_pos;
  }
  return n_new;
}
__global__ __launch_bounds__(256) void k_clean_count(CleanCols m, uint32_t *__restrict__ newcnt, uint32_t *res /* [0] rewritten records, [1] error bits */) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.n) return;
  uint32_t cnt = (uint32_t)(m.cigar_off[i + 1] - m.cigar_off[i]);
  if (m.state[i] != 2) {  // (records an earlier filter removed never reach CleanSam)
    bool changed;
    const int nn = clean_cigar(m, i, nullptr, &changed);
    if (nn < 0) atomicOr(&res[1], nn == -1 ? 1u : 2u);
    else cnt = (uint32_t)nn;
    if (changed) atomicAdd(&res[0], 1u);
  }
  newcnt[i] = cnt;
}
__global__ __launch_bounds__(256) void k_clean_write(CleanCols m, const uint32_t *__restrict__ newoff, uint32_t total, uint32_t *__restrict__ cigar_new,
                                                     uint64_t *__restrict__ off_new) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > m.n) return;
  if (i == m.n) { off_new[i] = total; return; }
  off_new[i] = newoff[i];
  bool changed;
  if (m.state[i] != 2) (void)clean_cigar(m, i, cigar_new + newoff[i], &changed);
  else {
    const uint64_t c0 = m.cigar_off[i], c1 = m.cigar_off[i + 1];
    for (uint64_t k = c0; k < c1; k++) cigar_new[newoff[i] + (k - c0)] = m.cigar[k];
  }
}
// aln.MAPQ = 0 for unmapped reads - queued once the count pass has shown that no record makes the call fail (a failed call leaves the
// columns as they were: ADVICE r4)
__global__ __launch_bounds__(256) void k_clean_mapq(uint64_t n, const uint16_t *__restrict__ flag, const uint8_t *__restrict__ state, uint8_t *__restrict__ mapq) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && state[i] != 2 && (flag[i] & F_UNMAPPED)) mapq[i] = 0;
}
__global__ __launch_bounds__(256) void k_copy_u64(uint64_t n, const uint64_t *__restrict__ src, uint64_t *__restrict__ dst) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

}  // namespace elp

extern "C" int elp_clean_sam(elp_ctx *c, uint64_t *n_clipped_out) {
  using namespace elp;
  if (!c) return ELP_ERR_ARG;
  std::lock_guard<std::mutex> g(c->stage_mu);
  ELP_HIP(c, hipSetDevice(c->device));
  if (!c->have_header) return set_error(c, ELP_ERR_ARG, "elp_clean_sam: call elp_set_header first");
  if (n_clipped_out) *n_clipped_out = 0;
  const uint64_t n = c->n;
  if (!n) return 0;
  if (c->cigar_ops + n >= 0xFFFFFFFFull) return set_error(c, ELP_ERR_UNSUPPORTED, "elp_clean_sam: more than 2^32 CIGAR operations in one context");
  hipStream_t st = c->stream;
  uint32_t *wk;
  ELP_TRY(scratch(c, 4, 2 * (n + 8) + 16, &wk));
  uint32_t *newcnt = wk, *newoff = wk + n + 8, *res = newoff + n + 8;
  ELP_HIP(c, hipMemsetAsync(res, 0, 8, st));
  CleanCols m{n, c->refid.p, c->pos.p, c->ref_len.p, c->flag.p, c->mapq.p, c->has_sr.p, c->cigar_off.p, c->cigar.p, c->n_ref};
  ELP_LAUNCH(c, "clean_count", k_clean_count, dim3(blocks_for(n, 256)), dim3(256), 0, m, newcnt, res);
  uint32_t hr[2] = {0, 0};
  ELP_HIP(c, hipMemcpyAsync(hr, res, 8, hipMemcpyDeviceToHost, st));
  ELP_HIP(c, elp::stream_wait(st));
  // MAPQ changed, CIGARs may: whatever was derived from them is stale
  c->adapted = c->sorted = c->marked = false;
  if (hr[1] & 1u) return set_error(c, ELP_ERR_DATA, "Unexpected non-0 relative clipping position in CleanSam. (reference: log.Panic, filters/utils.go:93)");
  if (hr[1] & 2u) return set_error(c, ELP_ERR_UNSUPPORTED, "elp_clean_sam: a clipped CIGAR needs an operation length outside the 28 bits of a BAM CIGAR field");
  ELP_LAUNCH(c, "clean_mapq", k_clean_mapq, dim3(blocks_for(n, 256)), dim3(256), 0, n, (const uint16_t *)c->flag.p, (const uint8_t *)c->has_sr.p, c->mapq.p);
  if (n_clipped_out) *n_clipped_out = hr[0];
  if (!hr[0]) { ELP_HIP(c, elp::stream_wait(st)); return 0; }
  uint32_t total = 0;
  ELP_TRY(exclusive_scan_u32(c, newcnt, newoff, n, &total));
  uint32_t *cig_new;
  uint64_t *off_new;
  ELP_TRY(scratch(c, 5, (size_t)total + 64, &cig_new));
  ELP_TRY(scratch(c, 6, n + 8, &off_new));
  ELP_LAUNCH(c, "clean_write", k_clean_write, dim3(blocks_for(n + 1, 256)), dim3(256), 0, m, (const uint32_t *)newoff, total, cig_new, off_new);
  // the rebuilt column replaces the staged one (the offsets first: the kernel above read the old ones)
  ELP_TRY(ensure(c, c->cigar, (size_t)total + 64));
  ELP_HIP(c, hipMemcpyAsync(c->cigar.p, cig_new, (size_t)total * 4, hipMemcpyDeviceToDevice, st));
  ELP_LAUNCH(c, "clean_offsets", k_copy_u64, dim3(blocks_for(n + 1, 256)), dim3(256), 0, n + 1, (const uint64_t *)off_new, c->cigar_off.p);
  ELP_HIP(c, elp::stream_wait(st));
  c->cigar_ops = total;
  return 0;
}

namespace elp {
}  // namespace elp

using namespace elp;

extern "C" {

int elp_filter_records(elp_ctx *c, const elp_predicates *p, uint64_t *n_dropped_out) {
  if (!c || !p) return ELP_ERR_ARG;
  ELP_HIP(c, hipSetDevice(c->device));
  if (p->use_regions && (!p->regions || !p->n_regions)) return set_error(c, ELP_ERR_ARG, "elp_filter_records: use_regions without regions");
  const uint64_t n = c->n;
  unsigned long long dropped = 0, dropped_tagged = 0;
  if (n) {
    // target regions -> device (per refid pointer table)
    std::vector<int32_t *> h_ptr((size_t)c->n_ref, nullptr);
    std::vector<int64_t> h_cnt((size_t)c->n_ref, 0);
    int32_t **d_ptr = nullptr;
    int64_t *d_cnt = nullptr;
    int32_t *pool = nullptr;
    if (p->use_regions) {
      size_t total = 0;
      for (int r = 0; r < c->n_ref; r++) total += (size_t)p->n_regions[r];
      uint8_t *blk;
      ELP_TRY(scratch(c, 5, total * 8 + (size_t)c->n_ref * 16 + 64, &blk));
      pool = reinterpret_cast<int32_t *>(blk);
      d_ptr = reinterpret_cast<int32_t **>(blk + ((total * 8 + 15) & ~(size_t)15));
      d_cnt = reinterpret_cast<int64_t *>(d_ptr + c->n_ref);
      size_t at = 0;
      for (int r = 0; r < c->n_ref; r++) {
        const size_t k = (size_t)p->n_regions[r];
        if (k) ELP_HIP(c, hipMemcpyAsync(pool + 2 * at, p->regions[r], k * 8, hipMemcpyHostToDevice, c->stream));
        h_ptr[r] = pool + 2 * at;
        h_cnt[r] = (int64_t)k;
        at += k;
      }
      if (c->n_ref) {
        ELP_HIP(c, hipMemcpyAsync(d_ptr, h_ptr.data(), (size_t)c->n_ref * sizeof(int32_t *), hipMemcpyHostToDevice, c->stream));
        ELP_HIP(c, hipMemcpyAsync(d_cnt, h_cnt.data(), (size_t)c->n_ref * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
      }
    }
    unsigned long long *cnt;
    ELP_TRY(scratch(c, 6, 4, &cnt));
    ELP_HIP(c, hipMemsetAsync(cnt, 0, 16, c->stream));
    FilterCols m{n, c->refid.p, c->pos.p, c->flag.p, c->mapq.p, c->cigar_off.p, c->cigar.p, c->has_sr.p, c->n_ref, d_ptr, d_cnt};
    ELP_LAUNCH(c, "filter_records", k_filter_records, dim3(blocks_for(n, 256)), dim3(256), 0, m, *p, cnt);
    unsigned long long both[2] = {0, 0};
    ELP_HIP(c, hipMemcpyAsync(both, cnt, 16, hipMemcpyDeviceToHost, c->stream));
    ELP_HIP(c, elp::stream_wait(c->stream));
    dropped = both[0];
    dropped_tagged = both[1];
  }
  c->n_sr += dropped;  // they leave the output like the tagged copies do
  c->n_filtered += dropped + dropped_tagged;  // state-2 records of either origin: none of them is a duplicate-marking candidate
  c->adapted = c->sorted = c->marked = false;
  if (n_dropped_out) *n_dropped_out = dropped;
  return 0;
}

// cgo-friendly form (no pointers to pointers): the regions of all contigs behind each other, region_off[r] .. region_off[r + 1] = the
// [k][2] rows of refid r (n_ref + 1 offsets, in intervals); regions NULL = no target-region test
int elp_filter_records_flat(elp_ctx *c, int remove_unmapped, int remove_unmapped_strict, int min_mapq, int remove_non_exact, int remove_duplicates,
                            const int32_t *regions, const int64_t *region_off, uint64_t *n_rejected_out) {
  if (!c) return ELP_ERR_ARG;
  if (regions && !region_off) return set_error(c, ELP_ERR_ARG, "elp_filter_records_flat: regions without region_off");
  std::vector<const int32_t *> ptr((size_t)c->n_ref + 1, nullptr);
  std::vector<int64_t> cnt((size_t)c->n_ref + 1, 0);
  if (regions)
    for (int r = 0; r < c->n_ref; r++) {
      if (region_off[r + 1] < region_off[r]) return set_error(c, ELP_ERR_ARG, "elp_filter_records_flat: region_off must not decrease");
      ptr[r] = regions + 2 * region_off[r];
      cnt[r] = region_off[r + 1] - region_off[r];
    }
  elp_predicates p;
  p.remove_unmapped = remove_unmapped; p.remove_unmapped_strict = remove_unmapped_strict; p.min_mapq = min_mapq; p.remove_non_exact = remove_non_exact;
  p.remove_duplicates = remove_duplicates; p.use_regions = regions ? 1 : 0; p.regions = ptr.data(); p.n_regions = cnt.data();
  return elp_filter_records(c, &p, n_rejected_out);
}

int elp_split_classify(elp_ctx *c, const int32_t *group_of_ref, int32_t n_groups, uint16_t *split_out, uint8_t *spread_out, uint64_t *counts_out) {
  if (!c || !group_of_ref || n_groups < 0 || n_groups > 8000) return set_error(c, ELP_ERR_ARG, "elp_split_classify: bad arguments");
  ELP_HIP(c, hipSetDevice(c->device));
  const uint64_t n = c->n;
  const size_t nc = (size_t)n_groups + 2;
  uint8_t *blk;
  ELP_TRY(scratch(c, 5, (size_t)c->n_ref * 4 + nc * 8 + n * 3 + 64, &blk));
  int32_t *d_gof = reinterpret_cast<int32_t *>(blk);
  unsigned long long *d_cnt = reinterpret_cast<unsigned long long *>(blk + (((size_t)c->n_ref * 4 + 15) & ~(size_t)15));
  uint16_t *d_split = reinterpret_cast<uint16_t *>(d_cnt + nc);
  uint8_t *d_spread = reinterpret_cast<uint8_t *>(d_split + n + 8);
  if (c->n_ref) ELP_HIP(c, hipMemcpyAsync(d_gof, group_of_ref, (size_t)c->n_ref * 4, hipMemcpyHostToDevice, c->stream));
  ELP_HIP(c, hipMemsetAsync(d_cnt, 0, nc * 8, c->stream));
  if (n) {
    const unsigned grid = std::min(blocks_for(n, 256), 2048u);
    ELP_LAUNCH(c, "split_classify", k_split_classify, dim3(grid), dim3(256), nc * sizeof(unsigned int), n, (const int32_t *)c->refid.p,
               (const int32_t *)c->next_refid.p, (const int32_t *)d_gof, c->n_ref, n_groups, d_split, d_spread, d_cnt, c->split.p);
    // the split ids are part of every duplicate-marking key; records with equal keys share their contig, hence their split: no result changes
    c->max_split = std::max<uint32_t>(c->max_split, (uint32_t)n_groups);
    c->marked = false;
    if (split_out) ELP_HIP(c, hipMemcpyAsync(split_out, d_split, n * 2, hipMemcpyDeviceToHost, c->stream));
    if (spread_out) ELP_HIP(c, hipMemcpyAsync(spread_out, d_spread, n, hipMemcpyDeviceToHost, c->stream));
  }
  std::vector<unsigned long long> h(nc);
  ELP_HIP(c, hipMemcpyAsync(h.data(), d_cnt, nc * 8, hipMemcpyDeviceToHost, c->stream));
  ELP_HIP(c, elp::stream_wait(c->stream));
  if (counts_out)
    for (size_t k = 0; k < nc; k++) counts_out[k] = h[k];
  return 0;
}

int elp_merge_spread(elp_ctx *groups, elp_ctx *spread, uint64_t *slot_of_spread_out) {
  if (!groups || !spread || groups == spread) return ELP_ERR_ARG;
  uint64_t *slots = nullptr;
  ELP_TRY(merge_spread_slots(groups, spread, &slots));
  const uint64_t ns = spread->n - spread->n_sr;
  if (ns) ELP_HIP(groups, hipMemcpyAsync(slot_of_spread_out, slots, ns * 8, hipMemcpyDeviceToHost, groups->stream));
  ELP_HIP(groups, elp::stream_wait(groups->stream));
  return 0;
}

}  // extern "C"
