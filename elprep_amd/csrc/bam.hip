// bam.hip — the data formats either side of the path, on the device: inflated BAM records in, BAM records out.
//
// In  (elp_stage_bam): parseBamAlignment (sam/bam-files.go:299-400) for the fields the path reads.  The host hands over the
//     bytes of inflated BGZF blocks (a run of whole alignment records: block_size, 32 fixed bytes, read_name, cigar, seq, qual,
//     tags - SAMv1 4.2); they cross PCIe once, as they are (pinned double buffer, copy overlapped with the host's memcpy into
//     it), and kernels cut them into the SoA columns: fixed fields by one thread per record (POS/PNEXT + 1, QNAME without its NUL,
//     RG:Z looked up in the header's read-group ids, sr tag -> record state), variable-length parts by one wavefront per record.
//     Only the walk along the block_size chain stays on the host (a pointer chase: 4 bytes read per record).
// Out (elp_emit_sorted_bam): formatBamAlignment (sam/bam-files.go:635-737) for the records in coordinate order
//     (elp_sort_coordinate's permutation, without the sr-tagged / filtered records): fixed fields from the columns with the
//     duplicate flags of elp_mark_duplicates, bin() recomputed (:443-468), read name, CIGAR, the ORIGINAL 4-bit bases, the
//     recalibrated qualities of elp_bqsr_apply, and the tags re-encoded as formatBamTag does (:481-632: integers in the
//     smallest type that holds them, unsigned if >= 0; everything else as it came).  This is the "payload permutation" the
//     reference does on pointers, done as one gather in HBM; BGZF deflate stays with the host.
#include <thread>

#include "common.hpp"

namespace elp {

__device__ __forceinline__ uint32_t ld_u16(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ uint32_t ld_u32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
__device__ __forceinline__ void st_u16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
__device__ __forceinline__ void st_u32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

// bytes of the value of one optional field of type `t` at p (end = end of the record); 0 = malformed
__device__ inline uint32_t tag_value_size(uint8_t t, const uint8_t *p, const uint8_t *end) {
  switch (t) {
    case 'A': case 'c': case 'C': return 1;
    case 's': case 'S': return 2;
    case 'i': case 'I': case 'f': return 4;
    case 'Z': case 'H': {
      uint32_t k = 0;
      while (p + k < end && p[k] != 0) k++;
      return p + k < end ? k + 1 : 0;
    }
    case 'B': {
      if (p + 5 > end) return 0;
      const uint8_t st = p[0];
      const uint32_t cnt = ld_u32(p + 1);
      const uint32_t es = (st == 'c' || st == 'C') ? 1 : ((st == 's' || st == 'S') ? 2 : ((st == 'i' || st == 'I' || st == 'f') ? 4 : 0));
      if (!es) return 0;
      const uint64_t sz = 5ull + (uint64_t)cnt * es;  // 64-bit: a malformed count must not wrap into a small size
      return sz > (uint64_t)(end - p) ? 0u : (uint32_t)sz;
    }
    default: return 0;
  }
}
__device__ __forceinline__ bool tag_is_int(uint8_t t) { return t == 'c' || t == 'C' || t == 's' || t == 'S' || t == 'i' || t == 'I'; }
__device__ inline long long tag_int_value(uint8_t t, const uint8_t *p) {
  switch (t) {
    case 'c': return (long long)(int8_t)p[0];
    case 'C': return (long long)p[0];
    case 's': return (long long)(int16_t)ld_u16(p);
    case 'S': return (long long)ld_u16(p);
    case 'i': return (long long)(int32_t)ld_u32(p);
    default: return (long long)ld_u32(p);
  }
}
// formatBamTag's integer rule (:492-525): type and size of the re-encoded value
__device__ inline uint32_t int_out(long long v, uint8_t *type) {
  if (v < 0) {
    if (v >= -128) { *type = 'c'; return 1; }
    if (v >= -32768) { *type = 's'; return 2; }
    *type = 'i';
    return 4;
  }
  if (v <= 255) { *type = 'C'; return 1; }
  if (v <= 65535) { *type = 'S'; return 2; }
  *type = 'I';
  return 4;
}

struct BamIn {
  const uint8_t *raw;        // records of this piece (device copy); rec_off are offsets from raw
  const uint64_t *rec_off;   // n + 1
  uint32_t n;
  uint64_t at;               // first staging index of the piece
  int32_t *refid, *pos, *next_refid, *pnext, *tlen;
  uint16_t *flag, *rgid, *split;
  uint8_t *mapq, *state;
  uint32_t *l_seq;
  uint32_t *len_q, *len_c, *len_s, *len_l;  // per record: QNAME bytes, CIGAR ops, SEQ bytes, QUAL bytes
  const uint8_t *rg_ids;     // header read-group ids, concatenated
  const uint32_t *rg_off;    // n_rg + 1
  int32_t n_rg, n_ref;
  uint16_t split_id;
  uint32_t *stats;           // [0] max QNAME length, [1] max l_seq, [2] max POS, [3] sr-tagged records, [4] error bits
};

__global__ __launch_bounds__(256) void k_bam_fixed(BamIn m) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  bool sr = false;
  if (r < m.n) {
    const uint8_t *p = m.raw + m.rec_off[r];
    const uint32_t bs = ld_u32(p);
    const uint8_t *rec = p + 4, *end = rec + bs;
    const uint64_t i = m.at + r;
    const int32_t refid = (int32_t)ld_u32(rec), pos0 = (int32_t)ld_u32(rec + 4);
    const uint32_t l_name = rec[8], n_cig = ld_u16(rec + 12), l_seq = ld_u32(rec + 16);
    const uint32_t seqb = (l_seq + 1) >> 1;
    const uint64_t fixed = 32ull + l_name + 4ull * n_cig + seqb + l_seq;
    uint32_t err = 0;
    if (bs < 32 || l_name == 0 || fixed > bs || refid >= m.n_ref || (uint64_t)bs + 4 != m.rec_off[r + 1] - m.rec_off[r]) err |= 1u;
    m.refid[i] = refid < 0 ? -1 : refid;                 // RNAME "*" (:322-326)
    m.pos[i] = pos0 + 1;                                 // :328
    m.mapq[i] = rec[9];
    m.flag[i] = (uint16_t)ld_u16(rec + 14);
    m.l_seq[i] = l_seq;
    const int32_t nref = (int32_t)ld_u32(rec + 20);
    m.next_refid[i] = nref < 0 ? -1 : nref;
    m.pnext[i] = (int32_t)ld_u32(rec + 24) + 1;
    m.tlen[i] = (int32_t)ld_u32(rec + 28);
    m.split[i] = m.split_id;
    m.len_q[r] = err ? 0 : l_name - 1;                   // QNAME without the NUL (:352)
    m.len_c[r] = err ? 0 : n_cig;
    m.len_s[r] = err ? 0 : seqb;
    m.len_l[r] = err ? 0 : l_seq;
    // optional fields: RG:Z -> dense id of the header's read groups, sr -> record state (:373-396)
    uint32_t rg = ELP_NIL16;
    if (!err) {
      const uint8_t *t = rec + fixed;
      while (t + 3 <= end) {
        const uint8_t k0 = t[0], k1 = t[1], ty = t[2];
        const uint8_t *v = t + 3;
        const uint32_t sz = tag_value_size(ty, v, end);
        if (!sz || v + sz > end) { err |= 2u; break; }
        if (k0 == 's' && k1 == 'r') sr = true;
        if (k0 == 'C' && k1 == 'G' && ty == 'B') err |= 4u;  // CIGAR in a tag (> 65535 operations): not supported
        if (k0 == 'R' && k1 == 'G' && ty == 'Z') {
          const uint32_t l = sz - 1;
          rg = 0xFFFEu;  // a read group the header does not know
          for (int g = 0; g < m.n_rg; g++) {
            const uint32_t o = m.rg_off[g], gl = m.rg_off[g + 1] - o;
            if (gl != l) continue;
            bool eq = true;
            for (uint32_t b = 0; b < l; b++) eq &= m.rg_ids[o + b] == v[b];
            if (eq) { rg = (uint32_t)g; break; }
          }
        }
        t = v + sz;
      }
      if (t != end && !(err & 2u)) err |= 2u;
    }
    if (rg == 0xFFFEu) { err |= 8u; rg = ELP_NIL16; }
    m.rgid[i] = (uint16_t)rg;
    m.state[i] = sr ? 1 : 0;
    atomicMax(&m.stats[0], err ? 0u : l_name - 1);
    atomicMax(&m.stats[1], l_seq);
    atomicMax(&m.stats[2], (uint32_t)(pos0 + 1));
    if (err) atomicOr(&m.stats[4], err);
  }
  const unsigned long long b = __ballot(sr);
  if ((threadIdx.x & 63) == 0 && b) atomicAdd(&m.stats[3], (uint32_t)__popcll(b));
}

struct BamPay {
  const uint8_t *raw;
  const uint64_t *rec_off;
  uint32_t n;
  uint64_t at;
  const uint32_t *sc_q, *sc_c, *sc_s, *sc_l;  // exclusive scans of the lengths
  const uint32_t *len_q, *len_c, *len_s, *len_l;
  uint64_t base_q, base_c, base_s, base_l;     // column fill levels in front of the piece
  uint64_t *qname_off, *cigar_off, *seq_off, *qual_off;
  uint8_t *qname, *seq4, *qual;
  uint32_t *cigar;
};
// one wavefront per record: lanes copy the variable-length parts byte by byte (coalesced)
__global__ __launch_bounds__(256) void k_bam_payload(BamPay m) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t r = wave; r < m.n; r += nwaves) {
    const uint8_t *rec = m.raw + m.rec_off[r] + 4;
    const uint32_t lq = m.len_q[r], lc = m.len_c[r], ls = m.len_s[r], ll = m.len_l[r];
    const uint64_t oq = m.base_q + m.sc_q[r], oc = m.base_c + m.sc_c[r], os = m.base_s + m.sc_s[r], ol = m.base_l + m.sc_l[r];
    if (lane == 0) {
      const uint64_t i = m.at + r;
      m.qname_off[i] = oq; m.cigar_off[i] = oc; m.seq_off[i] = os; m.qual_off[i] = ol;
      if (r + 1 == m.n) { m.qname_off[i + 1] = oq + lq; m.cigar_off[i + 1] = oc + lc; m.seq_off[i + 1] = os + ls; m.qual_off[i + 1] = ol + ll; }
    }
    const uint8_t *src = rec + 32;
    for (uint32_t k = lane; k < lq; k += 64) m.qname[oq + k] = src[k];
    src += lq + 1;
    for (uint32_t k = lane; k < lc; k += 64) m.cigar[oc + k] = ld_u32(src + 4 * k);
    src += 4 * lc;
    for (uint32_t k = lane; k < ls; k += 64) m.seq4[os + k] = src[k];
    src += ls;
    for (uint32_t k = lane; k < ll; k += 64) m.qual[ol + k] = src[k];
  }
}

// ---------------------------------------------------------------- out
struct BamOut {
  uint64_t n_out;
  const uint32_t *perm;
  const uint8_t *raw;
  const uint64_t *raw_off;   // per staged record: offset of its BAM record in raw (n + 1)
  const int32_t *refid, *pos, *next_refid, *pnext, *tlen;
  const uint16_t *flag;
  const uint8_t *mapq;
  const uint32_t *l_seq;
  const uint64_t *qname_off, *cigar_off, *qual_off;
  const uint8_t *qname, *qual;
  const uint32_t *cigar;
};
// size of output record k (block_size field included); *tags_at = offset of the tags inside the input record
__device__ inline uint32_t out_size(const BamOut &m, uint32_t i, uint32_t *err) {
  const uint8_t *p = m.raw + m.raw_off[i];
  const uint32_t bs = ld_u32(p);
  const uint8_t *rec = p + 4, *end = rec + bs;
  const uint32_t l_name = rec[8], n_cig = ld_u16(rec + 12), l_seq = ld_u32(rec + 16);
  const uint64_t fixed = 32ull + l_name + 4ull * n_cig + ((l_seq + 1) >> 1) + l_seq;
  // (the record goes out with the CIGAR column's operations - elp_clean_sam may have rewritten them -, the tags come from the staged bytes)
  const uint64_t n_cig_out = m.cigar_off[i + 1] - m.cigar_off[i];
  uint32_t size = 4 + (uint32_t)(fixed + 4ull * n_cig_out - 4ull * n_cig);
  const uint8_t *t = rec + fixed;
  while (t + 3 <= end) {
    const uint8_t ty = t[2];
    const uint8_t *v = t + 3;
    const uint32_t sz = tag_value_size(ty, v, end);
    if (!sz) { *err |= 2u; break; }
    if (ty == 'H') *err |= 16u;  // parseBamByteArray looks for the character '0' as the terminator (:203): not reproduced
    uint8_t ot;
    size += 3 + (tag_is_int(ty) ? int_out(tag_int_value(ty, v), &ot) : sz);
    t = v + sz;
  }
  return size;
}
// Output record k of a MERGED stream (elp_emit_merged_bam) comes from one of two contexts: src[k] = rank of the record in the first
// context's sorted output, or MERGE_SECOND | its rank in the second's.  src == nullptr: one context, output record k = perm[k].
constexpr uint32_t MERGE_SECOND = 0x80000000u;
__device__ __forceinline__ const BamOut &out_source(const BamOut &m, const BamOut &m2, const uint32_t *__restrict__ src, uint64_t k, uint32_t *i) {
  if (!src) { *i = m.perm[k]; return m; }
  const uint32_t s = src[k];
  const BamOut &mm = (s & MERGE_SECOND) ? m2 : m;
  *i = mm.perm[s & ~MERGE_SECOND];
  return mm;
}
__global__ __launch_bounds__(256) void k_bam_out_sizes(BamOut m, BamOut m2, const uint32_t *__restrict__ src, uint64_t k0, uint32_t cnt, uint32_t *__restrict__ sizes,
                                                       uint32_t *err) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cnt) return;
  uint32_t e = 0, i;
  const BamOut &mm = out_source(m, m2, src, k0 + j, &i);
  sizes[j] = out_size(mm, i, &e);
  if (e) atomicOr(err, e);
}
// Alignment.bin(), sam/bam-files.go:443-468
__device__ inline uint32_t reg2bin(int32_t beg, int32_t end) {
  if (beg >> 14 == end >> 14) return (uint32_t)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (uint32_t)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (uint32_t)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (uint32_t)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (uint32_t)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}
// one wavefront per output record; `out` = this chunk's buffer, offs = exclusive scan of the chunk's sizes
__global__ __launch_bounds__(256) void k_bam_out_emit(BamOut m_first, BamOut m_second, const uint32_t *__restrict__ src, uint64_t k0, uint32_t cnt,
                                                      const uint32_t *__restrict__ offs, uint8_t *__restrict__ out) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
  for (uint32_t j = wave; j < cnt; j += nwaves) {
    uint32_t i;
    const BamOut &m = out_source(m_first, m_second, src, k0 + j, &i);
    const uint8_t *p = m.raw + m.raw_off[i];
    const uint32_t bs = ld_u32(p);
    const uint8_t *rec = p + 4, *end = rec + bs;
    uint8_t *o = out + offs[j];
    const uint32_t l_name = rec[8], l_seq = m.l_seq[i];
    const uint64_t c0 = m.cigar_off[i];
    const uint32_t n_cig = (uint32_t)(m.cigar_off[i + 1] - c0), seqb = (l_seq + 1) >> 1;
    const uint32_t fixed = 32 + l_name + 4 * n_cig + seqb + l_seq;
    if (lane == 0) {
      const uint16_t f = m.flag[i];
      const int32_t beg = m.pos[i] - 1;
      int32_t e2 = beg;
      if (!(f & F_UNMAPPED)) {
        for (uint32_t k = 0; k < n_cig; k++) {
          const uint32_t c = m.cigar[c0 + k];
          if (op_consumes_ref(c & 0xF)) e2 += (int32_t)(c >> 4);
        }
        e2--;
      }
      st_u32(o + 4, (uint32_t)m.refid[i]);
      st_u32(o + 8, (uint32_t)beg);
      o[12] = (uint8_t)l_name;
      o[13] = m.mapq[i];
      st_u16(o + 14, reg2bin(beg, e2));
      st_u16(o + 16, n_cig);
      st_u16(o + 18, f);
      st_u32(o + 20, l_seq);
      st_u32(o + 24, (uint32_t)m.next_refid[i]);
      st_u32(o + 28, (uint32_t)(m.pnext[i] - 1));
      st_u32(o + 32, (uint32_t)m.tlen[i]);
    }
    {
      // tags, re-encoded as formatBamTag does.  The WAVE walks them (round 6: lane 0 did, byte by byte - a dependent load per byte of
      // every string while 63 lanes waited): every lane reads the same tag header, a string's end is found by all lanes at once, a value
      // that stays as it is is copied by all lanes; only an integer's new form is written by lane 0.
      uint8_t *w = o + 4 + fixed;
      const uint8_t *t = rec + (32ull + l_name + 4ull * ld_u16(rec + 12) + ((ld_u32(rec + 16) + 1) >> 1) + ld_u32(rec + 16));
      while (t + 3 <= end) {
        const uint8_t ty = t[2];
        const uint8_t *v = t + 3;
        uint32_t sz;
        if (ty == 'Z' || ty == 'H') {  // (tag_value_size's loop, 64 bytes a step)
          sz = 0;
          for (uint32_t at = 0;; at += 64) {
            const uint8_t *q = v + at + lane;
            const bool in = q < end;
            const unsigned long long zero = __ballot(in && *q == 0);
            if (zero) { sz = at + (uint32_t)__builtin_ctzll(zero) + 1u; break; }
            if (__ballot(!in)) break;  // no NUL in front of the record's end: malformed, as tag_value_size says
          }
        } else {
          sz = tag_value_size(ty, v, end);
        }
        if (!sz) break;
        if (tag_is_int(ty)) {
          const long long val = tag_int_value(ty, v);
          uint8_t ot;
          const uint32_t os = int_out(val, &ot);
          if (lane == 0) {
            w[0] = t[0]; w[1] = t[1]; w[2] = ot;
            for (uint32_t b = 0; b < os; b++) w[3 + b] = (uint8_t)((unsigned long long)val >> (8 * b));
          }
          w += 3 + os;
        } else {
          for (uint32_t b = lane; b < 3 + sz; b += 64) w[b] = t[b];
          w += 3 + sz;
        }
        t = v + sz;
      }
      if (lane == 0) st_u32(o, (uint32_t)(w - o - 4));  // block_size
    }
    // read name (+ NUL), CIGAR, the original bases, the qualities as they are now
    uint8_t *w = o + 36;
    const uint64_t q0 = m.qname_off[i];
    for (uint32_t k = lane; k < l_name; k += 64) w[k] = k + 1 < l_name ? m.qname[q0 + k] : (uint8_t)0;
    w += l_name;
    for (uint32_t k = lane; k < n_cig; k += 64) st_u32(w + 4 * k, m.cigar[c0 + k]);
    w += 4 * n_cig;
    const uint8_t *seq_in = rec + 32 + l_name + 4 * ld_u16(rec + 12);
    for (uint32_t k = lane; k < seqb; k += 64) w[k] = seq_in[k];
    w += seqb;
    const uint64_t l0 = m.qual_off[i];
    for (uint32_t k = lane; k < l_seq; k += 64) w[k] = m.qual[l0 + k];
  }
}

int stage_reserve(elp_ctx *c, uint64_t n, uint64_t qb, uint64_t co, uint64_t sb, uint64_t lb);  // ctx.hip

// the columns of n_rec records whose inflated bytes lie in c->raw and whose offsets (n_rec + 1, into c->raw) lie at c->raw_off[c->n ..]:
// fixed fields, the four scans, payload copies, SEQ recoding, the commit.  piece_bytes bounds the variable-length columns of the piece;
// raw_end = the first byte of c->raw behind the last record.  Shared by elp_stage_bam and elp_stage_bgzf (bgzf.hip).
int stage_bam_columns(elp_ctx *c, uint32_t n_rec, uint64_t piece_bytes, uint64_t raw_end, uint64_t max_raw_rec, uint16_t split_id) {
  hipStream_t st = c->stream;
  // columns
  ELP_TRY(stage_reserve(c, c->n + n_rec, c->qname_bytes + piece_bytes, c->cigar_ops + piece_bytes / 4, c->seq_bytes + piece_bytes, c->qual_bytes + piece_bytes));
  uint32_t *wk;
  ELP_TRY(scratch(c, 4, (size_t)8 * (n_rec + 8) + 16, &wk));
  const size_t np = (size_t)n_rec + 8;
  uint32_t *len_q = wk, *len_c = wk + np, *len_s = wk + 2 * np, *len_l = wk + 3 * np, *sc_q = wk + 4 * np, *sc_c = wk + 5 * np, *sc_s = wk + 6 * np,
           *sc_l = wk + 7 * np, *stats = wk + 8 * np;
  ELP_HIP(c, hipMemsetAsync(stats, 0, 32, st));
  BamIn in{c->raw.p, c->raw_off.p + c->n, n_rec, c->n, c->refid.p, c->pos.p, c->next_refid.p, c->pnext.p, c->tlen.p, c->flag.p, c->rgid.p, c->split.p,
           c->mapq.p, c->has_sr.p, c->l_seq.p, len_q, len_c, len_s, len_l, c->rg_ids.p, c->rg_ids_off.p, c->n_rg, c->n_ref, split_id, stats};
  ELP_LAUNCH(c, "stage_bam_fixed", k_bam_fixed, dim3(blocks_for(n_rec, 256)), dim3(256), 0, in);
  uint32_t tq = 0, tc = 0, ts = 0, tl = 0;
  ELP_TRY(exclusive_scan_u32(c, len_q, sc_q, n_rec, &tq));
  ELP_TRY(exclusive_scan_u32(c, len_c, sc_c, n_rec, &tc));
  ELP_TRY(exclusive_scan_u32(c, len_s, sc_s, n_rec, &ts));
  ELP_TRY(exclusive_scan_u32(c, len_l, sc_l, n_rec, &tl));
  uint32_t hs[8];
  ELP_HIP(c, hipMemcpyAsync(hs, stats, 32, hipMemcpyDeviceToHost, st));
  ELP_HIP(c, elp::stream_wait(st));
  if (hs[4]) {
    if (hs[4] & 8u) return set_error(c, ELP_ERR_ARG, "elp_stage_bam: an RG:Z tag names a read group that is not in the header");
    if (hs[4] & 4u) return set_error(c, ELP_ERR_UNSUPPORTED, "elp_stage_bam: CIGAR in a CG:B tag (more than 65535 operations)");
    return set_error(c, ELP_ERR_DATA, "elp_stage_bam: malformed alignment record (bits %u)", hs[4]);
  }
  if (hs[0] > elp_ctx::MAX_QNAME) return set_error(c, ELP_ERR_UNSUPPORTED, "QNAME of %u bytes (limit %u)", hs[0], elp_ctx::MAX_QNAME);
  if (hs[1] > 0x3FFFFFu) return set_error(c, ELP_ERR_UNSUPPORTED, "record with more than 4194303 bases");
  BamPay pay{c->raw.p, c->raw_off.p + c->n, n_rec, c->n, sc_q, sc_c, sc_s, sc_l, len_q, len_c, len_s, len_l, c->qname_bytes, c->cigar_ops, c->seq_bytes,
             c->qual_bytes, c->qname_off.p, c->cigar_off.p, c->seq_off.p, c->qual_off.p, c->qname.p, c->seq4.p, c->qual.p, c->cigar.p};
  if (n_rec) {
    const unsigned grid = std::min<unsigned>(blocks_for((uint64_t)n_rec * 64, 256), (unsigned)c->n_cu * 32);
    ELP_LAUNCH(c, "stage_bam_payload", k_bam_payload, dim3(grid), dim3(256), 0, pay);
    if (ts) ELP_TRY(stage_recode_seq(c, c->seq_bytes, ts));
  }
  c->n += n_rec; c->raw_n += n_rec; c->raw_bytes = raw_end;
  c->qname_bytes += tq; c->cigar_ops += tc; c->seq_bytes += ts; c->qual_bytes += tl;
  c->n_sr += hs[3];
  c->max_qname_len = std::max(c->max_qname_len, hs[0]);
  c->max_l_seq = std::max(c->max_l_seq, hs[1]);
  c->max_pos = std::max(c->max_pos, hs[2]);
  c->max_split = std::max<uint32_t>(c->max_split, split_id);
  c->max_raw_rec = std::max(c->max_raw_rec, max_raw_rec);
  return 0;
}

}  // namespace elp

using namespace elp;

extern "C" {

int elp_set_read_group_ids(elp_ctx *c, const char *const *ids) {
  if (!c || !c->have_header || (c->n_rg && !ids)) return set_error(c, ELP_ERR_ARG, "elp_set_read_group_ids: call elp_set_header first");
  ELP_HIP(c, hipSetDevice(c->device));
  std::vector<uint8_t> cat;
  std::vector<uint32_t> off(1, 0);
  for (int g = 0; g < c->n_rg; g++) {
    const size_t l = strlen(ids[g]);
    cat.insert(cat.end(), ids[g], ids[g] + l);
    off.push_back((uint32_t)cat.size());
  }
  ELP_TRY(ensure(c, c->rg_ids, cat.size() + 16));
  ELP_TRY(ensure(c, c->rg_ids_off, off.size() + 4));
  if (!cat.empty()) ELP_HIP(c, hipMemcpyAsync(c->rg_ids.p, cat.data(), cat.size(), hipMemcpyHostToDevice, c->stream));
  ELP_HIP(c, hipMemcpyAsync(c->rg_ids_off.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, c->stream));
  ELP_HIP(c, elp::stream_wait(c->stream));
  c->have_rg_ids = true;
  return 0;
}

// cgo-friendly form: the ids behind each other, id_off[g] .. id_off[g + 1] = the bytes of read group g (n_rg + 1 offsets)
int elp_set_read_group_ids_flat(elp_ctx *c, const uint8_t *ids, const uint32_t *id_off) {
  if (!c || !c->have_header || (c->n_rg && (!ids || !id_off))) return set_error(c, ELP_ERR_ARG, "elp_set_read_group_ids_flat: call elp_set_header first");
  std::vector<std::string> str((size_t)c->n_rg);
  std::vector<const char *> ptr((size_t)c->n_rg + 1, nullptr);
  for (int g = 0; g < c->n_rg; g++) {
    if (id_off[g + 1] < id_off[g]) return set_error(c, ELP_ERR_ARG, "elp_set_read_group_ids_flat: id_off must not decrease");
    str[g].assign(reinterpret_cast<const char *>(ids) + id_off[g], id_off[g + 1] - id_off[g]);
    ptr[g] = str[g].c_str();
  }
  return elp_set_read_group_ids(c, ptr.data());
}

void *elp_pinned_alloc(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}
void elp_pinned_free(void *p) {
  if (p) (void)hipHostFree(p);
}

int elp_stage_bam(elp_ctx *c, const uint8_t *bytes, uint64_t n_bytes, const uint64_t *rec_off, uint64_t n_records, uint16_t split_id) {
  if (!c || (!bytes && n_bytes)) return ELP_ERR_ARG;
  if (rec_off && (n_records == 0 ? n_bytes != 0 : (rec_off[0] != 0 || rec_off[n_records] != n_bytes)))
    return set_error(c, ELP_ERR_ARG, "elp_stage_bam: rec_off must run from 0 to n_bytes");
  std::lock_guard<std::mutex> g(c->stage_mu);
  ELP_HIP(c, hipSetDevice(c->device));
  if (!c->have_header) return set_error(c, ELP_ERR_ARG, "elp_stage_bam: call elp_set_header first");
  if (c->n_rg && !c->have_rg_ids) return set_error(c, ELP_ERR_ARG, "elp_stage_bam: call elp_set_read_group_ids first");
  if (c->n != c->raw_n) return set_error(c, ELP_ERR_ARG, "elp_stage_bam: the context already holds records staged with elp_stage");
  hipStream_t st = c->stream;
  // pieces are committed one at a time: whatever was derived from the records staged so far is invalid from here on, also if a later
  // piece fails
  c->adapted = c->sorted = c->marked = false;
  c->have_qual_present = false;
  c->have_snapshot = false;
  c->flat_index_n = 0;
  c->uniform_n = ~0ull;
  if (!c->copy_stream) ELP_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  // is the caller's buffer page-locked (elp_pinned_alloc / hipHostRegister)?  then the DMA engine reads it directly
  hipPointerAttribute_t pa;
  const bool pinned = n_bytes && hipPointerGetAttributes(&pa, bytes) == hipSuccess && pa.type == hipMemoryTypeHost;
  (void)hipGetLastError();
  constexpr uint64_t PIECE = 256ull << 20;  // device work unit: <= 256 MiB of records (u32 scans, bounded scratch)
  constexpr size_t BOUNCE = 32ull << 20;
  if (!pinned && n_bytes && !c->bounce[0]) {
    for (int k = 0; k < 2; k++) {
      ELP_HIP(c, hipHostMalloc(&c->bounce[k], BOUNCE, hipHostMallocDefault));
      ELP_HIP(c, hipEventCreateWithFlags(&c->bounce_ev[k], hipEventDisableTiming));
    }
  }
  uint64_t at_byte = 0, at_rec = 0;
  uint64_t max_raw_rec = c->max_raw_rec;
  std::vector<uint64_t> off;
  while (at_byte < n_bytes) {
    // record starts of this piece: given by the caller (a BAM reader knows where every record it hands over begins), else by a walk
    // along the block_size chain (a dependent load per record: the one serial part, ~100 ns per record when the bytes are cold)
    off.clear();
    uint64_t p = at_byte;
    if (rec_off) {
      while (at_rec < n_records && rec_off[at_rec] - at_byte < PIECE) {
        const uint64_t o = rec_off[at_rec], e = rec_off[at_rec + 1];
        if (e < o + 36 || e > n_bytes) return set_error(c, ELP_ERR_DATA, "elp_stage_bam: bad record offsets at record %llu", (unsigned long long)at_rec);
        off.push_back(o - at_byte);
        at_rec++;
      }
      p = rec_off[at_rec];
    } else {
      while (p < n_bytes && p - at_byte < PIECE) {
        if (p + 4 > n_bytes) return set_error(c, ELP_ERR_DATA, "elp_stage_bam: truncated record at byte %llu", (unsigned long long)p);
        uint32_t bs;
        memcpy(&bs, bytes + p, 4);
        if (bs < 32 || p + 4 + bs > n_bytes) return set_error(c, ELP_ERR_DATA, "elp_stage_bam: bad block_size %u at byte %llu", bs, (unsigned long long)p);
        off.push_back(p - at_byte);
        p += 4ull + bs;
      }
    }
    const uint64_t piece_bytes = p - at_byte;
    const uint32_t n_rec = (uint32_t)off.size();
    off.push_back(piece_bytes);
    for (uint32_t k = 0; k < n_rec; k++) max_raw_rec = std::max<uint64_t>(max_raw_rec, off[k + 1] - off[k]);
    if (c->n + n_rec > 0xFFFFFFF0ull) return set_error(c, ELP_ERR_UNSUPPORTED, "more than 2^32-16 records per context");
    // raw bytes + record offsets into HBM (kept: elp_emit_sorted_bam reads bases and tags from them)
    ELP_TRY(ensure(c, c->raw, c->raw_bytes + piece_bytes + 64, true, c->raw_bytes));
    ELP_TRY(ensure(c, c->raw_off, c->n + n_rec + 2, true, c->n + 1));
    uint8_t *d_raw = c->raw.p + c->raw_bytes;
    if (pinned) {
      ELP_HIP(c, hipMemcpyAsync(d_raw, bytes + at_byte, piece_bytes, hipMemcpyHostToDevice, st));
    } else {
      // bounce: while the DMA engine moves buffer k, this thread fills buffer 1 - k
      int k = 0;
      for (uint64_t o = 0; o < piece_bytes; o += BOUNCE, k ^= 1) {
        const size_t len = (size_t)std::min<uint64_t>(BOUNCE, piece_bytes - o);
        ELP_HIP(c, hipEventSynchronize(c->bounce_ev[k]));
        memcpy(c->bounce[k], bytes + at_byte + o, len);
        ELP_HIP(c, hipMemcpyAsync(d_raw + o, c->bounce[k], len, hipMemcpyHostToDevice, c->copy_stream));
        ELP_HIP(c, hipEventRecord(c->bounce_ev[k], c->copy_stream));
      }
      ELP_HIP(c, elp::stream_wait(c->copy_stream));
    }
    for (auto &o : off) o += c->raw_bytes;  // offsets into c->raw
    ELP_HIP(c, hipMemcpyAsync(c->raw_off.p + c->n, off.data(), (size_t)(n_rec + 1) * 8, hipMemcpyHostToDevice, st));
    ELP_TRY(stage_bam_columns(c, n_rec, piece_bytes, c->raw_bytes + piece_bytes, max_raw_rec, split_id));
    at_byte = p;
  }
  ELP_HIP(c, elp::stream_wait(st));
  c->adapted = c->sorted = c->marked = false;
  c->have_qual_present = false;
  c->have_snapshot = false;
  c->flat_index_n = 0;
  c->uniform_n = ~0ull;
  return 0;
}

// the chunked size / scan / gather loop of both emit calls; src (device, n_out entries) or nullptr
// bgzf: the stream leaves as BGZF blocks (stored DEFLATE, bgzf.hip), framed on the device pass by pass
static int emit_stream(elp_ctx *c, const BamOut &m, const BamOut &m2, const uint32_t *src, uint64_t n_out, uint64_t max_raw_rec, uint8_t *out, uint64_t cap,
                       uint64_t *n_bytes_out, bool bgzf = false) {
  // records per device pass: sizes and offsets of a pass are scanned in 32 bits, so a pass must stay below 4 GiB of output.  An output
  // record is never longer than the staged one (integer fields only shrink when re-encoded) - except that elp_clean_sam may have added
  // ONE CIGAR operation (4 bytes) - so the largest staged record + 4 bounds it; BGZF framing adds 26 bytes per 65280 (< 0.1 %: the bound
  // leaves 1/64 of headroom).
  const uint32_t CHUNK = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(1u << 21, 0xFC000000ull / (std::max<uint64_t>(max_raw_rec, 64) + 4)));
  uint64_t total = 0;
  hipStream_t st = c->stream;
  for (uint64_t k0 = 0; k0 < n_out; k0 += CHUNK) {
    const uint32_t cnt = (uint32_t)std::min<uint64_t>(CHUNK, n_out - k0);
    uint32_t *sizes;
    ELP_TRY(scratch(c, 4, (size_t)2 * (cnt + 8) + 8, &sizes));
    uint32_t *offs = sizes + cnt + 8, *err = offs + cnt + 8;
    ELP_HIP(c, hipMemsetAsync(err, 0, 4, st));
    ELP_LAUNCH(c, "emit_bam_sizes", k_bam_out_sizes, dim3(blocks_for(cnt, 256)), dim3(256), 0, m, m2, src, k0, cnt, sizes, err);
    uint32_t chunk_bytes = 0;
    ELP_TRY(exclusive_scan_u32(c, sizes, offs, cnt, &chunk_bytes));
    uint32_t he = 0;
    ELP_HIP(c, hipMemcpyAsync(&he, err, 4, hipMemcpyDeviceToHost, st));
    ELP_HIP(c, elp::stream_wait(st));
    if (he & 16u) return set_error(c, ELP_ERR_UNSUPPORTED, "elp_emit_sorted_bam: H-typed optional field");
    if (he) return set_error(c, ELP_ERR_DATA, "elp_emit_sorted_bam: malformed optional fields");
    // (BGZF: the size of the stored form - what the pass needs room for; the compressed members are never larger, their actual size is
    // known behind the device pass.  A size query returns this upper bound)
    uint64_t out_bytes = bgzf ? bgzf_framed_size(chunk_bytes) : (uint64_t)chunk_bytes;
    if (!out) { total += out_bytes; continue; }  // size query
    if (total + out_bytes > cap) return set_error(c, ELP_ERR_ARG, "elp_emit_sorted_bam: output buffer too small (%llu bytes needed so far)", (unsigned long long)(total + out_bytes));
    uint8_t *d_out;
    ELP_TRY(scratch(c, 5, (size_t)chunk_bytes + 64, &d_out));
    const unsigned grid = std::min<unsigned>(blocks_for((uint64_t)cnt * 64, 256), (unsigned)c->n_cu * 32);
    ELP_LAUNCH(c, "emit_bam", k_bam_out_emit, dim3(grid), dim3(256), 0, m, m2, src, k0, cnt, (const uint32_t *)offs, d_out);
    const uint8_t *d_send = d_out;
    if (bgzf) {
      uint8_t *d_framed;
      ELP_TRY(scratch(c, 7, (size_t)out_bytes + 64, &d_framed));
      if (c->tune.bgzf_stored) ELP_TRY(bgzf_frame(c, d_out, chunk_bytes, d_framed));  // stored DEFLATE blocks (tests, measurements)
      else ELP_TRY(bgzf_deflate(c, d_out, chunk_bytes, d_framed, &out_bytes));
      d_send = d_framed;
    }
    ELP_HIP(c, hipMemcpyAsync(out + total, d_send, out_bytes, hipMemcpyDeviceToHost, st));
    ELP_HIP(c, elp::stream_wait(st));
    total += out_bytes;
  }
  *n_bytes_out = total;
  return 0;
}
static BamOut bam_out_of(const elp_ctx *c) {
  return BamOut{c->n - c->n_sr, c->perm.p, c->raw.p, c->raw_off.p, c->refid.p, c->pos.p, c->next_refid.p, c->pnext.p, c->tlen.p, c->flag.p, c->mapq.p, c->l_seq.p,
                c->qname_off.p, c->cigar_off.p, c->qual_off.p, c->qname.p, c->qual.p, c->cigar.p};
}

int elp_emit_sorted_bam(elp_ctx *c, uint8_t *out, uint64_t cap, uint64_t *n_bytes_out) {
  if (!c || !n_bytes_out) return ELP_ERR_ARG;
  ELP_HIP(c, hipSetDevice(c->device));
  if (!c->sorted) return set_error(c, ELP_ERR_ARG, "elp_emit_sorted_bam: call elp_sort_coordinate first");
  ELP_TRY(radix_check(c));
  if (c->raw_n != c->n) return set_error(c, ELP_ERR_ARG, "elp_emit_sorted_bam: records were not staged with elp_stage_bam");
  const BamOut m = bam_out_of(c);
  return emit_stream(c, m, m, nullptr, c->n - c->n_sr, c->max_raw_rec, out, cap, n_bytes_out);
}

// The same records as BGZF blocks (utils/bgzf/bgzf-files.go:324-383): members of at most 65280 input bytes, COMPRESSED on the device since
// round 5 (bgzf.hip: strip-parallel LZ77 + fixed-Huffman DEFLATE; the reference writes compress/flate's default level with dynamic codes -
// its files are ~7 % smaller, the bytes inside are the same; `elp_set_tuning "bgzf_stored"` = 1 keeps round 4's stored blocks), CRC-32 and
// ISIZE per member, framed on the device: what follows a BAM file's header blocks; the host appends the 28-byte end-of-file block (:53-62).
// Inflating the members gives elp_emit_sorted_bam's bytes.
int elp_emit_sorted_bgzf(elp_ctx *c, uint8_t *out, uint64_t cap, uint64_t *n_bytes_out) {
  if (!c || !n_bytes_out) return ELP_ERR_ARG;
  ELP_HIP(c, hipSetDevice(c->device));
  if (!c->sorted) return set_error(c, ELP_ERR_ARG, "elp_emit_sorted_bgzf: call elp_sort_coordinate first");
  ELP_TRY(radix_check(c));
  if (c->raw_n != c->n) return set_error(c, ELP_ERR_ARG, "elp_emit_sorted_bgzf: records were not staged with elp_stage_bam");
  const BamOut m = bam_out_of(c);
  return emit_stream(c, m, m, nullptr, c->n - c->n_sr, c->max_raw_rec, out, cap, n_bytes_out, true);
}

// MergeSortedFilesSplitPerChromosome (sam/split-merge.go:410-576) with payloads: the records of `groups` and of `spread` as ONE stream in
// the merge's order - every spread read behind all group reads of its (refid, POS) - gathered in HBM from both contexts' columns and
// inflated records (elp_merge_spread gives the order alone, for a host that merges bytes itself).
__global__ __launch_bounds__(256) void k_merge_mark(uint64_t ns, const uint64_t *__restrict__ slots, uint32_t *__restrict__ is_spread, uint32_t *__restrict__ src) {
  const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= ns) return;
  is_spread[slots[j]] = 1u;
  src[slots[j]] = MERGE_SECOND | (uint32_t)j;
}
__global__ __launch_bounds__(256) void k_merge_fill(uint64_t n_out, const uint32_t *__restrict__ is_spread, const uint32_t *__restrict__ before, uint32_t *__restrict__ src) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s < n_out && !is_spread[s]) src[s] = (uint32_t)s - before[s];  // group reads fill the remaining slots in order
}

int elp_emit_merged_bam(elp_ctx *groups, elp_ctx *spread, uint8_t *out, uint64_t cap, uint64_t *n_bytes_out) {
  if (!groups || !spread || groups == spread || !n_bytes_out) return ELP_ERR_ARG;
  if (groups->raw_n != groups->n || spread->raw_n != spread->n) return set_error(groups, ELP_ERR_ARG, "elp_emit_merged_bam: records were not staged with elp_stage_bam");
  uint64_t *slots = nullptr;
  ELP_TRY(merge_spread_slots(groups, spread, &slots));  // checks: both sorted, one device
  const uint64_t ng = groups->n - groups->n_sr, ns = spread->n - spread->n_sr, n_out = ng + ns;
  if (n_out >= MERGE_SECOND) return set_error(groups, ELP_ERR_UNSUPPORTED, "elp_emit_merged_bam: more than 2^31 output records");
  if (n_out == 0) { *n_bytes_out = 0; return 0; }
  uint32_t *w;
  ELP_TRY(scratch(groups, 6, 3 * (n_out + 8), &w));
  uint32_t *src = w, *is_spread = w + (n_out + 8), *before = w + 2 * (n_out + 8);
  hipStream_t st = groups->stream;
  ELP_HIP(groups, hipMemsetAsync(is_spread, 0, n_out * 4, st));
  if (ns) ELP_LAUNCH(groups, "merge_mark", k_merge_mark, dim3(blocks_for(ns, 256)), dim3(256), 0, ns, (const uint64_t *)slots, is_spread, src);
  ELP_TRY(exclusive_scan_u32(groups, is_spread, before, n_out, nullptr));
  ELP_LAUNCH(groups, "merge_fill", k_merge_fill, dim3(blocks_for(n_out, 256)), dim3(256), 0, n_out, (const uint32_t *)is_spread, (const uint32_t *)before, src);
  const BamOut mg = bam_out_of(groups), ms = bam_out_of(spread);
  return emit_stream(groups, mg, ms, src, n_out, std::max(groups->max_raw_rec, spread->max_raw_rec), out, cap, n_bytes_out);
}

}  // extern "C"
