// bqsr.hip — base quality score recalibration on the HBM column store: covariate-table gather and LUT apply.
//
// Reference: BaseRecalibrator.Recalibrate (filters/bqsr.go:467-551) with recalibrateAln (:225-244), computeSnpEvents
// (:254-285), computeStrandedClippedSeq (:312-362), contextWith (:87-131), cycle covariates (:364-387), calculateSkipSlice
// (:389-414); the clipping helpers hardClipAdaptorSequence / hardClipSoftClippedBases / hardClip / hardClipCigar /
// cleanHardClippedCigar / getReadCoordinateForReferenceCoordinate (filters/utils.go:148-534); intervals.Intersect
// (intervals/intervals.go:166-173); BaseRecalibratorTables.ApplyBQSR (filters/bqsr.go:936-1005).
//
// Float finalisation (FinalizeBQSRTables, the hierarchical Bayesian estimate) is host work; the device consumes its result
// as a dense byte LUT.
//
// Per-base covariates are local functions of the read:
//   cycle(k)   = cycleFactor + k * increment                                   (bqsr.go:376-387)
//   context(k) = 2-mer key of (previous, current) base in sequencing direction, -1 at the first sequenced base, next to a
//                non-ACGT base, or inside the low-quality tails (quality <= 2 from either end)   (bqsr.go:87-146, 312-362)
#include "common.hpp"

namespace elp {

constexpr int MAX_BQSR_READ = 1024;  // bases; longer reads exceed any sane --max-cycle and make the reference panic anyway

// ------------------------------------------------------------------ CIGAR helpers (BAM-encoded ops)
__device__ __forceinline__ uint32_t c_op(uint32_t c) { return c & 0xF; }
__device__ __forceinline__ int32_t c_len(uint32_t c) { return (int32_t)(c >> 4); }
__device__ __forceinline__ uint32_t c_make(uint32_t op, int32_t len) { return ((uint32_t)len << 4) | op; }

struct RAln {  // working copy of one alignment (`*aln = *alignment`, bqsr.go:479)
  int32_t pos, pnext, tlen, refid, next_refid;
  uint16_t flag;
  const uint32_t *cig;  // current CIGAR
  int ncig;
  int off, len;         // surviving bases [off, off+len) in original read coordinates
  uint32_t *buf[2];     // ping-pong scratch for rewritten CIGARs
  int cur;              // index of the buffer holding `cig`, -1 = original
};

__device__ inline int32_t read_len_of(const uint32_t *c, int n) {
  int32_t l = 0;
  for (int i = 0; i < n; i++) l += op_consumes_read(c_op(c[i])) ? c_len(c[i]) : 0;
  return l;
}
__device__ inline int32_t ref_len_of(const uint32_t *c, int n) {
  int32_t l = 0;
  for (int i = 0; i < n; i++) l += op_consumes_ref(c_op(c[i])) ? c_len(c[i]) : 0;
  return l;
}
__device__ inline int32_t aln_end(const RAln &a) { return a.pos + ref_len_of(a.cig, a.ncig) - 1; }  // sam/sam-types.go:769-775
__device__ inline bool strict_unmapped(const RAln &a) { return (a.flag & F_UNMAPPED) || a.refid < 0 || a.pos == 0; }      // utils.go:141-143
__device__ inline bool strict_next_unmapped(const RAln &a) { return (a.flag & F_NEXT_UNMAPPED) || a.next_refid < 0 || a.pnext == 0; }

// utils.go:224-248
__device__ inline int soft_start(const RAln &a) {
  int32_t s = a.pos;
  for (int i = 0; i < a.ncig; i++) {
    const uint32_t op = c_op(a.cig[i]);
    if (op == OP_S) s -= c_len(a.cig[i]);
    else if (op != OP_H) break;
  }
  return s;
}
__device__ inline int soft_end(const RAln &a) {
  const int32_t end = aln_end(a);
  int32_t se = end;
  for (int i = a.ncig - 1; i >= 0; i--) {
    const uint32_t op = c_op(a.cig[i]);
    if (op == OP_S) se += c_len(a.cig[i]);
    else if (op != OP_H) return se;
  }
  return end;
}

// utils.go:267-326; returns read coordinate or -1, *falls = fallsInsideOrJustBeforeDeletionOrSkippedRegion
__device__ inline int compute_read_coord(const uint32_t *c, int n, int softstart, int ref_index, bool *falls) {
  const int goal = ref_index - softstart;
  *falls = false;
  if (goal < 0) return -1;
  int read_bases = 0, ref_bases = 0;
  bool falls_inside = false, ends_before = false, fob = false;
  int index = 0;
  while (ref_bases != goal && index < n) {
    const uint32_t el = c[index++];
    const uint32_t op = c_op(el);
    const int el_len = c_len(el);
    int shift = 0;
    if (op_consumes_ref(op) || op == OP_S) {
      shift = (ref_bases + el_len < goal) ? el_len : goal - ref_bases;
      ref_bases += shift;
    }
    const int cr = op_consumes_read(op) ? 1 : 0;
    if (ref_bases != goal) {
      read_bases += cr * el_len;
    } else {
      if (shift >= el_len && index == n) return -1;
      uint32_t next_op = 0xF;
      if (shift < el_len) {
        falls_inside = op == OP_D || op == OP_N;
      } else {
        uint32_t nx = c[index++];
        if (c_op(nx) == OP_I) {
          read_bases += c_len(nx);
          if (index == n) return -1;
          nx = c[index++];
        }
        next_op = c_op(nx);
        ends_before = next_op == OP_D || next_op == OP_N;
      }
      fob = ends_before || falls_inside;
      if (!fob) read_bases += cr * shift;
      else if (ends_before) read_bases += cr * (shift - 1);
      else if (falls_inside) read_bases--;
    }
  }
  if (ref_bases != goal) return -1;
  *falls = fob;
  return read_bases;
}

// utils.go:335-349 (+ readStartsWithInsertion bqsr.go:287-299)
__device__ inline int get_read_coord(const uint32_t *c, int n, int softstart, int ref_index, bool right_tail, bool *ok) {
  bool falls;
  int rb = compute_read_coord(c, n, softstart, ref_index, &falls);
  if (rb == -1) { *ok = false; return -1; }
  if (right_tail && falls) rb++;
  if (!right_tail && rb == 0) {
    for (int i = 0; i < n; i++) {
      const uint32_t op = c_op(c[i]);
      if (op == OP_I) {
        const int32_t m = read_len_of(c, n) - 1;
        rb = c_len(c[i]) < m ? c_len(c[i]) : m;
        break;
      }
      if (op == OP_H || op == OP_S) continue;
      break;
    }
  }
  *ok = true;
  return rb;
}

// utils.go:351-372
__device__ inline int32_t hard_soft_offset(const uint32_t *c, int n) {
  int32_t size = 0;
  int i = 0;
  for (; i < n && c_op(c[i]) == OP_H; i++) size += c_len(c[i]);
  for (; i < n && c_op(c[i]) == OP_S; i++) size += c_len(c[i]);
  return size;
}
// utils.go:378-386
__device__ inline int clip_shift(uint32_t el, int cigar_length) {
  const uint32_t op = c_op(el);
  if (op == OP_I) return -cigar_length;
  if (op == OP_D || op == OP_N) return c_len(el);
  return 0;
}

// utils.go:488-517, in place
__device__ inline int clean_hard_clipped(uint32_t *c, int n) {
  int total = 0, index = 0;
  for (; index < n; index++) {
    const uint32_t op = c_op(c[index]);
    if (op == OP_H || op == OP_D || op == OP_N) total += c_len(c[index]);
    else break;
  }
  if (index > 0) {
    c[0] = c_make(OP_H, total);
    for (int k = index; k < n; k++) c[1 + k - index] = c[k];
    n = 1 + (n - index);
  }
  total = 0;
  index = n - 1;
  for (; index >= 0; index--) {
    const uint32_t op = c_op(c[index]);
    if (op == OP_H || op == OP_D || op == OP_N) total += c_len(c[index]);
    else break;
  }
  if (index < n - 1) {
    n = index + 1;
    c[n++] = c_make(OP_H, total);
  }
  return n;
}

// utils.go:406-486; writes the new CIGAR to `out` (capacity ncig + 4) and returns its length
__device__ inline int hard_clip_cigar(const RAln &a, int start, int stop, uint32_t *out) {
  const uint32_t *cv = a.cig;
  const int n = a.ncig;
  int index = 0, total_hard = stop - start + 1, shift_acc = 0, no = 0;
  if (start == 0) {
    int ci = 0;
    for (int k = 0; k < n; k++) {  // Go: for cigarOpIndex, cigarOp = range cigarVec
      ci = k;
      if (c_op(cv[k]) != OP_H) break;
      total_hard += c_len(cv[k]);
    }
    for (; index <= stop && ci < n; ci++) {
      const uint32_t el = cv[ci];
      const int el_len = c_len(el);
      const int shift = op_consumes_read(c_op(el)) ? el_len : 0;
      if (index + shift == stop + 1) {
        shift_acc += clip_shift(el, el_len);
        out[no++] = c_make(OP_H, total_hard + shift_acc);
      } else if (index + shift > stop + 1) {
        const int after = el_len - (stop - index + 1);
        shift_acc += clip_shift(el, stop - index + 1);
        out[no++] = c_make(OP_H, total_hard + shift_acc);
        out[no++] = c_make(c_op(el), after);
      }
      index += shift;
      shift_acc += clip_shift(el, shift);
    }
    for (; ci < n; ci++) out[no++] = cv[ci];
  } else {
    int ci = 0;
    for (; index < start && ci < n; ci++) {
      const uint32_t el = cv[ci];
      const int el_len = c_len(el);
      const int shift = op_consumes_read(c_op(el)) ? el_len : 0;
      if (index + shift < start) {
        out[no++] = el;
      } else {
        const int after = start - index;
        shift_acc += clip_shift(el, el_len - (start - index));
        if (c_op(el) == OP_H) total_hard += after;
        else out[no++] = c_make(c_op(el), after);
      }
      index += shift;
    }
    for (; ci < n; ci++) {
      const uint32_t el = cv[ci];
      shift_acc += clip_shift(el, c_len(el));
      if (c_op(el) == OP_H) total_hard += c_len(el);
    }
    out[no++] = c_make(OP_H, total_hard + shift_acc);
  }
  return clean_hard_clipped(out, no);
}

// utils.go:388-404
__device__ inline void hard_clip(RAln &a, int start, int stop) {
  const int nb = a.cur == 0 ? 1 : 0;
  uint32_t *out = a.buf[nb];
  const int no = hard_clip_cigar(a, start, stop, out);
  const int new_len = a.len - (stop - start + 1);
  const int copy_start = start == 0 ? stop + 1 : 0;
  const int32_t old_off = hard_soft_offset(a.cig, a.ncig);
  a.cig = out; a.ncig = no; a.cur = nb;
  a.off += copy_start;
  a.len = new_len;
  if (start == 0 && !strict_unmapped(a)) a.pos += hard_soft_offset(a.cig, a.ncig) - old_off;
}

// utils.go:149-180, 214-222; returns false where the reference panics
__device__ inline bool hard_clip_adaptor(RAln &a) {
  const bool rev = a.flag & F_REVERSED;
  if (!(a.tlen != 0 && (a.flag & F_MULTIPLE) && !strict_unmapped(a) && !strict_next_unmapped(a) && rev != (bool)(a.flag & F_NEXT_REVERSED)))
    return true;
  int end_v;
  bool well;
  if (rev) { const int32_t e = aln_end(a); well = e > a.pnext; end_v = e; }
  else { well = a.pos <= a.pnext + a.tlen; end_v = -1; }
  if (!well) return true;
  const int boundary = rev ? (int)a.pnext - 1 : (int)a.pos + (a.tlen < 0 ? -(int)a.tlen : (int)a.tlen);
  if (boundary < (int)a.pos) return true;  // isInsideRead
  if (end_v < 0) end_v = aln_end(a);
  if (boundary > end_v) return true;
  bool ok;
  if (rev) {
    const int stop = get_read_coord(a.cig, a.ncig, soft_start(a), boundary, false, &ok);
    if (!ok) return false;
    hard_clip(a, 0, stop);
  } else {
    const int start = get_read_coord(a.cig, a.ncig, soft_start(a), boundary, true, &ok);
    if (!ok) return false;
    hard_clip(a, start, a.len - 1);
  }
  return true;
}

// utils.go:519-548
__device__ inline void hard_clip_soft_clipped(RAln &a) {
  int read_index = 0, cut_left = -1, cut_right = -1;
  bool right_tail = false;
  for (int i = 0; i < a.ncig; i++) {
    const uint32_t op = c_op(a.cig[i]);
    const int ln = c_len(a.cig[i]);
    if (op == OP_S) {
      if (right_tail) cut_right = read_index;
      else cut_left = read_index + ln - 1;
    } else if (op != OP_H) {
      right_tail = true;
    }
    read_index += op_consumes_read(op) ? ln : 0;
  }
  if (cut_right >= 0) hard_clip(a, cut_right, a.len - 1);
  if (cut_left >= 0) hard_clip(a, 0, cut_left);
}

// ------------------------------------------------------------------ bases
__device__ __forceinline__ uint32_t nibble_at(const uint8_t *__restrict__ s4, int k) {
  const uint32_t b = s4[k >> 1];
  return (k & 1) ? (b & 0xF) : (b >> 4);
}
// simpleBaseToBaseIndex on Sequence.Base(): A0 C1 G2 T3, everything else -1 (bqsr.go:55-62; '=' is not '*')
__device__ __forceinline__ int base_index_of_nibble(uint32_t nb) { return nb == 1 ? 0 : nb == 2 ? 1 : nb == 4 ? 2 : nb == 8 ? 3 : -1; }
// baseToIntMap on a raw reference byte (bqsr.go:247-252)
__device__ __forceinline__ int base_code_of_ref(uint8_t c) {
  switch (c) {
    case 'a': case 'A': case '*': return 1;
    case 'c': case 'C': return 2;
    case 'g': case 'G': return 3;
    case 't': case 'T': return 4;
    default: return 0;
  }
}
// baseToIntMap on Sequence.Base(): "=ACMGRSVTWYHKDBN" -> A1 C2 G3 T4 else 0
__device__ __forceinline__ int base_code_of_nibble(uint32_t nb) { return nb == 1 ? 1 : nb == 2 ? 2 : nb == 4 ? 3 : nb == 8 ? 4 : 0; }

struct ReadView {
  const uint8_t *seq4;  // original packed bases of the record
  const uint8_t *qual;  // original quals of the record
  int off, len;         // current window
  bool reversed;
  int left, right;      // low-quality-tail mask bounds inside the window (left > right: whole read masked)
};

// computeStrandedClippedSeq mask bounds, bqsr.go:316-332
__device__ inline void low_quality_bounds(ReadView &v) {
  int left = v.len;
  for (int i = 0; i < v.len; i++) if (v.qual[v.off + i] > 2) { left = i; break; }
  int right = left - 1;
  for (int i = v.len - 1; i >= left; i--) if (v.qual[v.off + i] > 2) { right = i; break; }
  v.left = left; v.right = right;
}
__device__ __forceinline__ int masked_index(const ReadView &v, int k) {  // base index or -1 (masked / non-ACGT / outside)
  if (k < v.left || k > v.right) return -1;
  return base_index_of_nibble(nibble_at(v.seq4, v.off + k));
}
// context covariate of base k, bqsr.go:87-146
__device__ __forceinline__ int context_key(const ReadView &v, int k) {
  if (!v.reversed) {
    if (k < 1) return -1;
    const int p = masked_index(v, k - 1), q = masked_index(v, k);
    if (p < 0 || q < 0) return -1;
    return 2 | (p << 4) | (q << 6);
  }
  if (k > v.len - 2) return -1;
  const int p = masked_index(v, k + 1), q = masked_index(v, k);
  if (p < 0 || q < 0) return -1;
  return 2 | ((3 - p) << 4) | ((3 - q) << 6);  // complement: A<->T, C<->G
}
__device__ __forceinline__ void cycle_params(uint16_t flag, int len, int *factor, int *incr) {  // bqsr.go:376-383
  const int reversed = (flag & F_REVERSED) >> 4, last = (flag & F_LAST) >> 7;
  const int rof = 1 - 2 * last;
  *factor = rof + reversed * (len - 1) * rof;
  *incr = (1 - 2 * reversed) * rof;
}

struct BqCols {
  uint64_t n;
  const int32_t *refid, *pos, *next_refid, *pnext, *tlen;
  const uint16_t *flag, *rgid;
  const uint8_t *mapq, *has_sr;
  const uint32_t *l_seq;
  const uint64_t *cigar_off, *seq_off, *qual_off;
  const uint32_t *cigar;
  const uint8_t *seq4;
  const uint8_t *qual;
  const int32_t *ref_len;
  const uint16_t *rg_cov;
  int32_t n_ref;
  uint8_t *const *ref_seq;
  const int64_t *ref_seq_len;
  int32_t *const *sites;
  const int64_t *n_sites;
};

// recalibrateAln, bqsr.go:225-244 (+ utils.go:121-139)
__device__ inline bool recalibrate_aln(const BqCols &m, uint64_t i) {
  if (m.has_sr[i]) return false;
  const uint8_t mq = m.mapq[i];
  if (!(mq > 0 && mq < 255)) return false;
  const uint16_t f = m.flag[i];
  if (f & (F_SECONDARY | F_DUPLICATE | F_QCFAILED)) return false;
  const int32_t r = m.refid[i], p = m.pos[i];
  if ((f & F_UNMAPPED) || r < 0 || p == 0) return false;
  if (!(p > 0)) return false;
  const uint32_t ls = m.l_seq[i];
  if (ls == 0) return false;
  if ((uint64_t)ls != m.qual_off[i + 1] - m.qual_off[i]) return false;
  if (m.rgid[i] == ELP_NIL16) return false;
  if (!(r < m.n_ref && p <= m.ref_len[r])) return false;
  int32_t rl = 0, refl = 0;
  for (uint64_t k = m.cigar_off[i]; k < m.cigar_off[i + 1]; k++) {
    const uint32_t c = m.cigar[k];
    if (c_op(c) == OP_N) return false;
    if (op_consumes_read(c_op(c))) rl += c_len(c);
    if (op_consumes_ref(c_op(c))) refl += c_len(c);
  }
  return refl >= 0 && (int32_t)ls == rl;
}

// Recalibrate per-read body, bqsr.go:478-539.  One thread per record; tables updated with 64-bit atomics.
__global__ __launch_bounds__(256) void k_bqsr_gather(BqCols m, int max_cycle, uint32_t *__restrict__ cig_scratch, unsigned long long *qual_tbl,
                                                     unsigned long long *cycle_tbl, unsigned long long *ctx_tbl, uint32_t *err) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.n) return;
  if (!recalibrate_aln(m, i)) return;
  if (m.l_seq[i] > (uint32_t)MAX_BQSR_READ) { atomicOr(&err[0], 2u); return; }
  RAln a;
  a.pos = m.pos[i]; a.pnext = m.pnext[i]; a.tlen = m.tlen[i]; a.refid = m.refid[i]; a.next_refid = m.next_refid[i];
  a.flag = m.flag[i];
  a.cig = m.cigar + m.cigar_off[i];
  a.ncig = (int)(m.cigar_off[i + 1] - m.cigar_off[i]);
  a.off = 0; a.len = (int)m.l_seq[i];
  uint32_t *sc = cig_scratch + 2 * (m.cigar_off[i] + 4 * i);
  a.buf[0] = sc; a.buf[1] = sc + (a.ncig + 4);
  a.cur = -1;
  if (!hard_clip_adaptor(a)) { atomicOr(&err[0], 4u); return; }
  if (a.len == 0) return;
  hard_clip_soft_clipped(a);
  if (a.len == 0) return;

  const uint8_t *seq4 = m.seq4 + m.seq_off[i];
  const uint8_t *qual = m.qual + m.qual_off[i];
  uint32_t skip[MAX_BQSR_READ / 32], snp[MAX_BQSR_READ / 32];
  const int nw = (a.len + 31) >> 5;
  for (int w = 0; w < nw; w++) { skip[w] = 0; snp[w] = 0; }

  // calculateSkipSlice :389-414
  {
    const int ss = soft_start(a), se = soft_end(a);
    const int32_t *sv = m.sites[a.refid];
    const int64_t ns = m.n_sites[a.refid];
    int64_t lo = 0, hi = ns;
    while (lo < hi) { const int64_t md = lo + (hi - lo) / 2; if (!(sv[2 * md + 1] >= ss)) lo = md + 1; else hi = md; }
    int64_t first = lo;
    lo = 0; hi = ns;
    while (lo < hi) { const int64_t md = lo + (hi - lo) / 2; if (!(sv[2 * md] > se)) lo = md + 1; else hi = md; }
    const int64_t last = lo;
    for (int64_t s = first; s < last; s++) {
      bool ok;
      int fs = get_read_coord(a.cig, a.ncig, ss, sv[2 * s], false, &ok);
      if (!ok || fs < 0) fs = 0;
      int fe = get_read_coord(a.cig, a.ncig, ss, sv[2 * s + 1], false, &ok);
      if (!ok || fe > a.len - 1) fe = a.len - 1;
      for (int k = fs; k <= fe; k++) skip[k >> 5] |= 1u << (k & 31);
    }
  }
  // computeSnpEvents :254-285 (reference bytes past the contig end: the Go code panics; read as 'N' here)
  {
    const uint8_t *ref = m.ref_seq[a.refid];
    const int64_t rlen = m.ref_seq_len[a.refid];
    int ri = 0;
    int64_t j = (int64_t)a.pos - 1;
    for (int c = 0; c < a.ncig; c++) {
      const uint32_t op = c_op(a.cig[c]);
      const int ln = c_len(a.cig[c]);
      if (op == OP_M || op == OP_EQ || op == OP_X) {
        for (int k = 0; k < ln; k++, ri++, j++) {
          if (ri >= a.len) continue;
          const int rb = (j >= 0 && j < rlen) ? base_code_of_ref(ref[j]) : 0;
          if (base_code_of_nibble(nibble_at(seq4, a.off + ri)) != rb) snp[ri >> 5] |= 1u << (ri & 31);
        }
      } else if (op == OP_D || op == OP_N) {
        j += ln;
      } else if (op == OP_I || op == OP_S) {
        ri += ln;
      }
    }
  }
  const uint32_t cov = m.rg_cov[m.rgid[i]];
  int cf, ci;
  cycle_params(a.flag, a.len, &cf, &ci);
  ReadView v{seq4, qual, a.off, a.len, (bool)(a.flag & F_REVERSED), 0, -1};
  low_quality_bounds(v);
  const int ncyc = 2 * max_cycle + 1;
  for (int k = 0; k < a.len; k++) {
    if (skip[k >> 5] & (1u << (k & 31))) continue;
    if (base_index_of_nibble(nibble_at(seq4, a.off + k)) < 0) continue;
    const uint32_t q = qual[a.off + k];
    if (q < 6) continue;
    if (q >= ELP_NQUAL) { atomicOr(&err[0], 8u); return; }
    const unsigned long long e = (snp[k >> 5] >> (k & 31)) & 1u;
    const size_t qi = (size_t)cov * ELP_NQUAL + q;
    atomicAdd(&qual_tbl[2 * qi], 1ull);
    if (e) atomicAdd(&qual_tbl[2 * qi + 1], 1ull);
    const int cyc = cf + k * ci;
    if (cyc > max_cycle || cyc < -max_cycle) { atomicOr(&err[0], 16u); return; }  // checkCycleCovariate :364-369
    const size_t cyi = qi * ncyc + (size_t)(cyc + max_cycle);
    atomicAdd(&cycle_tbl[2 * cyi], 1ull);
    if (e) atomicAdd(&cycle_tbl[2 * cyi + 1], 1ull);
    const int cx = context_key(v, k);
    if (cx >= 0) {
      const size_t xi = qi * ELP_NCTX + (size_t)((cx >> 4) & 15);
      atomicAdd(&ctx_tbl[2 * xi], 1ull);
      if (e) atomicAdd(&ctx_tbl[2 * xi + 1], 1ull);
    }
  }
}

// ApplyBQSR per-read body, bqsr.go:947-1003
__global__ __launch_bounds__(256) void k_bqsr_apply(uint64_t n, const uint16_t *__restrict__ flag, const uint16_t *__restrict__ rgid,
                                                    const uint16_t *__restrict__ rg_cov, const uint32_t *__restrict__ l_seq,
                                                    const uint64_t *__restrict__ seq_off, const uint8_t *__restrict__ seq4,
                                                    const uint64_t *__restrict__ qual_off, uint8_t *__restrict__ qual, int max_cycle,
                                                    const uint8_t *__restrict__ lut, const uint8_t *__restrict__ cov_present, uint32_t *err) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint16_t rg = rgid[i];
  if (rg == ELP_NIL16) { atomicOr(&err[0], 32u); return; }  // readGroupCovariate panics, bqsr.go:38
  const uint32_t cov = rg_cov[rg];
  if (!cov_present[cov]) return;  // :953-955
  const int len = (int)l_seq[i];
  if ((uint64_t)len != qual_off[i + 1] - qual_off[i]) { atomicOr(&err[0], 64u); return; }
  uint8_t *q = qual + qual_off[i];
  const uint16_t f = flag[i];
  ReadView v{seq4 + seq_off[i], q, 0, len, (bool)(f & F_REVERSED), 0, -1};
  low_quality_bounds(v);
  int cf, ci;
  cycle_params(f, len, &cf, &ci);
  const int ncyc = 2 * max_cycle + 1;
  for (int k = 0; k < len; k++) {
    const uint32_t qq = q[k];
    if (qq < 6) continue;
    if (qq >= ELP_NQUAL) { atomicOr(&err[0], 8u); return; }
    const int cyc = cf + k * ci;
    if (cyc > max_cycle || cyc < -max_cycle) { atomicOr(&err[0], 16u); return; }
    const int cx = context_key(v, k);
    const size_t li = (((size_t)cov * ELP_NQUAL + qq) * ncyc + (size_t)(cyc + max_cycle)) * 17 + (size_t)(cx < 0 ? 16 : ((cx >> 4) & 15));
    q[k] = lut[li];
  }
}

static int bqsr_error(elp_ctx *c, uint32_t e) {
  ELP_HIP(c, hipMemsetAsync(c->err_flag.p, 0, 4, c->stream));
  if (e & 2u) return set_error(c, ELP_ERR_UNSUPPORTED, "BQSR: read longer than %d bases", MAX_BQSR_READ);
  if (e & 4u) return set_error(c, ELP_ERR_DATA, "reference coordinate matches a non-existing base in read (reference: log.Panicf, filters/utils.go:253,262)");
  if (e & 8u) return set_error(c, ELP_ERR_DATA, "BQSR: base quality above 93");
  if (e & 16u) return set_error(c, ELP_ERR_DATA, "cycle value exceeds maximum cycle value (reference: log.Panic, filters/bqsr.go:364-369)");
  if (e & 32u) return set_error(c, ELP_ERR_DATA, "BQSR requires input with read groups (reference: log.Panic, filters/bqsr.go:38)");
  if (e & 64u) return set_error(c, ELP_ERR_DATA, "ApplyBQSR: len(QUAL) != len(SEQ) (reference: index out of range panic)");
  return set_error(c, ELP_ERR_DATA, "BQSR: device error word %u", e);
}

static int sync_bqsr_ptrs(elp_ctx *c) {
  if (!c->bqsr_ptrs_dirty) return 0;
  const size_t nr = (size_t)c->n_ref;
  ELP_TRY(ensure(c, c->d_ref_seq, nr + 1));
  ELP_TRY(ensure(c, c->d_ref_seq_len, nr + 1));
  ELP_TRY(ensure(c, c->d_sites, nr + 1));
  ELP_TRY(ensure(c, c->d_n_sites, nr + 1));
  if (nr) {
    ELP_HIP(c, hipMemcpyAsync(c->d_ref_seq.p, c->h_ref_seq.data(), nr * sizeof(uint8_t *), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipMemcpyAsync(c->d_ref_seq_len.p, c->h_ref_seq_len.data(), nr * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipMemcpyAsync(c->d_sites.p, c->h_sites.data(), nr * sizeof(int32_t *), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipMemcpyAsync(c->d_n_sites.p, c->h_n_sites.data(), nr * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipStreamSynchronize(c->stream));
  }
  c->bqsr_ptrs_dirty = false;
  return 0;
}

}  // namespace elp

using namespace elp;

extern "C" {

int elp_bqsr_set_reference(elp_ctx *c, int32_t refid, const uint8_t *bases, int64_t len) {
  if (!c || !c->have_header || refid < 0 || refid >= c->n_ref || len < 0 || (len && !bases)) return set_error(c, ELP_ERR_ARG, "elp_bqsr_set_reference: bad arguments");
  ELP_HIP(c, hipSetDevice(c->device));
  if (c->h_ref_seq[refid]) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->h_ref_seq[refid]); c->h_ref_seq[refid] = nullptr; }
  uint8_t *d = nullptr;
  ELP_HIP(c, hipMalloc((void **)&d, (size_t)len + 16));
  if (len) ELP_HIP(c, hipMemcpyAsync(d, bases, (size_t)len, hipMemcpyHostToDevice, c->stream));
  ELP_HIP(c, hipStreamSynchronize(c->stream));
  c->h_ref_seq[refid] = d;
  c->h_ref_seq_len[refid] = len;
  c->bqsr_ptrs_dirty = true;
  return 0;
}

int elp_bqsr_set_known_sites(elp_ctx *c, int32_t refid, const int32_t *start_end, int64_t n) {
  if (!c || !c->have_header || refid < 0 || refid >= c->n_ref || n < 0 || (n && !start_end)) return set_error(c, ELP_ERR_ARG, "elp_bqsr_set_known_sites: bad arguments");
  for (int64_t k = 1; k < n; k++)
    if (!(start_end[2 * k] > start_end[2 * k - 1])) return set_error(c, ELP_ERR_ARG, "known sites of refid %d are not sorted and flattened at index %lld", refid, (long long)k);
  ELP_HIP(c, hipSetDevice(c->device));
  if (c->h_sites[refid]) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->h_sites[refid]); c->h_sites[refid] = nullptr; }
  int32_t *d = nullptr;
  ELP_HIP(c, hipMalloc((void **)&d, (size_t)(2 * n + 4) * sizeof(int32_t)));
  if (n) ELP_HIP(c, hipMemcpyAsync(d, start_end, (size_t)(2 * n) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  ELP_HIP(c, hipStreamSynchronize(c->stream));
  c->h_sites[refid] = d;
  c->h_n_sites[refid] = n;
  c->bqsr_ptrs_dirty = true;
  return 0;
}

int elp_bqsr_gather(elp_ctx *c, int max_cycle, int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl) {
  if (!c || !qual_tbl || !cycle_tbl || !ctx_tbl || max_cycle < 1) return set_error(c, ELP_ERR_ARG, "elp_bqsr_gather: bad arguments");
  ELP_HIP(c, hipSetDevice(c->device));
  for (int r = 0; r < c->n_ref; r++)
    if (!c->h_ref_seq[r]) return set_error(c, ELP_ERR_ARG, "elp_bqsr_gather: no reference sequence set for refid %d", r);
  for (int r = 0; r < c->n_ref; r++)
    if (!c->h_sites[r]) ELP_TRY(elp_bqsr_set_known_sites(c, r, nullptr, 0));
  ELP_TRY(sync_bqsr_ptrs(c));
  const int ncyc = 2 * max_cycle + 1;
  const size_t nq = (size_t)c->n_cov * ELP_NQUAL * 2, nc = nq * ncyc, nx = nq * ELP_NCTX;
  unsigned long long *tb;
  ELP_TRY(scratch(c, 0, nq + nc + nx + 8, &tb));
  ELP_HIP(c, hipMemsetAsync(tb, 0, (nq + nc + nx) * sizeof(unsigned long long), c->stream));
  const uint64_t n = c->n;
  if (n) {
    uint32_t *cs;
    ELP_TRY(scratch(c, 1, 2 * (c->cigar_ops + 4 * n) + 64, &cs));
    BqCols m{n, c->refid.p, c->pos.p, c->next_refid.p, c->pnext.p, c->tlen.p, c->flag.p, c->rgid.p, c->mapq.p, c->has_sr.p, c->l_seq.p,
             c->cigar_off.p, c->seq_off.p, c->qual_off.p, c->cigar.p, c->seq4.p, c->qual.p, c->ref_len.p, c->rg_cov.p, c->n_ref,
             c->d_ref_seq.p, c->d_ref_seq_len.p, c->d_sites.p, c->d_n_sites.p};
    ELP_LAUNCH(c, "bqsr_gather", k_bqsr_gather, dim3(blocks_for(n, 256)), dim3(256), 0, m, max_cycle, cs, tb, tb + nq, tb + nq + nc, c->err_flag.p);
  }
  ELP_HIP(c, hipMemcpyAsync(qual_tbl, tb, nq * 8, hipMemcpyDeviceToHost, c->stream));
  ELP_HIP(c, hipMemcpyAsync(cycle_tbl, tb + nq, nc * 8, hipMemcpyDeviceToHost, c->stream));
  ELP_HIP(c, hipMemcpyAsync(ctx_tbl, tb + nq + nc, nx * 8, hipMemcpyDeviceToHost, c->stream));
  uint32_t e[4];
  ELP_TRY(fetch_err(c, e));
  if (e[0]) return bqsr_error(c, e[0]);
  return 0;
}

int elp_bqsr_apply(elp_ctx *c, int max_cycle, const uint8_t *lut, const uint8_t *cov_present) {
  if (!c || !lut || !cov_present || max_cycle < 1) return set_error(c, ELP_ERR_ARG, "elp_bqsr_apply: bad arguments");
  ELP_HIP(c, hipSetDevice(c->device));
  const size_t ncyc = 2 * (size_t)max_cycle + 1;
  const size_t lut_bytes = (size_t)c->n_cov * ELP_NQUAL * ncyc * 17;
  uint8_t *dl;
  ELP_TRY(scratch(c, 0, lut_bytes + (size_t)c->n_cov + 64, &dl));
  ELP_HIP(c, hipMemcpyAsync(dl, lut, lut_bytes, hipMemcpyHostToDevice, c->stream));
  ELP_HIP(c, hipMemcpyAsync(dl + lut_bytes, cov_present, (size_t)c->n_cov, hipMemcpyHostToDevice, c->stream));
  const uint64_t n = c->n;
  if (n) {
    ELP_LAUNCH(c, "bqsr_apply", k_bqsr_apply, dim3(blocks_for(n, 256)), dim3(256), 0, n, (const uint16_t *)c->flag.p, (const uint16_t *)c->rgid.p,
               (const uint16_t *)c->rg_cov.p, (const uint32_t *)c->l_seq.p, (const uint64_t *)c->seq_off.p, (const uint8_t *)c->seq4.p,
               (const uint64_t *)c->qual_off.p, c->qual.p, max_cycle, (const uint8_t *)dl, (const uint8_t *)(dl + lut_bytes), c->err_flag.p);
  }
  uint32_t e[4];
  ELP_TRY(fetch_err(c, e));
  if (e[0]) return bqsr_error(c, e[0]);
  c->adapted = false;  // scores depend on QUAL
  return 0;
}

}  // extern "C"
