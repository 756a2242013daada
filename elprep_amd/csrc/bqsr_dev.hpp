// bqsr_dev.hpp — device helpers of the BQSR kernels: CIGAR clipping, read-coordinate mapping, per-base covariates.
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
// Per-base covariates are local functions of the read (evaluated block-wise in bqsr.hip / flat.hpp):
//   cycle(k)   = cycleFactor + k * increment                                   (bqsr.go:376-387)
//   context(k) = 2-mer key of (previous, current) base in sequencing direction, -1 at the first sequenced base, next to a
//                non-ACGT base, or inside the low-quality tails (quality <= 2 from either end)   (bqsr.go:87-146, 312-362)
#pragma once
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

// ------------------------------------------------------------------ low-quality tails
struct ReadView {
  const uint8_t *seq4;  // unused by the bounds (kept for symmetry with the record's columns)
  const uint8_t *qual;  // original quals of the record
  int off, len;         // current window
  bool reversed;
  int left, right;      // low-quality-tail mask bounds inside the window (left > right: whole read masked)
};

// computeStrandedClippedSeq mask bounds, bqsr.go:316-332 (the general prologue needs them inside the clipped window; for
// unclipped reads they come from adapt_score)
__device__ inline void low_quality_bounds(ReadView &v) {
  int left = v.len;
  for (int i = 0; i < v.len; i++) if (v.qual[v.off + i] > 2) { left = i; break; }
  int right = left - 1;
  for (int i = v.len - 1; i >= left; i--) if (v.qual[v.off + i] > 2) { right = i; break; }
  v.left = left; v.right = right;
}

}  // namespace elp
