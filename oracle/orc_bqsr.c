/*
 * oracle/orc_bqsr.c — CPU oracle (test infrastructure; see orc.h): base quality score recalibration.
 *
 * Restates filters/bqsr.go (entire), the clipping helpers of filters/utils.go:121-534 and
 * intervals/intervals.go:88-173 as a sequential plain-C program on an AoS working copy of each record
 * (the reference also works on a copy: `*aln = *alignment`, bqsr.go:479).
 *
 * Count tables are dense arrays instead of Go maps; an entry "exists" iff Observations > 0 (a map entry is
 * created by its first update, bqsr.go:195-203).
 *
 * Float path: Go's math.Log10 / math.Pow are restated structurally (math/log10.go, math/pow.go) on top of Go's own
 * math.Log and math.Lgamma as orc_gomath.c restates them (round 4; rounds 1-3 leaned on glibc's log / lgamma_r);
 * math.Exp stays libm's (see orc_gomath.c).  These only feed an argmax over 61 bins and a %.4f print.
 */
#include "orc.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

static const char cigar_ops[] = "MIDNSHP=X";
static const char nibble_to_base[] = "=ACMGRSVTWYHKDBN"; /* sam/sam-types.go:228 */

typedef struct { int32_t len; char op; } cop;

/* working copy of one alignment */
typedef struct {
  int32_t pos, pnext, tlen, refid, next_refid;
  uint16_t flag; uint8_t mapq;
  cop *cigar; int n_cigar, cap_cigar;
  const uint8_t *seq4; /* original packed bases */
  const uint8_t *qual; /* original quals */
  int off, len;        /* surviving base range [off, off+len) in original read coordinates */
} waln;

static int consumes_read(char op) { return op == 'M' || op == 'I' || op == 'S' || op == '=' || op == 'X'; }
static int consumes_ref(char op) { return op == 'M' || op == 'D' || op == 'N' || op == '=' || op == 'X'; }

static char wbase(const waln *a, int i) { /* Sequence.Base, sam/sam-types.go:265 */
  int k = a->off + i;
  uint8_t byte = a->seq4[k >> 1];
  return nibble_to_base[(k & 1) ? (byte & 0xF) : (byte >> 4)];
}
static uint8_t wqual(const waln *a, int i) { return a->qual[a->off + i]; }

static void cigar_reserve(waln *a, int n) {
  if (n > a->cap_cigar) {
    a->cap_cigar = n * 2 + 8;
    a->cigar = (cop *)realloc(a->cigar, (size_t)a->cap_cigar * sizeof(cop));
  }
}

static void load_aln(const orc_batch *b, uint64_t i, const uint16_t *flags, waln *a) {
  a->pos = b->pos[i]; a->pnext = b->pnext[i]; a->tlen = b->tlen[i];
  a->refid = b->refid[i]; a->next_refid = b->next_refid[i];
  a->flag = flags ? flags[i] : b->flag[i]; a->mapq = b->mapq[i];
  int n = (int)(b->cigar_off[i + 1] - b->cigar_off[i]);
  cigar_reserve(a, n + 4);
  a->n_cigar = n;
  for (int k = 0; k < n; k++) {
    uint32_t c = b->cigar[b->cigar_off[i] + k];
    a->cigar[k].len = (int32_t)(c >> 4);
    a->cigar[k].op = (c & 0xF) < 9 ? cigar_ops[c & 0xF] : '?';
  }
  a->seq4 = b->seq4 + b->seq_off[i];
  a->qual = b->qual + b->qual_off[i];
  a->off = 0; a->len = (int)b->l_seq[i];
}

/* sam/sam-types.go:747-775 */
static int32_t read_length_from_cigar(const cop *c, int n) {
  int32_t l = 0;
  for (int i = 0; i < n; i++) l += consumes_read(c[i].op) * c[i].len;
  return l;
}
static int32_t reference_length_from_cigar(const cop *c, int n) {
  int32_t l = 0;
  for (int i = 0; i < n; i++) l += consumes_ref(c[i].op) * c[i].len;
  return l;
}
static int32_t aln_end(const waln *a) { return a->pos + reference_length_from_cigar(a->cigar, a->n_cigar) - 1; }

/* filters/utils.go:141-147; RNAME "*" <=> refid < 0 after AddREFID */
static int is_strict_unmapped(const waln *a) { return (a->flag & ORC_UNMAPPED) || a->refid < 0 || a->pos == 0; }
static int is_strict_next_unmapped(const waln *a) { return (a->flag & ORC_NEXT_UNMAPPED) || a->next_refid < 0 || a->pnext == 0; }

/* filters/bqsr.go:225-244 + utils.go:121-139 */
int orc_recalibrate_aln(const orc_batch *b, const orc_header *h, const uint16_t *flags, uint64_t i) {
  if (b->has_sr && b->has_sr[i]) return 0;
  uint16_t flag = flags ? flags[i] : b->flag[i];
  uint8_t mapq = b->mapq[i];
  if (!(mapq > 0 && mapq < 255)) return 0;
  if (flag & (ORC_SECONDARY | ORC_DUPLICATE | ORC_QCFAILED)) return 0;
  if ((flag & ORC_UNMAPPED) || b->refid[i] < 0 || b->pos[i] == 0) return 0;
  if (!(b->pos[i] > 0)) return 0;
  uint32_t lseq = b->l_seq[i];
  if (lseq == 0) return 0;
  if ((uint64_t)lseq != b->qual_off[i + 1] - b->qual_off[i]) return 0;
  if (b->rgid[i] == ORC_NIL16) return 0;
  if (!(b->refid[i] < h->n_ref && b->pos[i] <= h->ref_len[b->refid[i]])) return 0; /* alignmentAgreesWithHeader */
  int32_t rl = 0, refl = 0;
  for (uint64_t k = b->cigar_off[i]; k < b->cigar_off[i + 1]; k++) {
    uint32_t c = b->cigar[k];
    char op = (c & 0xF) < 9 ? cigar_ops[c & 0xF] : '?';
    if (op == 'N') return 0;
    rl += consumes_read(op) * (int32_t)(c >> 4);
    refl += consumes_ref(op) * (int32_t)(c >> 4);
  }
  return refl >= 0 && (int32_t)lseq == rl;
}

/* filters/bqsr.go:287-299 */
static int read_starts_with_insertion(const cop *c, int n, int32_t *len) {
  for (int i = 0; i < n; i++) {
    if (c[i].op == 'I') { *len = c[i].len; return 1; }
    if (c[i].op == 'H' || c[i].op == 'S') continue;
    return 0;
  }
  return 0;
}

/* filters/utils.go:267-326 */
static int compute_read_coordinate(const cop *c, int n, int soft_start, int ref_index, int *falls) {
  int goal = ref_index - soft_start;
  *falls = 0;
  if (goal < 0) return -1;
  int read_bases = 0, ref_bases = 0;
  int falls_inside = 0, ends_just_before = 0, falls_or_before = 0;
  int index = 0;
  while (ref_bases != goal && index < n) {
    cop el = c[index++];
    int el_len = el.len;
    int shift = 0;
    if (consumes_ref(el.op) || el.op == 'S') {
      if (ref_bases + el_len < goal) shift = el_len;
      else shift = goal - ref_bases;
      ref_bases += shift;
    }
    if (ref_bases != goal) {
      read_bases += consumes_read(el.op) * el_len;
    } else {
      if (shift >= el_len && index == n) return -1;
      cop next; next.len = 0; next.op = 0;
      if (shift < el_len) {
        falls_inside = el.op == 'D' || el.op == 'N';
      } else {
        next = c[index++];
        if (next.op == 'I') {
          read_bases += next.len;
          if (index == n) return -1;
          next = c[index++];
        }
        ends_just_before = next.op == 'D' || next.op == 'N';
      }
      falls_or_before = ends_just_before || falls_inside;
      if (!falls_or_before) read_bases += consumes_read(el.op) * shift;
      else if (ends_just_before) read_bases += consumes_read(el.op) * (shift - 1);
      else if (falls_inside || (ends_just_before && (next.op == 'D' || next.op == 'N'))) read_bases--;
    }
  }
  if (ref_bases != goal) return -1;
  *falls = falls_or_before;
  return read_bases;
}

/* filters/utils.go:335-349; tail: 0 = left, 1 = right */
static int get_read_coordinate(const cop *c, int n, int soft_start, int ref_index, int right_tail, int *ok) {
  int falls;
  int read_bases = compute_read_coordinate(c, n, soft_start, ref_index, &falls);
  if (read_bases == -1) { *ok = 0; return -1; }
  if (right_tail && falls) read_bases++;
  if (!right_tail && read_bases == 0) {
    int32_t ins;
    if (read_starts_with_insertion(c, n, &ins)) {
      int32_t m = read_length_from_cigar(c, n) - 1;
      read_bases = ins < m ? ins : m;
    }
  }
  *ok = 1;
  return read_bases;
}

int orc_read_coordinate_for_reference_coordinate(const uint32_t *cigar, uint32_t n_cigar, int soft_start, int ref_index,
                                                 int right_tail, int *ok) {
  cop *c = (cop *)malloc((n_cigar + 1) * sizeof(cop));
  for (uint32_t k = 0; k < n_cigar; k++) { c[k].len = (int32_t)(cigar[k] >> 4); c[k].op = cigar_ops[cigar[k] & 0xF]; }
  int r = get_read_coordinate(c, (int)n_cigar, soft_start, ref_index, right_tail, ok);
  free(c);
  return r;
}

/* filters/utils.go:224-248 */
static int soft_start(const waln *a) {
  int32_t start = a->pos;
  for (int i = 0; i < a->n_cigar; i++) {
    if (a->cigar[i].op == 'S') start -= a->cigar[i].len;
    else if (a->cigar[i].op != 'H') break;
  }
  return start;
}
static int soft_end(const waln *a) {
  int32_t end = aln_end(a);
  int32_t se = end;
  for (int i = a->n_cigar - 1; i >= 0; i--) {
    if (a->cigar[i].op == 'S') se += a->cigar[i].len;
    else if (a->cigar[i].op != 'H') return se;
  }
  return end;
}

/* filters/utils.go:351-372 */
static int32_t hard_soft_offset(const cop *c, int n) {
  int32_t size = 0;
  int i = 0;
  for (; i < n; i++) { if (c[i].op == 'H') size += c[i].len; else break; }
  for (; i < n; i++) { if (c[i].op == 'S') size += c[i].len; else break; }
  return size;
}
/* :378-386 */
static int clip_alignment_shift(cop op, int cigar_length) {
  switch (op.op) {
  case 'I': return -cigar_length;
  case 'D': case 'N': return op.len;
  default: return 0;
  }
}

typedef struct { cop *v; int n, cap; } cvec;
static void cv_push(cvec *c, char op, int32_t len) {
  if (c->n == c->cap) { c->cap = c->cap * 2 + 8; c->v = (cop *)realloc(c->v, (size_t)c->cap * sizeof(cop)); }
  c->v[c->n].op = op; c->v[c->n].len = len; c->n++;
}

/* filters/utils.go:488-517 */
static void clean_hard_clipped_cigar(cvec *c) {
  int total = 0, index = 0;
  for (; index < c->n; index++) {
    char op = c->v[index].op;
    if (op == 'H' || op == 'D' || op == 'N') total += c->v[index].len;
    else break;
  }
  if (index > 0) {
    c->v[0].op = 'H'; c->v[0].len = total;
    memmove(c->v + 1, c->v + index, (size_t)(c->n - index) * sizeof(cop));
    c->n = 1 + (c->n - index);
  }
  total = 0;
  index = c->n - 1;
  for (; index >= 0; index--) {
    char op = c->v[index].op;
    if (op == 'H' || op == 'D' || op == 'N') total += c->v[index].len;
    else break;
  }
  if (index < c->n - 1) {
    c->n = index + 1;
    cv_push(c, 'H', total);
  }
}

/* filters/utils.go:406-486 */
static void hard_clip_cigar(const waln *a, int start, int stop, cvec *out) {
  const cop *cv = a->cigar;
  int n = a->n_cigar;
  int index = 0;
  int total_hard = stop - start + 1;
  int shift_acc = 0;
  out->n = 0;
  if (start == 0) {
    int ci = 0;
    for (int k = 0; k < n; k++) { /* Go: for cigarOpIndex, cigarOp = range cigarVec */
      ci = k;
      if (cv[k].op != 'H') break;
      total_hard += cv[k].len;
    }
    for (; index <= stop && ci < n; ci++) {
      cop op = cv[ci];
      int op_len = op.len;
      int shift = consumes_read(op.op) * op_len;
      if (index + shift == stop + 1) {
        shift_acc += clip_alignment_shift(op, op_len);
        cv_push(out, 'H', total_hard + shift_acc);
      } else if (index + shift > stop + 1) {
        int after = op_len - (stop - index + 1);
        shift_acc += clip_alignment_shift(op, stop - index + 1);
        cv_push(out, 'H', total_hard + shift_acc);
        cv_push(out, op.op, after);
      }
      index += shift;
      shift_acc += clip_alignment_shift(op, shift);
    }
    for (; ci < n; ci++) cv_push(out, cv[ci].op, cv[ci].len);
  } else {
    int ci = 0;
    for (; index < start && ci < n; ci++) {
      cop op = cv[ci];
      int op_len = op.len;
      int shift = consumes_read(op.op) * op_len;
      if (index + shift < start) {
        cv_push(out, op.op, op.len);
      } else {
        int after = start - index;
        shift_acc += clip_alignment_shift(op, op_len - (start - index));
        if (op.op == 'H') total_hard += after;
        else cv_push(out, op.op, after);
      }
      index += shift;
    }
    for (; ci < n; ci++) {
      cop op = cv[ci];
      shift_acc += clip_alignment_shift(op, op.len);
      if (op.op == 'H') total_hard += op.len;
    }
    cv_push(out, 'H', total_hard + shift_acc);
  }
  clean_hard_clipped_cigar(out);
}

/* filters/utils.go:388-404 */
static void hard_clip(waln *a, int start, int stop, cvec *scratch) {
  hard_clip_cigar(a, start, stop, scratch);
  int read_length = a->len;
  int new_length = read_length - (stop - start + 1);
  int copy_start = 0;
  if (start == 0) copy_start = stop + 1;
  int32_t old_off = hard_soft_offset(a->cigar, a->n_cigar);
  cigar_reserve(a, scratch->n + 4);
  memcpy(a->cigar, scratch->v, (size_t)scratch->n * sizeof(cop));
  a->n_cigar = scratch->n;
  a->off += copy_start;
  a->len = new_length;
  if (start == 0 && !is_strict_unmapped(a)) a->pos += hard_soft_offset(a->cigar, a->n_cigar) - old_off;
}

/* filters/utils.go:149-180, 214-222 */
static int hard_clip_adaptor_sequence(waln *a, cvec *scratch) {
  if (!(a->tlen != 0 && (a->flag & ORC_MULTIPLE) && !is_strict_unmapped(a) && !is_strict_next_unmapped(a) &&
        ((a->flag & ORC_REVERSED) != 0) != ((a->flag & ORC_NEXT_REVERSED) != 0)))
    return 0;
  int well, end_v;
  if (a->flag & ORC_REVERSED) { int32_t e = aln_end(a); well = e > a->pnext; end_v = e; }
  else { well = a->pos <= a->pnext + a->tlen; end_v = -1; }
  if (!well) return 0;
  int boundary;
  if (a->flag & ORC_REVERSED) boundary = (int)a->pnext - 1;
  else boundary = (int)a->pos + abs((int)a->tlen);
  /* isInsideRead */
  if (boundary < (int)a->pos) return 0;
  if (end_v < 0) end_v = aln_end(a);
  if (boundary > end_v) return 0;
  int ok;
  if (a->flag & ORC_REVERSED) {
    int stop = get_read_coordinate(a->cigar, a->n_cigar, soft_start(a), boundary, 0, &ok);
    if (!ok) return -1; /* reference: log.Panicf */
    hard_clip(a, 0, stop, scratch);
  } else {
    int start = get_read_coordinate(a->cigar, a->n_cigar, soft_start(a), boundary, 1, &ok);
    int stop = a->len - 1;
    if (!ok) return -1;
    hard_clip(a, start, stop, scratch);
  }
  return 0;
}

/* filters/utils.go:519-548 */
static void hard_clip_soft_clipped_bases(waln *a, cvec *scratch) {
  int read_index = 0, cut_left = -1, cut_right = -1, right_tail = 0;
  for (int i = 0; i < a->n_cigar; i++) {
    char key = a->cigar[i].op;
    int ln = a->cigar[i].len;
    if (key == 'S') {
      if (right_tail) cut_right = read_index;
      else cut_left = read_index + ln - 1;
    } else if (key != 'H') {
      right_tail = 1;
    }
    read_index += consumes_read(key) * ln;
  }
  if (cut_right >= 0) hard_clip(a, cut_right, a->len - 1, scratch);
  if (cut_left >= 0) hard_clip(a, 0, cut_left, scratch);
}

int orc_clip_for_bqsr(const orc_batch *b, uint64_t i, int *a_out, int *b_out, int32_t *new_pos, uint32_t *cigar_out, int cap) {
  waln a; memset(&a, 0, sizeof a);
  cvec sc; memset(&sc, 0, sizeof sc);
  load_aln(b, i, NULL, &a);
  int rc = hard_clip_adaptor_sequence(&a, &sc);
  if (rc == 0 && a.len > 0) hard_clip_soft_clipped_bases(&a, &sc);
  *a_out = a.off; *b_out = a.off + a.len; *new_pos = a.pos;
  int n = a.n_cigar < cap ? a.n_cigar : cap;
  for (int k = 0; k < n; k++) {
    const char *p = strchr(cigar_ops, a.cigar[k].op);
    cigar_out[k] = ((uint32_t)a.cigar[k].len << 4) | (uint32_t)(p ? p - cigar_ops : 15);
  }
  int nc = a.n_cigar;
  free(a.cigar); free(sc.v);
  return rc < 0 ? rc : nc;
}

/* ---------- intervals/intervals.go ---------- */
static int iv_cmp_start(const void *x, const void *y) {
  const orc_interval *a = (const orc_interval *)x, *b = (const orc_interval *)y;
  return a->start < b->start ? -1 : (a->start > b->start ? 1 : 0);
}
void orc_sort_by_start(orc_interval *iv, size_t n) {
  /* SortByStart is stable (:44-48); Flatten's result does not depend on the order of equal starts. */
  qsort(iv, n, sizeof(orc_interval), iv_cmp_start);
}
static int iv_extend(orc_interval *a, orc_interval b) { /* :88-96 */
  if (b.start > a->end) return 0;
  if (b.end > a->end) a->end = b.end;
  return 1;
}
size_t orc_flatten(orc_interval *iv, size_t len) { /* :103-117 */
  if (len == 0) return 0;
  for (size_t i = 0, n = len - 1; i < n; i++) {
    if (iv_extend(&iv[i], iv[i + 1])) {
      n++;
      for (size_t j = i + 1; j < n; j++) {
        if (!iv_extend(&iv[i], iv[j])) { i++; iv[i] = iv[j]; }
      }
      return i + 1;
    }
  }
  return len;
}
int orc_overlap(const orc_interval *iv, size_t n, int32_t start, int32_t end) { /* :139-153 */
  for (int64_t left = 0, right = (int64_t)n - 1; left <= right;) {
    int64_t mid = (left + right) / 2;
    if (iv[mid].start > end - 1) right = mid - 1;
    else if (iv[mid].end <= start - 1) left = mid + 1;
    else return 1;
  }
  return 0;
}
void orc_intersect(const orc_interval *iv, size_t n, int32_t start, int32_t end, size_t *lo, size_t *hi) { /* :166-173 */
  size_t a = 0, b = n;
  while (a < b) { size_t m = a + (b - a) / 2; if (!(iv[m].end >= start)) a = m + 1; else b = m; }
  *lo = a;
  a = 0; b = n;
  while (a < b) { size_t m = a + (b - a) / 2; if (!(iv[m].start > end)) a = m + 1; else b = m; }
  *hi = a;
}

/* ---------- covariates ---------- */
static int simple_base_index(uint8_t c) { /* bqsr.go:55-62 */
  switch (c) {
  case 'A': case 'a': case '*': return 0;
  case 'C': case 'c': return 1;
  case 'G': case 'g': return 2;
  case 'T': case 't': return 3;
  default: return -1;
  }
}
static int base_to_int(uint8_t c) { /* bqsr.go:247-252 */
  switch (c) {
  case 'a': case 'A': case '*': return 1;
  case 'c': case 'C': return 2;
  case 'g': case 'G': return 3;
  case 't': case 'T': return 4;
  default: return 0;
  }
}
static uint8_t base_complement(uint8_t c) { /* :303-310 */
  switch (c) {
  case 'A': case 'a': return 'T';
  case 'C': case 'c': return 'G';
  case 'G': case 'g': return 'C';
  case 'T': case 't': return 'A';
  default: return c;
  }
}

#define LENGTH_BITS 4
static int32_t key_from_context(const uint8_t *dna, int start, int end) { /* :64-76 */
  int32_t key = end - start;
  unsigned bit = LENGTH_BITS;
  for (int i = start; i < end; i++) {
    int bi = simple_base_index(dna[i]);
    if (bi == -1) return -1;
    key |= bi << bit;
    bit += 2;
  }
  return key;
}
/* contextWith, bqsr.go:87-131 with contextSize = 2; returns number of keys or -1 on the reference's panic path */
static int context_with(const uint8_t *bases, int read_length, int context_size, int32_t *keys) {
  int32_t mask = 0;
  for (int i = 0; i < context_size; i++) mask = (mask << 2) | 3;
  mask <<= LENGTH_BITS;
  int nk = 0;
  for (int i = 1; i < context_size && i <= read_length; i++) keys[nk++] = -1;
  if (read_length < context_size) return nk;
  unsigned new_base_offset = 2 * (context_size - 1) + LENGTH_BITS;
  int32_t cur = key_from_context(bases, 0, context_size);
  keys[nk++] = cur;
  int n_penalty = 0;
  if (cur == -1) {
    cur = 0;
    n_penalty = context_size - 1;
    unsigned offset = new_base_offset;
    while (bases[n_penalty] != 'N') {
      int bi = simple_base_index(bases[n_penalty]);
      cur |= bi << offset;
      offset -= 2;
      n_penalty--;
      if (n_penalty < 0) return -1; /* reference walks off the slice start and panics (quirk 13) */
    }
  }
  for (int ci = context_size; ci < read_length; ci++) {
    int bi = simple_base_index(bases[ci]);
    if (bi == -1) {
      n_penalty = context_size;
      cur = 0;
    } else {
      cur = (cur >> 2) & mask;
      cur |= bi << new_base_offset;
      cur |= context_size;
    }
    if (n_penalty == 0) keys[nk++] = cur;
    else { n_penalty--; keys[nk++] = -1; }
  }
  return nk;
}
void orc_context_with(const uint8_t *bases, int n, int32_t *keys_out) { context_with(bases, n, 2, keys_out); }

/* computeStrandedClippedSeq, bqsr.go:312-362; returns 0 if nil */
static int stranded_clipped_seq(const waln *a, uint8_t *out) {
  int n = a->len;
  int left = n;
  for (int i = 0; i < left; i++) if (wqual(a, i) > 2) { left = i; break; }
  int right = left - 1;
  for (int i = n - 1; i >= left; i--) if (wqual(a, i) > 2) { right = i; break; }
  if (left > right) return 0;
  if (a->flag & ORC_REVERSED) {
    int j = -1;
    for (int i = right + 1; i < n; i++) out[++j] = 'N';
    for (int i = right; i >= left; i--) out[++j] = base_complement((uint8_t)wbase(a, i));
    for (int i = 0; i < left; i++) out[++j] = 'N';
  } else {
    for (int i = 0; i < left; i++) out[i] = 'N';
    for (int i = left; i <= right; i++) out[i] = (uint8_t)wbase(a, i);
    for (int i = right + 1; i < n; i++) out[i] = 'N';
  }
  return 1;
}
/* computeBaseContextCovariate :140-146; returns number of keys (0 if stranded seq is nil), -1 on panic path */
static int base_context_covariate(const waln *a, uint8_t *sbuf, int32_t *keys) {
  int nk;
  if (!stranded_clipped_seq(a, sbuf)) nk = context_with(sbuf, 0, 2, keys);
  else nk = context_with(sbuf, a->len, 2, keys);
  if (nk < 0) return -1;
  if (a->flag & ORC_REVERSED)
    for (int i = 0, j = nk - 1; i < j; i++, j--) { int32_t t = keys[i]; keys[i] = keys[j]; keys[j] = t; }
  return nk;
}
/* prepareCycleCovariates + computeBaseCycleCovariate :376-387 */
static void cycle_params(uint16_t flag, int len, int *factor, int *incr) {
  int reversed = (flag & ORC_REVERSED) >> 4;
  int last = (flag & ORC_LAST) >> 7;
  int rof = 1 - 2 * last;
  *factor = rof + reversed * (len - 1) * rof;
  *incr = (1 - 2 * reversed) * rof;
}
int orc_cycle(uint16_t flag, int l_seq, int index) {
  int f, inc;
  cycle_params(flag, l_seq, &f, &inc);
  return f + index * inc;
}

/* calculateSkipSlice :389-414 */
static void skip_slice(const waln *a, const orc_interval *sites, size_t n_sites, uint8_t *skip) {
  int n = a->len;
  int ss = soft_start(a), se = soft_end(a);
  memset(skip, 0, (size_t)n);
  size_t lo, hi;
  orc_intersect(sites, n_sites, ss, se, &lo, &hi);
  for (size_t s = lo; s < hi; s++) {
    int ok;
    int fs = get_read_coordinate(a->cigar, a->n_cigar, ss, sites[s].start, 0, &ok);
    if (!ok || fs < 0) fs = 0;
    int fe = get_read_coordinate(a->cigar, a->n_cigar, ss, sites[s].end, 0, &ok);
    if (!ok || fe > n - 1) fe = n - 1;
    for (int i = fs; i <= fe; i++) skip[i] = 1;
  }
}

/* computeSnpEvents :254-285.  Reference bytes beyond the contig end make the Go code panic (index out of range);
 * here they read as code 0 ('N'). */
static void snp_events(const waln *a, const uint8_t *ref, int64_t ref_len, int *snps) {
  for (int i = 0; i < a->len; i++) snps[i] = 0;
  int i = 0;
  int64_t j = (int64_t)a->pos - 1;
  for (int c = 0; c < a->n_cigar; c++) {
    int ln = a->cigar[c].len;
    switch (a->cigar[c].op) {
    case 'M': case '=': case 'X':
      for (int k = 0; k < ln; k++) {
        int rb = (j >= 0 && j < ref_len) ? base_to_int(ref[j]) : 0;
        if (i < a->len && base_to_int((uint8_t)wbase(a, i)) != rb) snps[i] = 1;
        i++; j++;
      }
      break;
    case 'D': case 'N': j += ln; break;
    case 'I': case 'S': i += ln; break;
    default: break;
    }
  }
}

/* Recalibrate :467-551 over the records [lo, hi) (the body of the RangeReduce worker :471-540); tables are added to */
static int gather_range(const orc_batch *b, const orc_header *h, const orc_bqsr_ref *r, const uint16_t *flags, int max_cycle,
                        uint64_t lo, uint64_t hi, uint32_t maxl, int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl);

int orc_bqsr_gather(const orc_batch *b, const orc_header *h, const orc_bqsr_ref *r, const uint16_t *flags, int max_cycle,
                    int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl) {
  int ncyc = 2 * max_cycle + 1;
  memset(qual_tbl, 0, (size_t)h->n_cov * ORC_NQUAL * 2 * sizeof(int64_t));
  memset(cycle_tbl, 0, (size_t)h->n_cov * ORC_NQUAL * ncyc * 2 * sizeof(int64_t));
  memset(ctx_tbl, 0, (size_t)h->n_cov * ORC_NQUAL * ORC_NCTX * 2 * sizeof(int64_t));
  uint32_t maxl = 1;
  for (uint64_t i = 0; i < b->n; i++) if (b->l_seq[i] > maxl) maxl = b->l_seq[i];
  return gather_range(b, h, r, flags, max_cycle, 0, b->n, maxl, qual_tbl, cycle_tbl, ctx_tbl);
}

/* the same with thread-private tables over sub-ranges, summed at the end: parallel.RangeReduce + bqsrTable.merge (:210-223, 471) */
#include <omp.h>
int orc_bqsr_gather_mt(const orc_batch *b, const orc_header *h, const orc_bqsr_ref *r, const uint16_t *flags, int max_cycle,
                       int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl, int n_threads) {
  int ncyc = 2 * max_cycle + 1;
  size_t nq = (size_t)h->n_cov * ORC_NQUAL * 2, nc = nq * ncyc, nx = nq * ORC_NCTX;
  memset(qual_tbl, 0, nq * sizeof(int64_t));
  memset(cycle_tbl, 0, nc * sizeof(int64_t));
  memset(ctx_tbl, 0, nx * sizeof(int64_t));
  if (n_threads < 1) n_threads = omp_get_max_threads();
  uint32_t maxl = 1;
  for (uint64_t i = 0; i < b->n; i++) if (b->l_seq[i] > maxl) maxl = b->l_seq[i];
  int rc_all = 0;
#pragma omp parallel num_threads(n_threads)
  {
    int64_t *t = (int64_t *)calloc(nq + nc + nx, sizeof(int64_t));
    int rc = t ? 0 : -2;
#pragma omp for schedule(dynamic, 1)
    for (uint64_t c = 0; c < (b->n + 16383) / 16384; c++) {
      uint64_t lo = c * 16384, hi = lo + 16384 < b->n ? lo + 16384 : b->n;
      if (rc == 0) rc = gather_range(b, h, r, flags, max_cycle, lo, hi, maxl, t, t + nq, t + nq + nc);
    }
#pragma omp critical
    {
      if (rc) rc_all = rc;
      if (t) {
        for (size_t k = 0; k < nq; k++) qual_tbl[k] += t[k];
        for (size_t k = 0; k < nc; k++) cycle_tbl[k] += t[nq + k];
        for (size_t k = 0; k < nx; k++) ctx_tbl[k] += t[nq + nc + k];
      }
    }
    free(t);
  }
  return rc_all;
}

static int gather_range(const orc_batch *b, const orc_header *h, const orc_bqsr_ref *r, const uint16_t *flags, int max_cycle,
                        uint64_t lo, uint64_t hi, uint32_t maxl, int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl) {
  int ncyc = 2 * max_cycle + 1;
  waln a; memset(&a, 0, sizeof a);
  cvec sc; memset(&sc, 0, sizeof sc);
  int *snps = (int *)malloc(maxl * sizeof(int));
  uint8_t *sbuf = (uint8_t *)malloc(maxl + 1);
  int32_t *ctx = (int32_t *)malloc((maxl + 2) * sizeof(int32_t));
  uint8_t *skip = (uint8_t *)malloc(maxl + 1);
  int rc = 0;
  for (uint64_t i = lo; i < hi && rc == 0; i++) {
    if (!orc_recalibrate_aln(b, h, flags, i)) continue;
    load_aln(b, i, flags, &a);
    if (hard_clip_adaptor_sequence(&a, &sc) < 0) { rc = -5; break; }
    if (a.len == 0) continue;
    hard_clip_soft_clipped_bases(&a, &sc);
    if (a.len == 0) continue;
    int refid = a.refid;
    skip_slice(&a, r->sites ? r->sites[refid] : NULL, r->sites ? (size_t)r->n_sites[refid] : 0, skip);
    snp_events(&a, r->ref_seq[refid], r->ref_seq_len[refid], snps);
    uint16_t rg = b->rgid[i];
    int cov = (int32_t)rg < h->n_rg ? h->rg_cov[rg] : -1;
    if (cov < 0 || cov >= h->n_cov) { rc = -6; break; }
    int cf, ci;
    cycle_params(a.flag, a.len, &cf, &ci);
    int nk = base_context_covariate(&a, sbuf, ctx);
    if (nk < 0) { rc = -7; break; }
    for (int k = 0; k < a.len; k++) {
      if (skip[k]) continue;
      if (simple_base_index((uint8_t)wbase(&a, k)) < 0) continue;
      uint8_t q = wqual(&a, k);
      if (q < 6) continue;
      if (q >= ORC_NQUAL) { rc = -8; break; }
      int err = snps[k];
      int64_t *e = qual_tbl + ((size_t)cov * ORC_NQUAL + q) * 2;
      e[0]++; e[1] += err;
      int cyc = cf + k * ci;
      if (cyc > max_cycle || cyc < -max_cycle) { rc = -9; break; } /* checkCycleCovariate :364-369 panics */
      e = cycle_tbl + (((size_t)cov * ORC_NQUAL + q) * ncyc + (size_t)(cyc + max_cycle)) * 2;
      e[0]++; e[1] += err;
      if (nk > 0 && ctx[k] >= 0) {
        e = ctx_tbl + (((size_t)cov * ORC_NQUAL + q) * ORC_NCTX + (size_t)((ctx[k] >> 4) & 15)) * 2;
        e[0]++; e[1] += err;
      }
    }
  }
  free(a.cigar); free(sc.v); free(snps); free(sbuf); free(ctx); free(skip);
  return rc;
}

/* ---------- float finalisation ---------- */
#define GO_LN2_OVER_LN10 0.30102999566398119521373889472449302676818988146210854131
#define GO_1_OVER_LN2 1.44269504088896340735992468100189213742664595415298593413
#define GO_LOG10E 0.43429448190325182765112891891660508229439700580366656611445378316586464920887077

static double go_log2(double x) { /* math/log10.go: log2 */
  int exp;
  double frac = frexp(x, &exp);
  if (frac == 0.5) return (double)(exp - 1);
  return orc_go_log(frac) * GO_1_OVER_LN2 + (double)exp;
}
double orc_go_log10(double x) { return go_log2(x) * GO_LN2_OVER_LN10; } /* math.Log10; filters/unpedantic.go:28 */

/* math.Pow(x, y) for finite x > 0 (math/pow.go structure) */
static double go_pow(double x, double y) {
  if (y == 0 || x == 1) return 1;
  if (y == 1) return x;
  if (y == 0.5) return sqrt(x);
  if (y == -0.5) return 1 / sqrt(x);
  double yi, yf;
  yf = modf(fabs(y), &yi);
  double a1 = 1.0;
  int ae = 0;
  if (yf != 0) {
    if (yf > 0.5) { yf--; yi++; }
    a1 = orc_go_exp(yf * orc_go_log(x));  /* (math.Exp: the pure-Go function, see orc_gomath.c for the amd64 caveat) */
  }
  int xe;
  double x1 = frexp(x, &xe);
  for (int64_t i = (int64_t)yi; i != 0; i >>= 1) {
    if (xe < -(1 << 12) || (1 << 12) < xe) {
      ae += xe;
      break;
    }
    if (i & 1) { a1 *= x1; ae += xe; }
    x1 *= x1;
    xe <<= 1;
    if (x1 < .5) { x1 += x1; xe--; }
  }
  if (y < 0) { a1 = 1 / a1; ae = -ae; }
  return ldexp(a1, ae);
}
double orc_go_pow10(double y) { return go_pow(10, y); }

static double quality_to_error_probability(double phred) { return go_pow(10, phred / -10); } /* :561 */

static const double prior_cache[21] = { /* :569-591 */
  -0.045757490560675115, -0.9143464543671788, -3.5201133457866898, -7.863058164819208, -13.943180911464733,
  -21.760481585723266, -31.314960187594806, -42.606616717079355, -55.63545117417691, -70.40146355888747,
  -86.90465387121104, -105.14502211114761, -125.1225682786972, -146.83729237385978, -170.2891943966354,
  -195.47827434702398, -222.4045322250256, -251.06796803064023, -281.46858176386786, -313.60637342472336,
  -1.7976931348623157e308};

static double log10_qual_empirical_prior(double emp, double rep) { /* :593-596 */
  int d = (int)(emp - rep);
  if (d < 0) d = -d;
  if (d > 20) d = 20;
  return prior_cache[d];
}
static double log10_gamma(int64_t n) { return orc_go_lgamma((double)n) * GO_LOG10E; } /* :598-601 */
static double log10_binomial_coefficient(int64_t n, int64_t k) { return log10_gamma(n + 1) - log10_gamma(k + 1) - log10_gamma(n - k + 1); }
static double log10_binomial_probability(int64_t n, int64_t k, double log10p) { /* :607-613 */
  if (log10p == 0.0) return -DBL_MAX;
  double log10minp = orc_go_log10(1.0 - go_pow(10, log10p));
  return log10_binomial_coefficient(n, k) + log10p * (double)k + log10minp * (double)(n - k);
}
static double log10_qual_empirical_likelihood(double emp, int64_t obs, int64_t mism) { /* :615-621 */
  if (obs == 0) return 0.0;
  return log10_binomial_probability(obs, mism, emp / -10.0);
}
uint8_t orc_bayesian_estimate(int64_t observations, int64_t mismatches, double prior) { /* :623-642 */
  const int64_t max_obs = 2147483647LL - 1;
  if (observations > max_obs) {
    mismatches = (int64_t)round((double)mismatches * ((double)max_obs / (double)observations));
    observations = max_obs;
  }
  double max = -DBL_MAX;
  uint8_t max_i = 0;
  for (int i = 0; i < 61; i++) {
    double fi = (double)i;
    double post = log10_qual_empirical_prior(fi, prior) + log10_qual_empirical_likelihood(fi, observations, mismatches);
    if (max < post) { max = post; max_i = (uint8_t)i; }
  }
  return max_i;
}
static uint8_t calc_empirical_quality(int64_t obs, int64_t mism, double prior) { /* :644-649 */
  uint8_t q = orc_bayesian_estimate(obs + 2, mism + 1, prior);
  return q < 93 ? q : 93;
}

struct orc_bqsr_final {
  int n_cov, max_cycle, ncyc;
  int64_t *qual_tbl, *cycle_tbl, *ctx_tbl;
  uint8_t *qual_emp, *cycle_emp, *ctx_emp; /* 255 = absent */
  double *c_reported; uint8_t *c_emp; int64_t *c_obs, *c_mism; uint8_t *c_present;
  double *m_dglobal, *m_dreported, *m_cyc, *m_ctx; uint8_t *h_dglobal, *h_dreported, *h_cyc, *h_ctx; /* lazily filled caches */
};

orc_bqsr_final *orc_bqsr_finalize(int n_cov, int max_cycle, const int64_t *qual_tbl, const int64_t *cycle_tbl, const int64_t *ctx_tbl) {
  orc_bqsr_final *f = (orc_bqsr_final *)calloc(1, sizeof *f);
  int ncyc = 2 * max_cycle + 1;
  f->n_cov = n_cov; f->max_cycle = max_cycle; f->ncyc = ncyc;
  size_t nq = (size_t)n_cov * ORC_NQUAL, nc = nq * ncyc, nx = nq * ORC_NCTX;
  f->qual_tbl = (int64_t *)malloc(nq * 2 * sizeof(int64_t)); memcpy(f->qual_tbl, qual_tbl, nq * 2 * sizeof(int64_t));
  f->cycle_tbl = (int64_t *)malloc(nc * 2 * sizeof(int64_t)); memcpy(f->cycle_tbl, cycle_tbl, nc * 2 * sizeof(int64_t));
  f->ctx_tbl = (int64_t *)malloc(nx * 2 * sizeof(int64_t)); memcpy(f->ctx_tbl, ctx_tbl, nx * 2 * sizeof(int64_t));
  f->qual_emp = (uint8_t *)malloc(nq); f->cycle_emp = (uint8_t *)malloc(nc); f->ctx_emp = (uint8_t *)malloc(nx);
  /* FinalizeBQSRTables :677-694: prior = the entry's reported quality */
  for (size_t i = 0; i < nq; i++) {
    int q = (int)(i % ORC_NQUAL);
    f->qual_emp[i] = qual_tbl[2 * i] > 0 ? calc_empirical_quality(qual_tbl[2 * i], qual_tbl[2 * i + 1], (double)q) : 255;
  }
  for (size_t i = 0; i < nc; i++) {
    int q = (int)((i / ncyc) % ORC_NQUAL);
    f->cycle_emp[i] = cycle_tbl[2 * i] > 0 ? calc_empirical_quality(cycle_tbl[2 * i], cycle_tbl[2 * i + 1], (double)q) : 255;
  }
  for (size_t i = 0; i < nx; i++) {
    int q = (int)((i / ORC_NCTX) % ORC_NQUAL);
    f->ctx_emp[i] = ctx_tbl[2 * i] > 0 ? calc_empirical_quality(ctx_tbl[2 * i], ctx_tbl[2 * i + 1], (double)q) : 255;
  }
  /* initializeCombinedBQSRTable :655-674, (rg, qual) entries visited in ascending qual (canonical choice; Go iterates a map) */
  f->c_reported = (double *)calloc(n_cov, sizeof(double)); f->c_emp = (uint8_t *)calloc(n_cov, 1);
  f->c_obs = (int64_t *)calloc(n_cov, sizeof(int64_t)); f->c_mism = (int64_t *)calloc(n_cov, sizeof(int64_t));
  f->c_present = (uint8_t *)calloc(n_cov, 1);
  for (int c = 0; c < n_cov; c++) {
    for (int q = 0; q < ORC_NQUAL; q++) {
      int64_t obs = qual_tbl[((size_t)c * ORC_NQUAL + q) * 2], mism = qual_tbl[((size_t)c * ORC_NQUAL + q) * 2 + 1];
      if (obs <= 0) continue;
      if (f->c_present[c]) {
        double sum_errors = (double)f->c_obs[c] * quality_to_error_probability(f->c_reported[c]) +
                            (double)obs * quality_to_error_probability((double)q);
        f->c_obs[c] += obs; f->c_mism[c] += mism;
        f->c_reported[c] = -10 * orc_go_log10(sum_errors / (double)f->c_obs[c]);
      } else {
        f->c_present[c] = 1; f->c_reported[c] = (double)q; f->c_obs[c] = obs; f->c_mism[c] = mism;
      }
    }
    if (f->c_present[c]) f->c_emp[c] = calc_empirical_quality(f->c_obs[c], f->c_mism[c], f->c_reported[c]);
  }
  return f;
}
void orc_bqsr_final_free(orc_bqsr_final *f) {
  if (!f) return;
  free(f->qual_tbl); free(f->cycle_tbl); free(f->ctx_tbl); free(f->qual_emp); free(f->cycle_emp); free(f->ctx_emp);
  free(f->c_reported); free(f->c_emp); free(f->c_obs); free(f->c_mism); free(f->c_present);
  free(f->m_dglobal); free(f->m_dreported); free(f->m_cyc); free(f->m_ctx); free(f->h_dglobal); free(f->h_dreported); free(f->h_cyc); free(f->h_ctx);
  free(f);
}
void orc_bqsr_final_empirical(const orc_bqsr_final *f, uint8_t *qual_emp, uint8_t *cycle_emp, uint8_t *ctx_emp) {
  size_t nq = (size_t)f->n_cov * ORC_NQUAL;
  memcpy(qual_emp, f->qual_emp, nq); memcpy(cycle_emp, f->cycle_emp, nq * f->ncyc); memcpy(ctx_emp, f->ctx_emp, nq * ORC_NCTX);
}
void orc_bqsr_final_combined(const orc_bqsr_final *f, double *reported, uint8_t *emp, int64_t *obs, int64_t *mism, uint8_t *present) {
  for (int c = 0; c < f->n_cov; c++) {
    reported[c] = f->c_reported[c]; emp[c] = f->c_emp[c]; obs[c] = f->c_obs[c]; mism[c] = f->c_mism[c]; present[c] = f->c_present[c];
  }
}

static int error_probability_to_quality(double prob) { /* :701-706 */
  if (prob == 0.0) return 93;
  int q = (int)round(-10 * orc_go_log10(prob));
  if (q > 93) q = 93;
  return q < 1 ? 1 : q;
}
static double quality_to_probability(double phred) { return 1 - go_pow(10, phred / -10); } /* :565 */

void orc_static_quantized_scores(const uint8_t *quals_in, int n, uint8_t *out) { /* :710-744 */
  uint8_t quals[256];
  if (n > 256) n = 256;
  memcpy(quals, quals_in, (size_t)n);
  memset(out, 0, 254);
  for (int i = 0; i < 6; i++) out[i] = (uint8_t)i;
  if (n == 1) { for (int i = 6; i < 254; i++) out[i] = quals[0]; return; }
  for (int i = 1; i < n; i++) { uint8_t v = quals[i]; int k = i; while (k > 0 && quals[k - 1] > v) { quals[k] = quals[k - 1]; k--; } quals[k] = v; }
  uint8_t prev_q = 6;
  double prev_p = quality_to_probability((double)prev_q);
  for (int k = 0; k < n; k++) {
    uint8_t next_q = quals[k];
    /* NB: the reference updates prevProb/prevQual INSIDE the inner loop (bqsr.go:727-737); restated literally */
    for (uint8_t i = prev_q; i < next_q; i++) {
      double next_p = quality_to_probability((double)next_q);
      double ip = quality_to_probability((double)i);
      if (ip - prev_p > next_p - ip) out[i] = next_q; else out[i] = prev_q;
      prev_p = next_p;
      prev_q = next_q;
    }
  }
  for (int i = prev_q; i < 254; i++) out[i] = prev_q;
}

typedef struct { int next; double error_rate; int64_t nobs, leaf_nobs, nerrors; } qinterval; /* :746-752 */
static double calc_error_rate(int64_t nobs, int64_t nerrors) { return nobs == 0 ? 0.0 : (double)(nerrors + 1) / (double)(nobs + 1); }
static double leaf_penalty(int k, const qinterval *iv, double global) { /* :781-787 */
  if (k <= 6) return 0.0;
  return fabs(orc_go_log10(iv[k].error_rate) - orc_go_log10(global)) * (double)iv[k].leaf_nobs;
}
static double merge_penalty(int i, int j, const qinterval *iv, int n) { /* :796-819 */
  int64_t nobs = iv[i].nobs + iv[j].nobs, nerr = iv[i].nerrors + iv[j].nerrors;
  double rate = calc_error_rate(nobs, nerr);
  if (rate == 0) return 0.0;
  double si = 0, sj = 0;
  for (int k = i; k < j; k++) si += leaf_penalty(k, iv, rate);
  int kend = iv[j].next >= 0 ? iv[j].next : n;
  for (int k = j; k < kend; k++) sj += leaf_penalty(k, iv, rate);
  return si + sj;
}
static int merge_minimal(qinterval *iv, int n) { /* :821-850 */
  int i = 0;
  int j = iv[0].next;
  if (j < 0) return 0;
  int min_i = i;
  double pen = merge_penalty(i, j, iv, n);
  for (;;) {
    i = j;
    j = iv[i].next;
    if (j < 0) break;
    double p = merge_penalty(i, j, iv, n);
    if (p < pen) { min_i = i; pen = p; }
  }
  qinterval *a = &iv[min_i], *bq = &iv[a->next];
  int64_t nobs = a->nobs + bq->nobs, nerr = a->nerrors + bq->nerrors;
  a->next = bq->next; a->nobs = nobs; a->nerrors = nerr;
  return 1;
}
void orc_bqsr_quantize(const orc_bqsr_final *f, int levels, int64_t *counts, uint8_t *scores) { /* :863-899 */
  for (int i = 0; i < 94; i++) { counts[i] = 0; scores[i] = 0; }
  if (levels == 0) { for (int i = 0; i < 94; i++) scores[i] = (uint8_t)i; return; }
  size_t nq = (size_t)f->n_cov * ORC_NQUAL;
  for (size_t i = 0; i < nq; i++)
    if (f->qual_tbl[2 * i] > 0) counts[f->qual_emp[i]] += f->qual_tbl[2 * i];
  qinterval iv[94];
  for (int i = 0; i < 94; i++) { /* initializeQuantizationIntervals :761-779 */
    double er = quality_to_error_probability((double)i);
    iv[i].next = (i + 1 == 94) ? -1 : i + 1;
    iv[i].error_rate = er; iv[i].nobs = counts[i]; iv[i].leaf_nobs = counts[i];
    iv[i].nerrors = (int64_t)((double)counts[i] * er);
  }
  int n = 94;
  while (n > levels) { if (merge_minimal(iv, 94)) n--; else break; }
  for (int i = 0; i >= 0;) {
    uint8_t qs;
    int leaf = iv[i].next < 0 ? (i == 93) : (iv[i].next == i + 1); /* leafInterval :754-759 */
    if (leaf) qs = (uint8_t)i;
    else qs = (uint8_t)error_probability_to_quality(calc_error_rate(iv[i].nobs, iv[i].nerrors));
    int kend = iv[i].next >= 0 ? iv[i].next : 94;
    for (int k = i; k < kend; k++) scores[k] = qs;
    i = iv[i].next;
  }
}

/* estimateHierarchicalBayesianQuality :901-919 + the final mapping :995-999.
 * The three calculateEmpiricalQuality sub-results are pure functions of (cov), (cov,qual), (cov,qual,cycle) and
 * (cov,qual,context) — the conditional prior depends on (cov,qual) only — so they are cached per key; the arithmetic
 * and its order are exactly the reference's. */
static double *memo_get(double **arr, uint8_t **have, size_t n) {
  if (!*arr) { *arr = (double *)calloc(n, sizeof(double)); *have = (uint8_t *)calloc(n, 1); }
  return *arr;
}
uint8_t orc_bqsr_recal_qual(const orc_bqsr_final *fc, int cov, int qual, int cycle, int ctx_key, const uint8_t *quantized,
                            const uint8_t *static_q) {
  orc_bqsr_final *f = (orc_bqsr_final *)fc; /* caches only */
  size_t nq = (size_t)f->n_cov * ORC_NQUAL;
  double epsilon = f->c_reported[cov]; /* globalQualityScorePrior = -1 -> epsilon = reportedQuality :959-964 */
  memo_get(&f->m_dglobal, &f->h_dglobal, (size_t)f->n_cov);
  if (!f->h_dglobal[cov]) {
    f->m_dglobal[cov] = (double)calc_empirical_quality(f->c_obs[cov], f->c_mism[cov], epsilon) - epsilon;
    f->h_dglobal[cov] = 1;
  }
  double d_global = f->m_dglobal[cov];
  size_t qi = (size_t)cov * ORC_NQUAL + (size_t)qual;
  memo_get(&f->m_dreported, &f->h_dreported, nq);
  if (!f->h_dreported[qi]) {
    double d = 0;
    if (f->qual_tbl[2 * qi] > 0)
      d = (double)calc_empirical_quality(f->qual_tbl[2 * qi], f->qual_tbl[2 * qi + 1], d_global + epsilon) - d_global - epsilon;
    f->m_dreported[qi] = d; f->h_dreported[qi] = 1;
  }
  double d_reported = f->m_dreported[qi];
  double d_cov = 0;
  double cond = d_reported + d_global + epsilon;
  if (cycle >= -f->max_cycle && cycle <= f->max_cycle) {
    size_t ci = qi * f->ncyc + (size_t)(cycle + f->max_cycle);
    if (f->cycle_tbl[2 * ci] > 0) {
      memo_get(&f->m_cyc, &f->h_cyc, nq * f->ncyc);
      if (!f->h_cyc[ci]) { f->m_cyc[ci] = (double)calc_empirical_quality(f->cycle_tbl[2 * ci], f->cycle_tbl[2 * ci + 1], cond) - cond; f->h_cyc[ci] = 1; }
      d_cov = f->m_cyc[ci];
    }
  }
  if (ctx_key >= 0) {
    size_t xi = qi * ORC_NCTX + (size_t)((ctx_key >> 4) & 15);
    if (f->ctx_tbl[2 * xi] > 0) {
      memo_get(&f->m_ctx, &f->h_ctx, nq * ORC_NCTX);
      if (!f->h_ctx[xi]) { f->m_ctx[xi] = (double)calc_empirical_quality(f->ctx_tbl[2 * xi], f->ctx_tbl[2 * xi + 1], cond) - cond; f->h_ctx[xi] = 1; }
      d_cov += f->m_ctx[xi];
    }
  }
  double est = cond + d_cov;
  int r = (int)round(est);
  if (r > 93) r = 93;
  if (r < 1) r = 1;
  uint8_t out = quantized[r];
  if (static_q) out = static_q[out];
  return out;
}

/* ApplyBQSR :936-1005 */
static int apply_range(const orc_batch *b, const orc_header *h, const orc_bqsr_final *f, const uint8_t *quantized, const uint8_t *static_q,
                       int max_cycle, int16_t *memo, uint32_t maxl, uint64_t lo, uint64_t hi, uint8_t *qual_out);

int orc_bqsr_apply(const orc_batch *b, const orc_header *h, const orc_bqsr_final *f, int quantize_levels, const uint8_t *sqq,
                   int n_sqq, int max_cycle, uint8_t *qual_out) {
  return orc_bqsr_apply_mt(b, h, f, quantize_levels, sqq, n_sqq, max_cycle, qual_out, 1);
}

/* n_threads > 1: the reads are split among threads as the reference's batch pipeline does (sam/filter-pipeline.go:269-278); the
 * memo (one per worker in the reference, :944-946) is shared here: every entry has one possible value */
int orc_bqsr_apply_mt(const orc_batch *b, const orc_header *h, const orc_bqsr_final *f, int quantize_levels, const uint8_t *sqq,
                      int n_sqq, int max_cycle, uint8_t *qual_out, int n_threads) {
  uint8_t static_q[254];
  int64_t counts[94];
  uint8_t quantized[94];
  if (n_sqq > 0) orc_static_quantized_scores(sqq, n_sqq, static_q);
  orc_bqsr_quantize(f, quantize_levels, counts, quantized);
  int ncyc = 2 * max_cycle + 1;
  size_t memo_n = (size_t)f->n_cov * ORC_NQUAL * ncyc * 17;
  int16_t *memo = (int16_t *)malloc(memo_n * sizeof(int16_t));
  for (size_t i = 0; i < memo_n; i++) memo[i] = -1;
  uint32_t maxl = 1;
  for (uint64_t i = 0; i < b->n; i++) if (b->l_seq[i] > maxl) maxl = b->l_seq[i];
  if (n_threads < 1) n_threads = omp_get_max_threads();
  int rc_all = 0;
  if (n_threads == 1) {
    memcpy(qual_out, b->qual, b->qual_off[b->n]);
    rc_all = apply_range(b, h, f, quantized, n_sqq > 0 ? static_q : NULL, max_cycle, memo, maxl, 0, b->n, qual_out);
  } else {
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
    for (uint64_t c = 0; c < (b->n + 16383) / 16384; c++) {
      uint64_t lo = c * 16384, hi = lo + 16384 < b->n ? lo + 16384 : b->n;
      memcpy(qual_out + b->qual_off[lo], b->qual + b->qual_off[lo], b->qual_off[hi] - b->qual_off[lo]);
      int rc = apply_range(b, h, f, quantized, n_sqq > 0 ? static_q : NULL, max_cycle, memo, maxl, lo, hi, qual_out);
      if (rc) {
#pragma omp atomic write
        rc_all = rc;
      }
    }
  }
  free(memo);
  return rc_all;
}

static int apply_range(const orc_batch *b, const orc_header *h, const orc_bqsr_final *f, const uint8_t *quantized, const uint8_t *static_q,
                       int max_cycle, int16_t *memo, uint32_t maxl, uint64_t lo, uint64_t hi, uint8_t *qual_out) {
  int ncyc = 2 * max_cycle + 1;
  uint8_t *sbuf = (uint8_t *)malloc(maxl + 1);
  int32_t *ctx = (int32_t *)malloc((maxl + 2) * sizeof(int32_t));
  waln a; memset(&a, 0, sizeof a);
  int rc = 0;
  for (uint64_t i = lo; i < hi; i++) {
    uint16_t rg = b->rgid[i];
    if (rg == ORC_NIL16) { rc = -10; break; } /* readGroupCovariate panics :38 */
    int cov = (int32_t)rg < h->n_rg ? h->rg_cov[rg] : -1;
    if (cov < 0 || cov >= f->n_cov || !f->c_present[cov]) continue; /* no recalibration, bqsr table empty :953-955 */
    load_aln(b, i, NULL, &a);
    if ((uint64_t)a.len != b->qual_off[i + 1] - b->qual_off[i]) { rc = -11; break; }
    int cf, ci;
    cycle_params(a.flag, a.len, &cf, &ci);
    int nk = base_context_covariate(&a, sbuf, ctx);
    if (nk < 0) { rc = -7; break; }
    uint8_t *qo = qual_out + b->qual_off[i];
    for (int k = 0; k < a.len; k++) {
      uint8_t q = a.qual[k];
      if (q < 6) continue;
      if (q >= ORC_NQUAL) { rc = -8; break; }
      int cyc = cf + k * ci;
      if (cyc > max_cycle || cyc < -max_cycle) { rc = -9; break; }
      int32_t cx = ctx[k]; /* nk == len here: some qual >= 6 > lowQualityTail */
      size_t mi = (((size_t)cov * ORC_NQUAL + q) * ncyc + (size_t)(cyc + max_cycle)) * 17 + (size_t)(cx < 0 ? 16 : ((cx >> 4) & 15));
      int16_t v = memo[mi];
      if (v < 0) {
        /* the finalized-table object fills its own lazily allocated caches: one thread at a time (misses are rare) */
#pragma omp critical(orc_recal_qual)
        v = orc_bqsr_recal_qual(f, cov, q, cyc, cx, quantized, static_q);
        memo[mi] = v;
      }
      qo[k] = (uint8_t)v;
    }
    if (rc) break;
  }
  free(sbuf); free(ctx); free(a.cigar);
  return rc;
}

/* ---------- report text, filters/print-bqsr.go ---------- */
typedef struct { char *p; size_t n, cap; } sbuf_t;
static void sb_printf(sbuf_t *s, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#include <stdarg.h>
static void sb_printf(sbuf_t *s, const char *fmt, ...) {
  va_list ap;
  for (;;) {
    va_start(ap, fmt);
    int w = vsnprintf(s->p + s->n, s->cap - s->n, fmt, ap);
    va_end(ap);
    if ((size_t)w < s->cap - s->n) { s->n += (size_t)w; return; }
    s->cap = s->cap * 2 + (size_t)w + 64;
    s->p = (char *)realloc(s->p, s->cap);
  }
}
static int ilen(int64_t v) { char t[32]; return snprintf(t, sizeof t, "%lld", (long long)v); }
static int imax(int a, int b) { return a > b ? a : b; }

static const char *ctx_key_to_string(int idx, char out[3]) { /* keyToString :163-175 for 2-mers; idx = (key>>4)&15 */
  static const char bases[] = "ACGT";
  out[0] = bases[idx & 3]; out[1] = bases[(idx >> 2) & 3]; out[2] = 0;
  return out;
}

typedef struct { int cov, qual; int is_cycle; char text[16]; int64_t obs, mism; uint8_t emp; } row2;
static const char *const *g_names;
static int cmp_cov(const void *a, const void *b) { return strcmp(g_names[*(const int *)a], g_names[*(const int *)b]); }
static int cmp_row2(const void *x, const void *y) {
  const row2 *a = (const row2 *)x, *b = (const row2 *)y;
  int c = strcmp(g_names[a->cov], g_names[b->cov]);
  if (c) return c;
  if (a->qual != b->qual) return a->qual < b->qual ? -1 : 1;
  return strcmp(a->text, b->text);
}

char *orc_bqsr_report(const orc_bqsr_final *f, const char *const *cov_names, const char *prefix) {
  sbuf_t s = {0, 0, 0};
  s.cap = 1 << 16; s.p = (char *)malloc(s.cap);
  g_names = cov_names;
  /* PrintBQSRTables :269-298 */
  sb_printf(&s, "#:%sReport.v1.1:5\n", prefix);
  sb_printf(&s, "#:%sTable:2:17:%%s:%%s:;\n", prefix);
  sb_printf(&s, "#:%sTable:Arguments:Recalibration argument collection values used in this run\n", prefix);
  static const char *args[] = {
    "Argument                    Value                                                                   ",
    "binary_tag_name             null                                                                    ",
    "covariate                   ReadGroupCovariate,QualityScoreCovariate,ContextCovariate,CycleCovariate",
    "default_platform            null                                                                    ",
    "deletions_default_quality   45                                                                      ",
    "force_platform              null                                                                    ",
    "indels_context_size         3                                                                       ",
    "insertions_default_quality  45                                                                      ",
    "low_quality_tail            2                                                                       ",
    "maximum_cycle_value         500                                                                     ",
    "mismatches_context_size     2                                                                       ",
    "mismatches_default_quality  -1                                                                      ",
    "no_standard_covs            false                                                                   ",
    "quantizing_levels           16                                                                      ",
    "recalibration_report        null                                                                    ",
    "run_without_dbsnp           false                                                                   ",
    "solid_nocall_strategy       THROW_EXCEPTION                                                         ",
    "solid_recal_mode            SET_Q_ZERO                                                              "};
  for (size_t i = 0; i < sizeof args / sizeof *args; i++) sb_printf(&s, "%s\n", args[i]);
  sb_printf(&s, "\n");
  /* printQuantizationTable :49-76 (always 16 levels, :33) */
  {
    int64_t counts[94]; uint8_t scores[94];
    orc_bqsr_quantize(f, 16, counts, scores);
    sb_printf(&s, "#:%sTable:3:%d:%%d:%%d:%%d:;\n", prefix, 94);
    sb_printf(&s, "#:%sTable:Quantized:Quality quantization map\n", prefix);
    int w1 = (int)strlen("QualityScore"), w2 = (int)strlen("Count"), w3 = (int)strlen("QuantizedScore");
    for (int i = 0; i < 94; i++) { w1 = imax(w1, ilen(i)); w2 = imax(w2, ilen(counts[i])); w3 = imax(w3, ilen(scores[i])); }
    sb_printf(&s, "%-*s  %-*s  %-*s\n", w1, "QualityScore", w2, "Count", w3, "QuantizedScore");
    for (int i = 0; i < 94; i++) sb_printf(&s, "%*d  %*lld  %*d\n", w1, i, w2, (long long)counts[i], w3, scores[i]);
    sb_printf(&s, "\n");
  }
  /* printCombinedBQSRTable :78-124 */
  {
    int nrg = 0;
    int *order = (int *)malloc((size_t)f->n_cov * sizeof(int));
    for (int c = 0; c < f->n_cov; c++) if (f->c_present[c]) order[nrg++] = c;
    sb_printf(&s, "#:%sTable:6:%d:%%s:%%s:%%.4f:%%.4f:%%d:%%.2f:;\n", prefix, nrg);
    sb_printf(&s, "#:%sTable:RecalTable0:\n", prefix);
    int wrg = (int)strlen("ReadGroup"), wev = (int)strlen("EventType"), wemp = (int)strlen("EmpiricalQuality"),
        west = (int)strlen("EstimatedQReported"), wobs = (int)strlen("Observations"), werr = (int)strlen("Errors");
    char t[64];
    for (int k = 0; k < nrg; k++) {
      int c = order[k];
      wrg = imax(wrg, (int)strlen(cov_names[c]));
      wemp = imax(wemp, ilen(f->c_emp[c]) + 5);
      west = imax(west, snprintf(t, sizeof t, "%.4f", f->c_reported[c]));
      wobs = imax(wobs, ilen(f->c_obs[c]));
      werr = imax(werr, ilen(f->c_mism[c]) + 3);
    }
    sb_printf(&s, "%-*s  %-*s  %-*s  %-*s  %-*s  %-*s\n", wrg, "ReadGroup", wev, "EventType", wemp, "EmpiricalQuality", west,
              "EstimatedQReported", wobs, "Observations", werr, "Errors");
    qsort(order, (size_t)nrg, sizeof(int), cmp_cov);
    for (int k = 0; k < nrg; k++) {
      int c = order[k];
      sb_printf(&s, "%-*s  %-*s  %*d.0000  %*.4f  %*lld  %*lld.00\n", wrg, cov_names[c], wev, "M", wemp - 5, f->c_emp[c], west,
                f->c_reported[c], wobs, (long long)f->c_obs[c], werr - 3, (long long)f->c_mism[c]);
    }
    sb_printf(&s, "\n");
    free(order);
  }
  /* printBQSRTable :126-178 */
  {
    int nent = 0;
    for (size_t i = 0; i < (size_t)f->n_cov * ORC_NQUAL; i++) if (f->qual_tbl[2 * i] > 0) nent++;
    sb_printf(&s, "#:%sTable:6:%d:%%s:%%d:%%s:%%.4f:%%d:%%.2f:;\n", prefix, nent);
    sb_printf(&s, "#:%sTable:RecalTable1:\n", prefix);
    int wrg = (int)strlen("ReadGroup"), wq = (int)strlen("QualityScore"), wev = (int)strlen("EventType"),
        wemp = (int)strlen("EmpiricalQuality"), wobs = (int)strlen("Observations"), werr = (int)strlen("Errors");
    int *order = (int *)malloc((size_t)f->n_cov * sizeof(int));
    for (int c = 0; c < f->n_cov; c++) order[c] = c;
    qsort(order, (size_t)f->n_cov, sizeof(int), cmp_cov);
    for (int c = 0; c < f->n_cov; c++)
      for (int q = 0; q < ORC_NQUAL; q++) {
        size_t i = (size_t)c * ORC_NQUAL + q;
        if (f->qual_tbl[2 * i] <= 0) continue;
        wrg = imax(wrg, (int)strlen(cov_names[c])); wq = imax(wq, ilen(q)); wemp = imax(wemp, ilen(f->qual_emp[i]) + 5);
        wobs = imax(wobs, ilen(f->qual_tbl[2 * i])); werr = imax(werr, ilen(f->qual_tbl[2 * i + 1]) + 3);
      }
    sb_printf(&s, "%-*s  %-*s  %-*s  %-*s  %-*s  %-*s\n", wrg, "ReadGroup", wq, "QualityScore", wev, "EventType", wemp,
              "EmpiricalQuality", wobs, "Observations", werr, "Errors");
    for (int k = 0; k < f->n_cov; k++) {
      int c = order[k];
      for (int q = 0; q < ORC_NQUAL; q++) {
        size_t i = (size_t)c * ORC_NQUAL + q;
        if (f->qual_tbl[2 * i] <= 0) continue;
        sb_printf(&s, "%-*s  %*d  %-*s  %*d.0000  %*lld  %*lld.00\n", wrg, cov_names[c], wq, q, wev, "M", wemp - 5, f->qual_emp[i],
                  wobs, (long long)f->qual_tbl[2 * i], werr - 3, (long long)f->qual_tbl[2 * i + 1]);
      }
    }
    sb_printf(&s, "\n");
    free(order);
  }
  /* printOtherCovariateTable :186-266 */
  {
    size_t nrows = 0;
    for (size_t i = 0; i < (size_t)f->n_cov * ORC_NQUAL * f->ncyc; i++) if (f->cycle_tbl[2 * i] > 0) nrows++;
    for (size_t i = 0; i < (size_t)f->n_cov * ORC_NQUAL * ORC_NCTX; i++) if (f->ctx_tbl[2 * i] > 0) nrows++;
    row2 *rows = (row2 *)malloc((nrows + 1) * sizeof(row2));
    size_t r = 0;
    int wrg = (int)strlen("ReadGroup"), wq = (int)strlen("QualityScore"), wcv = (int)strlen("CovariateValue"),
        wcn = (int)strlen("CovariateName"), wev = (int)strlen("EventType"), wemp = (int)strlen("EmpiricalQuality"),
        wobs = (int)strlen("Observations"), werr = (int)strlen("Errors");
    for (int c = 0; c < f->n_cov; c++)
      for (int q = 0; q < ORC_NQUAL; q++) {
        for (int cy = 0; cy < f->ncyc; cy++) {
          size_t i = ((size_t)c * ORC_NQUAL + q) * f->ncyc + cy;
          if (f->cycle_tbl[2 * i] <= 0) continue;
          row2 *w = &rows[r++];
          w->cov = c; w->qual = q; w->is_cycle = 1; snprintf(w->text, sizeof w->text, "%d", cy - f->max_cycle);
          w->obs = f->cycle_tbl[2 * i]; w->mism = f->cycle_tbl[2 * i + 1]; w->emp = f->cycle_emp[i];
        }
        for (int x = 0; x < ORC_NCTX; x++) {
          size_t i = ((size_t)c * ORC_NQUAL + q) * ORC_NCTX + x;
          if (f->ctx_tbl[2 * i] <= 0) continue;
          row2 *w = &rows[r++];
          char t[3];
          w->cov = c; w->qual = q; w->is_cycle = 0; snprintf(w->text, sizeof w->text, "%s", ctx_key_to_string(x, t));
          w->obs = f->ctx_tbl[2 * i]; w->mism = f->ctx_tbl[2 * i + 1]; w->emp = f->ctx_emp[i];
        }
      }
    for (size_t k = 0; k < r; k++) {
      wrg = imax(wrg, (int)strlen(cov_names[rows[k].cov])); wq = imax(wq, ilen(rows[k].qual)); wcv = imax(wcv, (int)strlen(rows[k].text));
      wemp = imax(wemp, ilen(rows[k].emp) + 5); wobs = imax(wobs, ilen(rows[k].obs)); werr = imax(werr, ilen(rows[k].mism) + 3);
    }
    sb_printf(&s, "#:%sTable:8:%zu:%%s:%%d:%%s:%%s:%%s:%%.4f:%%d:%%.2f:;\n", prefix, r);
    sb_printf(&s, "#:%sTable:RecalTable2:\n", prefix);
    sb_printf(&s, "%-*s  %-*s  %-*s  %-*s  %-*s  %-*s  %-*s  %-*s\n", wrg, "ReadGroup", wq, "QualityScore", wcv, "CovariateValue", wcn,
              "CovariateName", wev, "EventType", wemp, "EmpiricalQuality", wobs, "Observations", werr, "Errors");
    /* sort.Slice by (ReadGroup, Qual, text) — keys are unique except a cycle and a context can never share a text */
    qsort(rows, r, sizeof(row2), cmp_row2);
    for (size_t k = 0; k < r; k++) {
      row2 *w = &rows[k];
      sb_printf(&s, "%-*s  %*d  %-*s  %-*s  %-*s  %*d.0000  %*lld  %*lld.00\n", wrg, cov_names[w->cov], wq, w->qual, wcv, w->text, wcn,
                w->is_cycle ? "Cycle" : "Context", wev, "M", wemp - 5, w->emp, wobs, (long long)w->obs, werr - 3, (long long)w->mism);
    }
    sb_printf(&s, "\n");
    free(rows);
  }
  return s.p;
}
void orc_free(void *p) { free(p); }
