/*
 * oracle/orc_sort.c — CPU oracle (test infrastructure; see orc.h): coordinate order.
 * Restates sam.CoordinateLess + modFlag (sam/sam-types.go:408-473) and the observable result of
 * By.ParallelStableSort (sam/sam-types.go:599-641): a permutation consistent with CoordinateLess in
 * which records that compare equal under all keys keep their input order.
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

/* sam/sam-types.go:408-421 */
uint16_t orc_mod_flag(uint16_t flag) {
  if ((flag & ORC_MULTIPLE) == 0) {
    flag &= (uint16_t)~ORC_NEXT_UNMAPPED;
    flag &= (uint16_t)~ORC_NEXT_REVERSED;
  }
  if (flag & ORC_UNMAPPED) flag &= (uint16_t)~ORC_REVERSED;
  if (flag & ORC_NEXT_UNMAPPED) flag &= (uint16_t)~ORC_NEXT_REVERSED;
  return flag;
}

/* Go string comparison: bytewise, shorter prefix is smaller */
static int qname_cmp(const orc_batch *b, uint64_t i, uint64_t j) {
  uint64_t li = b->qname_off[i + 1] - b->qname_off[i], lj = b->qname_off[j + 1] - b->qname_off[j];
  uint64_t m = li < lj ? li : lj;
  int c = memcmp(b->qname + b->qname_off[i], b->qname + b->qname_off[j], m);
  if (c) return c;
  return li < lj ? -1 : (li > lj ? 1 : 0);
}

/* sam/sam-types.go:425-473 */
int orc_coordinate_less(const orc_batch *b, uint64_t i, uint64_t j) {
  int32_t refid1 = b->refid[i], refid2 = b->refid[j];
  if (refid1 < refid2) return refid1 >= 0;
  if (refid2 < refid1) return refid2 < 0;
  if (b->pos[i] < b->pos[j]) return 1;
  if (b->pos[i] > b->pos[j]) return 0;
  int rev1 = (b->flag[i] & ORC_REVERSED) != 0, rev2 = (b->flag[j] & ORC_REVERSED) != 0;
  if (rev1 != rev2) return !rev1;
  uint64_t l1 = b->qname_off[i + 1] - b->qname_off[i], l2 = b->qname_off[j + 1] - b->qname_off[j];
  if (l1 != 0 && l2 != 0) {
    int c = qname_cmp(b, i, j);
    if (c < 0) return 1;
    if (c > 0) return 0;
  }
  uint16_t flag1 = orc_mod_flag(b->flag[i]), flag2 = orc_mod_flag(b->flag[j]);
  if (flag1 < flag2) return 1;
  if (flag1 > flag2) return 0;
  if (b->mapq[i] < b->mapq[j]) return 1;
  if (b->mapq[i] > b->mapq[j]) return 0;
  if ((b->flag[i] & ORC_MULTIPLE) && (b->flag[j] & ORC_MULTIPLE)) {
    int32_t n1 = b->next_refid[i], n2 = b->next_refid[j];
    if (n1 < n2) return 1; /* no special treatment of negative values (reference comment) */
    if (n1 > n2) return 0;
    if (b->pnext[i] < b->pnext[j]) return 1;
    if (b->pnext[i] > b->pnext[j]) return 0;
  }
  return b->tlen[i] < b->tlen[j];
}

/* top-down stable merge sort on the index permutation */
static void msort(const orc_batch *b, uint32_t *a, uint32_t *tmp, uint64_t lo, uint64_t hi) {
  if (hi - lo < 2) return;
  if (hi - lo <= 8) { /* stable insertion sort */
    for (uint64_t i = lo + 1; i < hi; i++) {
      uint32_t v = a[i];
      uint64_t k = i;
      while (k > lo && orc_coordinate_less(b, v, a[k - 1])) { a[k] = a[k - 1]; k--; }
      a[k] = v;
    }
    return;
  }
  uint64_t mid = lo + (hi - lo) / 2;
  msort(b, a, tmp, lo, mid);
  msort(b, a, tmp, mid, hi);
  if (!orc_coordinate_less(b, a[mid], a[mid - 1])) return; /* already ordered */
  uint64_t i = lo, j = mid, k = lo;
  while (i < mid && j < hi) {
    if (orc_coordinate_less(b, a[j], a[i])) tmp[k++] = a[j++]; /* take right only if strictly less: stable */
    else tmp[k++] = a[i++];
  }
  while (i < mid) tmp[k++] = a[i++];
  while (j < hi) tmp[k++] = a[j++];
  memcpy(a + lo, tmp + lo, (hi - lo) * sizeof(uint32_t));
}

/* records that survive RemoveOptionalReads (filters/simple-filters.go:146-152): the filter sits behind MarkDuplicates in
 * filters1 (cmd/filter.go:773,803), so a record with the sr tag is dropped before Slice(&alns) (sam/filter-pipeline.go:108-124)
 * collects it: it is never sorted, never seen by MarkOpticalDuplicates, Recalibrate or ApplyBQSR. */
uint64_t orc_num_sorted(const orc_batch *b) {
  uint64_t k = 0;
  for (uint64_t i = 0; i < b->n; i++)
    if (!(b->has_sr && b->has_sr[i])) k++;
  return k;
}

/* perm_out[0 .. orc_num_sorted) = the surviving records in coordinate order; the dropped (sr-tagged) ones follow in input order */
int orc_sort_coordinate(const orc_batch *b, uint32_t *perm_out) {
  uint64_t n = b->n, k = 0;
  if (n > 0xFFFFFFFFull) return -1;
  for (uint64_t i = 0; i < n; i++)
    if (!(b->has_sr && b->has_sr[i])) perm_out[k++] = (uint32_t)i;
  uint64_t n_out = k;
  for (uint64_t i = 0; i < n; i++)
    if (b->has_sr && b->has_sr[i]) perm_out[k++] = (uint32_t)i;
  if (n_out < 2) return 0;
  uint32_t *tmp = (uint32_t *)malloc(n_out * sizeof(uint32_t));
  if (!tmp) return -2;
  msort(b, perm_out, tmp, 0, n_out);
  free(tmp);
  return 0;
}

/* ---- the same with all host cores (bench.py's CPU baseline): chunks sorted independently, then merged pairwise; every merge is
 * split among the threads along the merge path (the output slot k is produced from (i, j), i + j = k, found by binary search), which
 * is what a parallel merge sort (pargo sort.StableSort, sam/sam-types.go:639-641) does.  Same permutation as orc_sort_coordinate. */
#include <omp.h>
static void merge_seg(const orc_batch *b, const uint32_t *x, uint64_t nx, const uint32_t *y, uint64_t ny, uint32_t *out, uint64_t k0, uint64_t k1) {
  /* co-rank: i = number of elements of x among the first k0 outputs of the stable merge (x before y on ties) */
  uint64_t lo = k0 > ny ? k0 - ny : 0, hi = k0 < nx ? k0 : nx;
  while (lo < hi) {
    uint64_t i = lo + (hi - lo) / 2, j = k0 - i;
    /* too few from x if y[j-1] is strictly less than x[i] is false ... : x[i] must come before y[j-1] iff !(y[j-1] < x[i]) */
    if (j > 0 && i < nx && !orc_coordinate_less(b, y[j - 1], x[i])) lo = i + 1;
    else hi = i;
  }
  uint64_t i = lo, j = k0 - lo;
  for (uint64_t k = k0; k < k1; k++) {
    if (i < nx && (j >= ny || !orc_coordinate_less(b, y[j], x[i]))) out[k] = x[i++];
    else out[k] = y[j++];
  }
}
int orc_sort_coordinate_mt(const orc_batch *b, uint32_t *perm_out, int n_threads) {
  uint64_t n = b->n, k = 0;
  if (n > 0xFFFFFFFFull) return -1;
  if (n_threads < 1) n_threads = omp_get_max_threads();
  for (uint64_t i = 0; i < n; i++)
    if (!(b->has_sr && b->has_sr[i])) perm_out[k++] = (uint32_t)i;
  uint64_t n_out = k;
  for (uint64_t i = 0; i < n; i++)
    if (b->has_sr && b->has_sr[i]) perm_out[k++] = (uint32_t)i;
  if (n_out < 2) return 0;
  uint32_t *tmp = (uint32_t *)malloc(n_out * sizeof(uint32_t));
  if (!tmp) return -2;
  uint64_t chunks = 1;
  while (chunks < (uint64_t)n_threads * 4 && n_out / (chunks * 2) >= 4096) chunks *= 2;
  uint64_t clen = (n_out + chunks - 1) / chunks;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (uint64_t c = 0; c < chunks; c++) {
    uint64_t lo = c * clen, hi = lo + clen < n_out ? lo + clen : n_out;
    if (lo < hi) msort(b, perm_out, tmp, lo, hi);
  }
  uint32_t *src = perm_out, *dst = tmp;
  for (uint64_t w = clen; w < n_out; w *= 2) {
    uint64_t pairs = (n_out + 2 * w - 1) / (2 * w);
    uint64_t seg = 1 << 16;  /* output slots per task */
#pragma omp parallel for schedule(dynamic, 4) num_threads(n_threads)
    for (uint64_t t = 0; t < pairs * ((2 * w + seg - 1) / seg); t++) {
      uint64_t per = (2 * w + seg - 1) / seg, p = t / per, s0 = (t % per) * seg;
      uint64_t lo = p * 2 * w, mid = lo + w < n_out ? lo + w : n_out, hi = lo + 2 * w < n_out ? lo + 2 * w : n_out;
      uint64_t len = hi - lo;
      if (s0 >= len) continue;
      uint64_t s1 = s0 + seg < len ? s0 + seg : len;
      merge_seg(b, src + lo, mid - lo, src + mid, hi - mid, dst + lo, s0, s1);
    }
    uint32_t *t2 = src; src = dst; dst = t2;
  }
  if (src != perm_out) memcpy(perm_out, src, n_out * sizeof(uint32_t));
  free(tmp);
  return 0;
}

/* sam/split-merge.go:178-213 computeContigGroups (group numbering only; "unmapped" is group 0) */
int orc_contig_groups(const int32_t *ref_len, int n_ref, int contig_group_size, int32_t *group_of_ref) {
  if (contig_group_size <= 0) {
    for (int i = 0; i < n_ref; i++)
      if (ref_len[i] > contig_group_size) contig_group_size = ref_len[i];
    if (contig_group_size <= 0) return -1;
  }
  int idx = 1;
  int64_t cur = 0;
  for (int i = 0; i < n_ref; i++) {
    int64_t ln = ref_len[i];
    if (cur > 0 && cur + ln > contig_group_size) { idx++; cur = 0; }
    group_of_ref[i] = idx;
    cur += ln;
  }
  return n_ref ? idx : 0;
}
