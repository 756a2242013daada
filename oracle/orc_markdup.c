/*
 * oracle/orc_markdup.c — CPU oracle (test infrastructure; see orc.h): duplicate marking and
 * optical-duplicate / DuplicationMetrics counting.
 *
 * Restates filters/mark-duplicates.go (entire) and filters/mark-optical-duplicates.go:50-93,176-525,
 * filters/graph.go, filters/unpedantic.go:32-34 as ONE sequential execution: records enter the
 * MarkDuplicates filter in input order (the pargo sync.Map / CAS tournament then degenerates to the
 * plain loops below); MarkOpticalDuplicates walks the records in sorted order, left to right.
 */
#include "orc.h"
#include <stdlib.h>
#include <string.h>

/* ---------- small open-addressing hash map: fixed-size byte keys -> int64 value ---------- */
typedef struct {
  uint8_t *keys; int64_t *vals; uint8_t *used;
  size_t ksz; uint64_t cap, cnt;
} fmap;

static uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
  return x;
}
static uint64_t hash_bytes(const uint8_t *p, size_t n) {
  uint64_t h = 0x9e3779b97f4a7c15ull;
  for (size_t i = 0; i < n; i++) h = mix64(h ^ p[i]);
  return h;
}
static int fmap_init(fmap *m, size_t ksz, uint64_t expect) {
  uint64_t cap = 64;
  while (cap < expect * 2 + 16) cap <<= 1;
  m->ksz = ksz; m->cap = cap; m->cnt = 0;
  m->keys = (uint8_t *)malloc(cap * ksz);
  m->vals = (int64_t *)malloc(cap * sizeof(int64_t));
  m->used = (uint8_t *)calloc(cap, 1);
  return (m->keys && m->vals && m->used) ? 0 : -1;
}
static void fmap_free(fmap *m) { free(m->keys); free(m->vals); free(m->used); }
/* returns pointer to the value slot; *found tells whether the key was present (if absent it is inserted with val init) */
static int64_t *fmap_get(fmap *m, const void *key, int insert, int64_t init, int *found) {
  uint64_t h = hash_bytes((const uint8_t *)key, m->ksz) & (m->cap - 1);
  for (;;) {
    if (!m->used[h]) {
      *found = 0;
      if (!insert) return NULL;
      m->used[h] = 1;
      memcpy(m->keys + h * m->ksz, key, m->ksz);
      m->vals[h] = init;
      m->cnt++;
      return &m->vals[h];
    }
    if (memcmp(m->keys + h * m->ksz, key, m->ksz) == 0) { *found = 1; return &m->vals[h]; }
    h = (h + 1) & (m->cap - 1);
  }
}

/* ---------- (library, QNAME) map: stores a representative record index, compares QNAME bytes ---------- */
typedef struct { int64_t *rep; int64_t *val; uint64_t cap; } qmap;
static int qmap_init(qmap *m, uint64_t expect) {
  uint64_t cap = 64;
  while (cap < expect * 2 + 16) cap <<= 1;
  m->cap = cap;
  m->rep = (int64_t *)malloc(cap * sizeof(int64_t));
  m->val = (int64_t *)malloc(cap * sizeof(int64_t));
  if (!m->rep || !m->val) return -1;
  for (uint64_t i = 0; i < cap; i++) m->rep[i] = -1;
  return 0;
}
static void qmap_free(qmap *m) { free(m->rep); free(m->val); }
static int qname_eq(const orc_batch *b, uint64_t i, uint64_t j) {
  uint64_t li = b->qname_off[i + 1] - b->qname_off[i], lj = b->qname_off[j + 1] - b->qname_off[j];
  return li == lj && memcmp(b->qname + b->qname_off[i], b->qname + b->qname_off[j], li) == 0;
}
static int qname_cmp(const orc_batch *b, uint64_t i, uint64_t j) {
  uint64_t li = b->qname_off[i + 1] - b->qname_off[i], lj = b->qname_off[j + 1] - b->qname_off[j];
  uint64_t m = li < lj ? li : lj;
  int c = memcmp(b->qname + b->qname_off[i], b->qname + b->qname_off[j], m);
  if (c) return c;
  return li < lj ? -1 : (li > lj ? 1 : 0);
}
/* pargo sync.Map.DeleteOrStore({lb, qname}, aln): if an entry is stored, remove and return it (>=0);
 * otherwise store `rec` and return -1.  mark-duplicates.go:336-340 */
static int64_t qmap_delete_or_store(qmap *m, const orc_batch *b, uint16_t lb, const uint16_t *lib_of, uint64_t rec) {
  uint64_t h = hash_bytes(b->qname + b->qname_off[rec], b->qname_off[rec + 1] - b->qname_off[rec]);
  h = mix64(h ^ lb) & (m->cap - 1);
  for (;;) {
    if (m->rep[h] < 0) { m->rep[h] = (int64_t)rec; m->val[h] = (int64_t)rec; return -1; }
    uint64_t r = (uint64_t)m->rep[h];
    if (lib_of[r] == lb && qname_eq(b, r, rec)) {
      if (m->val[h] >= 0) { int64_t old = m->val[h]; m->val[h] = -1; return old; }
      m->val[h] = (int64_t)rec;
      return -1;
    }
    h = (h + 1) & (m->cap - 1);
  }
}

/* ---------- filters/mark-duplicates.go:36-68 ---------- */
int32_t orc_phred_score(const uint8_t *qual, uint32_t n, int *invalid) {
  int32_t score = 0;
  int err = 0;
  for (uint32_t i = 0; i < n; i++) {
    uint8_t c = qual[i];
    if (c > 126 - 33) err = 1;       /* phredScoreTable: error flag */
    else if (c >= 15) score += c;    /* qualities below 15 contribute 0 */
  }
  if (invalid) *invalid = err;
  return score;
}

static const char cigar_ops[] = "MIDNSHP=X";
static int op_clipped(char p) { return p == 'S' || p == 'H'; }
static int op_reference(char p) { return p == 'M' || p == 'D' || p == 'N' || p == '=' || p == 'X'; }

/* filters/mark-duplicates.go:79-110 */
int32_t orc_unclipped_position(int32_t pos, uint16_t flag, const uint32_t *cigar, uint32_t n_cigar) {
  int32_t result = pos;
  if (n_cigar == 0) return result;
  if (flag & ORC_REVERSED) {
    int32_t clipped = 1;
    result--;
    for (int64_t i = (int64_t)n_cigar - 1; i >= 0; i--) {
      char p = cigar_ops[cigar[i] & 0xF];
      int32_t len = (int32_t)(cigar[i] >> 4);
      int32_t c = op_clipped(p), r = op_reference(p);
      clipped *= c;
      result += (r | clipped) * len;
    }
  } else {
    for (uint32_t i = 0; i < n_cigar; i++) {
      char p = cigar_ops[cigar[i] & 0xF];
      if (!op_clipped(p)) break;
      result -= (int32_t)(cigar[i] >> 4);
    }
  }
  return result;
}

/* :177-184 */
static int is_true_fragment(uint16_t f) { return (f & (ORC_MULTIPLE | ORC_NEXT_UNMAPPED)) != ORC_MULTIPLE; }
static int is_true_pair(uint16_t f) { return (f & (ORC_MULTIPLE | ORC_NEXT_UNMAPPED)) == ORC_MULTIPLE; }

typedef struct { uint16_t lb; uint8_t reversed; uint8_t pad; int32_t refid; int32_t pos; } frag_key;   /* :188-193 */
typedef struct { uint16_t lb; uint8_t rev1, rev2; int32_t refid1, refid2; int64_t pos; } pair_key;       /* :272-277 */
typedef struct { int32_t score; int64_t aln1, aln2; int64_t opt_head; } pair_rec;                         /* :286-290 */
typedef struct { int64_t aln; int64_t next; } aln_cons;

typedef struct {
  const orc_batch *b;
  uint16_t *flag;       /* working FLAG column */
  uint16_t *lib_of;     /* LIBID per record (ORC_NIL16 = nil) */
  int32_t *upos, *score;
  fmap fragments, pairs;
  qmap pair_frags;
  pair_rec *prec; uint64_t n_prec;
  aln_cons *cons; uint64_t n_cons;
} md_state;

static void md_free(md_state *s) {
  free(s->lib_of); free(s->upos); free(s->score);
  fmap_free(&s->fragments); fmap_free(&s->pairs); qmap_free(&s->pair_frags);
  free(s->prec); free(s->cons);
}

/* :210-254 */
static void classify_fragment(md_state *s, uint64_t aln) {
  frag_key k;
  memset(&k, 0, sizeof k);
  k.lb = s->lib_of[aln]; k.refid = s->b->refid[aln]; k.pos = s->upos[aln];
  k.reversed = (s->flag[aln] & ORC_REVERSED) != 0;
  int found;
  int64_t *best = fmap_get(&s->fragments, &k, 1, (int64_t)aln, &found);
  if (!found) return;
  uint64_t best_aln = (uint64_t)*best;
  if (is_true_fragment(s->flag[aln])) {
    int32_t aln_score = s->score[aln];
    if (is_true_pair(s->flag[best_aln])) {
      s->flag[aln] |= ORC_DUPLICATE;
    } else if (s->score[best_aln] > aln_score) {
      s->flag[aln] |= ORC_DUPLICATE;
    } else if (s->score[best_aln] == aln_score) {
      if (qname_cmp(s->b, aln, best_aln) > 0) {
        s->flag[aln] |= ORC_DUPLICATE;
      } else { /* CAS succeeds in a sequential run */
        *best = (int64_t)aln;
        s->flag[best_aln] |= ORC_DUPLICATE;
      }
    } else {
      *best = (int64_t)aln;
      s->flag[best_aln] |= ORC_DUPLICATE;
    }
  } else {
    if (!is_true_pair(s->flag[best_aln])) {
      *best = (int64_t)aln;
      s->flag[best_aln] |= ORC_DUPLICATE;
    }
  }
}

/* orders the two ends (:347-353) and fills the pair key (:354-361) */
static void order_pair(const md_state *s, uint64_t *aln1, uint64_t *aln2, pair_key *k) {
  const orc_batch *b = s->b;
  int32_t r1 = b->refid[*aln1], r2 = b->refid[*aln2];
  int32_t p1 = s->upos[*aln1], p2 = s->upos[*aln2];
  int rev1 = (s->flag[*aln1] & ORC_REVERSED) != 0, rev2 = (s->flag[*aln2] & ORC_REVERSED) != 0;
  if (r1 > r2 || (r1 == r2 && (p1 > p2 || (p1 == p2 && rev1 && !rev2)))) {
    uint64_t t = *aln1; *aln1 = *aln2; *aln2 = t;
    int32_t ti = r1; r1 = r2; r2 = ti;
    ti = p1; p1 = p2; p2 = ti;
  }
  memset(k, 0, sizeof *k);
  k->lb = s->lib_of[*aln1];
  k->refid1 = r1; k->refid2 = r2;
  k->pos = (int64_t)((uint64_t)(int64_t)p1 << 32) + (int64_t)p2;
  k->rev1 = (s->flag[*aln1] & ORC_REVERSED) != 0;
  k->rev2 = (s->flag[*aln2] & ORC_REVERSED) != 0;
}

/* :342-396: the pair (aln1 = the mate that arrived later, aln2) enters the pair tournament */
static void classify_pair_of(md_state *s, uint64_t aln1, uint64_t aln2);

/* :329-396 */
static void classify_pair(md_state *s, uint64_t aln) {
  if (!is_true_pair(s->flag[aln])) return;
  int64_t e = qmap_delete_or_store(&s->pair_frags, s->b, s->lib_of[aln], s->lib_of, aln);
  if (e < 0) return;
  classify_pair_of(s, aln, (uint64_t)e);
}

static void classify_pair_of(md_state *s, uint64_t aln1, uint64_t aln2) {
  int32_t score = s->score[aln1] + s->score[aln2];
  pair_key k;
  order_pair(s, &aln1, &aln2, &k);
  int found;
  int64_t *slot = fmap_get(&s->pairs, &k, 1, (int64_t)s->n_prec, &found);
  if (!found) {
    pair_rec *p = &s->prec[s->n_prec++];
    p->score = score; p->aln1 = (int64_t)aln1; p->aln2 = (int64_t)aln2; p->opt_head = -1;
    return;
  }
  pair_rec *best = &s->prec[*slot];
  if (best->score > score) {
    s->flag[aln1] |= ORC_DUPLICATE; s->flag[aln2] |= ORC_DUPLICATE;
  } else if (best->score == score) {
    if (qname_cmp(s->b, aln1, (uint64_t)best->aln1) > 0) {
      s->flag[aln1] |= ORC_DUPLICATE; s->flag[aln2] |= ORC_DUPLICATE;
    } else {
      s->flag[best->aln1] |= ORC_DUPLICATE; s->flag[best->aln2] |= ORC_DUPLICATE;
      best->score = score; best->aln1 = (int64_t)aln1; best->aln2 = (int64_t)aln2; /* opticalDuplicates of the new pair is empty */
      best->opt_head = -1;
    }
  } else {
    s->flag[best->aln1] |= ORC_DUPLICATE; s->flag[best->aln2] |= ORC_DUPLICATE;
    best->score = score; best->aln1 = (int64_t)aln1; best->aln2 = (int64_t)aln2;
    best->opt_head = -1;
  }
}

/* MarkDuplicates closure, :398-445.  Both variants (alsoOpticals or not) flag the same reads; they differ only in which
 * reads carry a LIBID afterwards, and this oracle derives LIBID from rgid on demand. */
static int md_run(md_state *s, const orc_batch *b, const orc_header *h, uint16_t *flag_out) {
  uint64_t n = b->n;
  memset(s, 0, sizeof *s);
  s->b = b; s->flag = flag_out;
  s->lib_of = (uint16_t *)malloc((n + 1) * sizeof(uint16_t));
  s->upos = (int32_t *)calloc(n + 1, sizeof(int32_t));
  s->score = (int32_t *)calloc(n + 1, sizeof(int32_t));
  s->prec = (pair_rec *)malloc((n / 2 + 1) * sizeof(pair_rec));
  s->cons = (aln_cons *)malloc((n / 2 + 1) * sizeof(aln_cons));
  if (!s->lib_of || !s->upos || !s->score || !s->prec || !s->cons) return -2;
  if (fmap_init(&s->fragments, sizeof(frag_key), n) || fmap_init(&s->pairs, sizeof(pair_key), n / 2 + 1) ||
      qmap_init(&s->pair_frags, n))
    return -2;
  for (uint64_t i = 0; i < n; i++) {
    flag_out[i] = b->flag[i];
    uint16_t rg = b->rgid[i];
    s->lib_of[i] = (rg != ORC_NIL16 && (int32_t)rg < h->n_rg) ? h->rg_lib[rg] : ORC_NIL16; /* addLIBID :142-150 */
  }
  for (uint64_t i = 0; i < n; i++) {
    if ((flag_out[i] & (ORC_UNMAPPED | ORC_SECONDARY | ORC_SUPPLEMENTARY)) != 0) continue;
    int invalid;
    const uint32_t *cg = b->cigar + b->cigar_off[i];
    s->upos[i] = orc_unclipped_position(b->pos[i], flag_out[i], cg, (uint32_t)(b->cigar_off[i + 1] - b->cigar_off[i]));
    s->score[i] = orc_phred_score(b->qual + b->qual_off[i], (uint32_t)(b->qual_off[i + 1] - b->qual_off[i]), &invalid);
    if (invalid) return -3; /* reference: log.Panic("Invalid QUAL character") */
    classify_fragment(s, i);
    classify_pair(s, i);
  }
  return 0;
}

int orc_mark_duplicates(const orc_batch *b, const orc_header *h, uint16_t *flag_out, int32_t *upos_out, int32_t *score_out) {
  md_state s;
  int rc = md_run(&s, b, h, flag_out);
  if (rc == 0) {
    if (upos_out) memcpy(upos_out, s.upos, b->n * sizeof(int32_t));
    if (score_out) memcpy(score_out, s.score, b->n * sizeof(int32_t));
  }
  md_free(&s);
  return rc;
}

/* ---------- optical duplicates ---------- */

/* Go strconv.ParseInt(s, 10, 64) for the subset that does not error; returns 0 on syntax error via *ok */
static int64_t parse_int(const uint8_t *p, uint32_t n, int *ok) {
  *ok = 0;
  if (n == 0) return 0;
  uint32_t i = 0;
  int neg = 0;
  if (p[0] == '+' || p[0] == '-') { neg = p[0] == '-'; i = 1; }
  if (i >= n) return 0;
  int64_t v = 0;
  for (; i < n; i++) {
    if (p[i] < '0' || p[i] > '9') return 0;
    v = v * 10 + (p[i] - '0');
  }
  *ok = 1;
  return neg ? -v : v;
}

/* filters/mark-optical-duplicates.go:50-71 */
void orc_tile_info(const uint8_t *qname, uint32_t len, int64_t *t, int64_t *x, int64_t *y) {
  uint32_t start[9], end[9];
  int ncol = 0;
  uint32_t s = 0;
  for (uint32_t i = 0; i <= len; i++) {
    if (i == len || qname[i] == ':') {
      if (ncol < 9) { start[ncol] = s; end[ncol] = i; }
      ncol++;
      s = i + 1;
    }
  }
  int a;
  if (ncol == 7) a = 4;
  else if (ncol == 5) a = 2;
  else { *t = *x = *y = -1; return; }
  int ok1, ok2, ok3;
  *t = parse_int(qname + start[a], end[a] - start[a], &ok1);
  *x = parse_int(qname + start[a + 1], end[a + 1] - start[a + 1], &ok2);
  *y = parse_int(qname + start[a + 2], end[a + 2] - start[a + 2], &ok3);
  if (!ok1 || !ok2 || !ok3) { *t = *x = *y = -1; } /* reference panics here (internal.ParseInt); treated as "no tile info" */
}

typedef struct { int64_t t, x, y; uint16_t rg; } tinfo;

static int64_t abs64(int64_t v) { return v < 0 ? -v : v; }
/* filters/unpedantic.go:32-34 */
static int optical_short(const tinfo *a, const tinfo *b, int dist) {
  return abs64(a->x - b->x) <= dist && abs64(a->y - b->y) <= dist;
}
/* :82-93 */
static int is_optical(const tinfo *a, const tinfo *b, int dist) {
  if (a->rg != b->rg) return 0;
  if (a->t == -1 || b->t == -1) return 0;
  if (a->t != b->t) return 0;
  return optical_short(a, b, dist);
}

/* filters/graph.go:47-85 */
static int find_rep(int *g, int node) {
  int rep = node;
  while (rep != g[rep]) rep = g[rep];
  while (node != rep) { int nx = g[node]; g[node] = rep; node = nx; }
  return rep;
}

/* :232-273 countOpticalDuplicatesWithGraph: edges between members of the same (rg, tile != -1) group within distance;
 * result = sum over clusters of (size - 1) */
static int count_with_graph(const tinfo *d, int n, int dist) {
  int *g = (int *)malloc(n * sizeof(int));
  for (int i = 0; i < n; i++) g[i] = i;
  for (int i = 0; i < n; i++) {
    if (d[i].t == -1) continue;
    for (int j = i + 1; j < n; j++) {
      if (d[j].t == -1 || d[j].t != d[i].t || d[j].rg != d[i].rg) continue;
      if (optical_short(&d[i], &d[j], dist)) {
        int r1 = find_rep(g, j), r2 = find_rep(g, i);
        if (r1 != r2) g[r1] = r2;
      }
    }
  }
  int clusters = 0;
  for (int i = 0; i < n; i++)
    if (find_rep(g, i) == i) clusters++;
  free(g);
  return n - clusters;
}

/* :327-368 */
static int count_from_slice(const tinfo *d, int n, int dist) {
  if (n > 300000) return 0;
  if (n >= 4) return count_with_graph(d, n, dist);
  if (n < 2) return 0;
  int ctr = 0;
  if (is_optical(&d[0], &d[1], dist)) ctr++;
  if (n < 3) return ctr;
  if (is_optical(&d[0], &d[2], dist)) ctr++;
  if (ctr == 2) return 2;
  if (is_optical(&d[1], &d[2], dist)) return ctr + 1;
  return ctr;
}

static void get_tinfo(const orc_batch *b, uint64_t rec, tinfo *t) {
  orc_tile_info(b->qname + b->qname_off[rec], (uint32_t)(b->qname_off[rec + 1] - b->qname_off[rec]), &t->t, &t->x, &t->y);
  t->rg = b->rgid[rec];
}

int orc_dup_metrics(const orc_batch *b, const orc_header *h, const uint32_t *perm, int pixel_dist, uint16_t *flag_out,
                    int64_t *counters, int64_t *hist, int hist_len) {
  md_state s;
  uint64_t n = b->n;
  int nl = h->n_lib + 1; /* last row: "Unknown Library" (:437) */
  int rc = md_run(&s, b, h, flag_out);
  if (rc) { md_free(&s); return rc; }
  memset(counters, 0, (size_t)nl * ORC_NCTR * sizeof(int64_t));
  if (hist) memset(hist, 0, (size_t)nl * 3 * hist_len * sizeof(int64_t));
  qmap pf;
  if (qmap_init(&pf, n)) { md_free(&s); return -2; }
  /* MarkOpticalDuplicates :469-502, sequential left-to-right over reads.Alignments = the sorted records.  Records with the sr
   * tag are not among them: RemoveOptionalReads (filters/simple-filters.go:146-152) follows the mark-duplicates filter in
   * filters1 (cmd/filter.go:773,803), so they took part in the tournaments above and were then dropped. */
  for (uint64_t kk = 0; kk < n; kk++) {
    uint64_t aln = perm ? perm[kk] : kk;
    if (b->has_sr && b->has_sr[aln]) continue;
    uint16_t f = flag_out[aln];
    int lib = s.lib_of[aln] == ORC_NIL16 ? h->n_lib : s.lib_of[aln];
    int64_t *ctr = counters + (size_t)lib * ORC_NCTR;
    if (f & ORC_UNMAPPED) { ctr[3]++; continue; }
    if (f & (ORC_SECONDARY | ORC_SUPPLEMENTARY)) { ctr[2]++; continue; }
    if (is_true_fragment(f)) ctr[0]++;
    if (is_true_pair(f)) ctr[1]++;
    if (f & ORC_DUPLICATE) {
      if (is_true_fragment(f)) ctr[4]++;            /* markOpticalDuplicatesFragment :176-180 */
      if (is_true_pair(f)) {                        /* markOpticalDuplicatesPair :182-224 */
        uint64_t aln1 = aln, aln2;
        int64_t e = qmap_delete_or_store(&pf, b, s.lib_of[aln], s.lib_of, aln);
        if (e < 0) continue;
        aln2 = (uint64_t)e;
        ctr[5]++;
        pair_key k;
        order_pair(&s, &aln1, &aln2, &k);
        int found;
        int64_t *slot = fmap_get(&s.pairs, &k, 0, 0, &found);
        if (!found) { qmap_free(&pf); md_free(&s); return -4; } /* reference: log.Panicf("origin for duplicate read pair ... unknown") */
        pair_rec *best = &s.prec[*slot];
        if ((uint64_t)best->aln1 != aln1) {
          aln_cons *c = &s.cons[s.n_cons];
          c->aln = (int64_t)((flag_out[aln1] & ORC_FIRST) ? aln1 : aln2);
          c->next = best->opt_head;
          best->opt_head = (int64_t)s.n_cons++;
        }
      }
    }
  }
  qmap_free(&pf);
  for (int l = 0; l < nl; l++) counters[(size_t)l * ORC_NCTR + 1] /= 2; /* :504-506 */
  /* countOpticalDuplicatesPairs :370-431 + countOpticalDuplicates :275-325 */
  tinfo *fw = NULL, *rv = NULL;
  size_t cap_f = 0, cap_r = 0;
  for (uint64_t p = 0; p < s.n_prec; p++) {
    pair_rec *origin = &s.prec[p];
    size_t nf = 0, nr = 0;
    uint64_t origin_aln = (flag_out[origin->aln1] & ORC_FIRST) ? (uint64_t)origin->aln1 : (uint64_t)origin->aln2;
    for (int pass = 0; pass < 2; pass++) { /* pass 0: count, pass 1: fill */
      size_t cf = 0, cr = 0;
      if (flag_out[origin_aln] & ORC_REVERSED) { if (pass) get_tinfo(b, origin_aln, &rv[cr]); cr++; }
      else { if (pass) get_tinfo(b, origin_aln, &fw[cf]); cf++; }
      for (int64_t e = origin->opt_head; e >= 0; e = s.cons[e].next) {
        uint64_t a = (uint64_t)s.cons[e].aln;
        if (flag_out[a] & ORC_REVERSED) { if (cr <= 300000) { if (pass) get_tinfo(b, a, &rv[cr]); cr++; } }
        else { if (cf <= 300000) { if (pass) get_tinfo(b, a, &fw[cf]); cf++; } }
      }
      if (!pass) {
        nf = cf; nr = cr;
        if (nf > cap_f) { cap_f = nf * 2; fw = (tinfo *)realloc(fw, cap_f * sizeof(tinfo)); }
        if (nr > cap_r) { cap_r = nr * 2; rv = (tinfo *)realloc(rv, cap_r * sizeof(tinfo)); }
      }
    }
    int fc = count_from_slice(fw, (int)nf, pixel_dist);
    int rc2 = count_from_slice(rv, (int)nr, pixel_dist);
    int opt = fc + rc2;
    int dupcount = (int)(nf + nr);
    int lib = s.lib_of[origin->aln1] == ORC_NIL16 ? h->n_lib : s.lib_of[origin->aln1];
    counters[(size_t)lib * ORC_NCTR + 6] += opt;
    if (hist) { /* incrementDuplicatesCountsHistograms :150-174 */
      int idx1 = dupcount, idx2 = 0, idx3 = 0;
      if (dupcount - opt > 0) idx2 = dupcount - opt;
      if (opt > 0) idx3 = opt + 1;
      int64_t *hl = hist + (size_t)lib * 3 * hist_len;
      int c1 = idx1 < hist_len ? idx1 : hist_len - 1;
      hl[c1] += 1;
      if (idx2 > 0) hl[hist_len + (idx2 < hist_len ? idx2 : hist_len - 1)] += 1;      /* nonOpticalDuplicatesCountHistogram */
      if (idx3 > 0) hl[2 * hist_len + (idx3 < hist_len ? idx3 : hist_len - 1)] += 1;  /* opticalDuplicatesCountHistogram */
    }
  }
  free(fw); free(rv);
  md_free(&s);
  return 0;
}

/* ---------- the same on all host cores (bench.py's CPU baseline) ----------
 * The reference runs the three tournaments on sharded maps (sync.NewMap(16 * GOMAXPROCS), mark-duplicates.go:407-410): a key lives
 * in exactly one shard.  Here every shard is owned by one thread at a time and sees its records in input order, so the result is
 * the sequential execution's (orc_dup_metrics), whatever the thread count:
 *   records -> fragment-key shards -> classifyFragment;  records -> {library, QNAME} shards -> DeleteOrStore toggling;
 *   completed pairs (in order of completion) -> pair-key shards -> pair tournament;  the metrics pass likewise. */
#include <omp.h>

typedef struct { uint64_t *idx; uint64_t *start; int n_shards; } shard_lists;

/* stable counting sort of the items [0, n) with shard_of[i] >= 0 into per-shard lists (ascending item order inside a shard) */
static int build_shards(uint64_t n, const int32_t *shard_of, int n_shards, int n_threads, shard_lists *out) {
  out->n_shards = n_shards;
  out->start = (uint64_t *)calloc((size_t)n_shards + 1, sizeof(uint64_t));
  uint64_t *cnt = (uint64_t *)calloc((size_t)n_threads * n_shards, sizeof(uint64_t));
  if (!out->start || !cnt) return -2;
#pragma omp parallel num_threads(n_threads)
  {
    int t = omp_get_thread_num();
    uint64_t lo = n * (uint64_t)t / (uint64_t)n_threads, hi = n * (uint64_t)(t + 1) / (uint64_t)n_threads;
    uint64_t *c = cnt + (size_t)t * n_shards;
    for (uint64_t i = lo; i < hi; i++)
      if (shard_of[i] >= 0) c[shard_of[i]]++;
  }
  uint64_t total = 0;
  for (int sh = 0; sh < n_shards; sh++) {
    out->start[sh] = total;
    for (int t = 0; t < n_threads; t++) { uint64_t c = cnt[(size_t)t * n_shards + sh]; cnt[(size_t)t * n_shards + sh] = total; total += c; }
  }
  out->start[n_shards] = total;
  out->idx = (uint64_t *)malloc((total + 1) * sizeof(uint64_t));
  if (!out->idx) return -2;
#pragma omp parallel num_threads(n_threads)
  {
    int t = omp_get_thread_num();
    uint64_t lo = n * (uint64_t)t / (uint64_t)n_threads, hi = n * (uint64_t)(t + 1) / (uint64_t)n_threads;
    uint64_t *c = cnt + (size_t)t * n_shards;
    for (uint64_t i = lo; i < hi; i++)
      if (shard_of[i] >= 0) out->idx[c[shard_of[i]]++] = i;
  }
  free(cnt);
  return 0;
}
static void free_shards(shard_lists *l) { free(l->idx); free(l->start); }

static uint64_t frag_key_hash(const md_state *s, uint64_t i) {
  frag_key k;
  memset(&k, 0, sizeof k);
  k.lb = s->lib_of[i]; k.refid = s->b->refid[i]; k.pos = s->upos[i]; k.reversed = (s->flag[i] & ORC_REVERSED) != 0;
  return hash_bytes((const uint8_t *)&k, sizeof k);
}
static uint64_t qname_key_hash(const md_state *s, uint64_t i) {
  return mix64(hash_bytes(s->b->qname + s->b->qname_off[i], s->b->qname_off[i + 1] - s->b->qname_off[i]) ^ s->lib_of[i]);
}

int orc_dup_metrics_mt(const orc_batch *b, const orc_header *h, const uint32_t *perm, int pixel_dist, uint16_t *flag_out, int64_t *counters,
                       int n_threads) {
  uint64_t n = b->n;
  if (n_threads < 1) n_threads = omp_get_max_threads();
  int n_shards = 16 * n_threads;
  int nl = h->n_lib + 1;
  md_state g; /* shared columns */
  memset(&g, 0, sizeof g);
  g.b = b; g.flag = flag_out;
  g.lib_of = (uint16_t *)malloc((n + 1) * sizeof(uint16_t));
  g.upos = (int32_t *)calloc(n + 1, sizeof(int32_t));
  g.score = (int32_t *)calloc(n + 1, sizeof(int32_t));
  int32_t *sh_f = (int32_t *)malloc((n + 1) * sizeof(int32_t)), *sh_q = (int32_t *)malloc((n + 1) * sizeof(int32_t)), *sh_p = (int32_t *)malloc((n + 1) * sizeof(int32_t));
  int64_t *pair_of = (int64_t *)malloc((n + 1) * sizeof(int64_t));
  if (!g.lib_of || !g.upos || !g.score || !sh_f || !sh_q || !sh_p || !pair_of) return -2;
  int bad = 0;
  /* adapt + shard ids (MarkDuplicates closure :425-440) */
#pragma omp parallel for schedule(static) num_threads(n_threads) reduction(|:bad)
  for (uint64_t i = 0; i < n; i++) {
    flag_out[i] = b->flag[i];
    uint16_t rg = b->rgid[i];
    g.lib_of[i] = (rg != ORC_NIL16 && (int32_t)rg < h->n_rg) ? h->rg_lib[rg] : ORC_NIL16;
    sh_f[i] = sh_q[i] = -1;
    pair_of[i] = -1;
    if ((flag_out[i] & (ORC_UNMAPPED | ORC_SECONDARY | ORC_SUPPLEMENTARY)) != 0) continue;
    int invalid;
    g.upos[i] = orc_unclipped_position(b->pos[i], flag_out[i], b->cigar + b->cigar_off[i], (uint32_t)(b->cigar_off[i + 1] - b->cigar_off[i]));
    g.score[i] = orc_phred_score(b->qual + b->qual_off[i], (uint32_t)(b->qual_off[i + 1] - b->qual_off[i]), &invalid);
    bad |= invalid;
    sh_f[i] = (int32_t)(frag_key_hash(&g, i) % (uint64_t)n_shards);
    if (is_true_pair(flag_out[i])) sh_q[i] = (int32_t)(qname_key_hash(&g, i) % (uint64_t)n_shards);
  }
  if (bad) return -3;
  shard_lists lf, lq, lp;
  if (build_shards(n, sh_f, n_shards, n_threads, &lf) || build_shards(n, sh_q, n_shards, n_threads, &lq)) return -2;
  /* fragments, and mate matching */
  int rc_all = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int sh = 0; sh < n_shards; sh++) {
    md_state s = g;
    uint64_t cnt = lf.start[sh + 1] - lf.start[sh];
    if (fmap_init(&s.fragments, sizeof(frag_key), cnt)) { rc_all = -2; continue; }
    for (uint64_t k = lf.start[sh]; k < lf.start[sh + 1]; k++) classify_fragment(&s, lf.idx[k]);
    fmap_free(&s.fragments);
    cnt = lq.start[sh + 1] - lq.start[sh];
    if (qmap_init(&s.pair_frags, cnt)) { rc_all = -2; continue; }
    for (uint64_t k = lq.start[sh]; k < lq.start[sh + 1]; k++) {
      uint64_t aln = lq.idx[k];
      int64_t e = qmap_delete_or_store(&s.pair_frags, b, g.lib_of[aln], g.lib_of, aln);
      if (e >= 0) pair_of[aln] = e;
    }
    qmap_free(&s.pair_frags);
  }
  if (rc_all) return rc_all;
  /* pairs -> pair-key shards, in order of completion */
#pragma omp parallel for schedule(static) num_threads(n_threads)
  for (uint64_t i = 0; i < n; i++) {
    sh_p[i] = -1;
    if (pair_of[i] < 0) continue;
    uint64_t a1 = i, a2 = (uint64_t)pair_of[i];
    pair_key k;
    order_pair(&g, &a1, &a2, &k);
    sh_p[i] = (int32_t)(hash_bytes((const uint8_t *)&k, sizeof k) % (uint64_t)n_shards);
  }
  if (build_shards(n, sh_p, n_shards, n_threads, &lp)) return -2;
  md_state *ps = (md_state *)calloc((size_t)n_shards, sizeof(md_state));
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int sh = 0; sh < n_shards; sh++) {
    md_state *s = &ps[sh];
    *s = g;
    uint64_t cnt = lp.start[sh + 1] - lp.start[sh];
    s->prec = (pair_rec *)malloc((cnt + 1) * sizeof(pair_rec));
    s->cons = (aln_cons *)malloc((cnt + 1) * sizeof(aln_cons));
    s->n_prec = s->n_cons = 0;
    if (!s->prec || !s->cons || fmap_init(&s->pairs, sizeof(pair_key), cnt)) { rc_all = -2; continue; }
    for (uint64_t k = lp.start[sh]; k < lp.start[sh + 1]; k++) classify_pair_of(s, lp.idx[k], (uint64_t)pair_of[lp.idx[k]]);
  }
  if (rc_all) return rc_all;
  if (!counters) { /* MarkDuplicates only (the phase-1 filter): no metrics pass */
    for (int sh = 0; sh < n_shards; sh++) { fmap_free(&ps[sh].pairs); free(ps[sh].prec); free(ps[sh].cons); }
    free(ps); free(sh_f); free(sh_q); free(sh_p); free(pair_of);
    free_shards(&lf); free_shards(&lq); free_shards(&lp);
    free(g.lib_of); free(g.upos); free(g.score);
    return 0;
  }
  /* MarkOpticalDuplicates :469-502: counters over the sorted reads (thread-private, summed: RangeReduce) */
  memset(counters, 0, (size_t)nl * ORC_NCTR * sizeof(int64_t));
#pragma omp parallel num_threads(n_threads)
  {
    int64_t *c = (int64_t *)calloc((size_t)nl * ORC_NCTR, sizeof(int64_t));
#pragma omp for schedule(static)
    for (uint64_t kk = 0; kk < n; kk++) {
      uint64_t aln = perm ? perm[kk] : kk;
      if (b->has_sr && b->has_sr[aln]) continue;
      uint16_t f = flag_out[aln];
      int lib = g.lib_of[aln] == ORC_NIL16 ? h->n_lib : g.lib_of[aln];
      int64_t *ctr = c + (size_t)lib * ORC_NCTR;
      if (f & ORC_UNMAPPED) { ctr[3]++; continue; }
      if (f & (ORC_SECONDARY | ORC_SUPPLEMENTARY)) { ctr[2]++; continue; }
      if (is_true_fragment(f)) ctr[0]++;
      if (is_true_pair(f)) ctr[1]++;
      if ((f & ORC_DUPLICATE) && is_true_fragment(f)) ctr[4]++;
    }
#pragma omp critical
    for (int k = 0; k < nl * ORC_NCTR; k++) counters[k] += c[k];
    free(c);
  }
  /* duplicate pairs: toggling per {library, QNAME} shard over the SORTED reads (:182-190), then each completed pair is attached to
   * its origin in its pair-key shard (:191-222) */
  int32_t *sh_d = (int32_t *)malloc((n + 1) * sizeof(int32_t));
  int64_t *done_with = (int64_t *)malloc((n + 1) * sizeof(int64_t)); /* per sorted slot: the stored mate it completed a pair with */
#pragma omp parallel for schedule(static) num_threads(n_threads)
  for (uint64_t kk = 0; kk < n; kk++) {
    uint64_t aln = perm ? perm[kk] : kk;
    uint16_t f = flag_out[aln];
    done_with[kk] = -1;
    sh_d[kk] = -1;
    if (b->has_sr && b->has_sr[aln]) continue;
    if ((f & (ORC_UNMAPPED | ORC_SECONDARY | ORC_SUPPLEMENTARY)) || !(f & ORC_DUPLICATE) || !is_true_pair(f)) continue;
    sh_d[kk] = (int32_t)(qname_key_hash(&g, aln) % (uint64_t)n_shards);
  }
  shard_lists ld;
  if (build_shards(n, sh_d, n_shards, n_threads, &ld)) return -2;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int sh = 0; sh < n_shards; sh++) {
    qmap pf;
    if (qmap_init(&pf, ld.start[sh + 1] - ld.start[sh])) { rc_all = -2; continue; }
    for (uint64_t k = ld.start[sh]; k < ld.start[sh + 1]; k++) {
      uint64_t kk = ld.idx[k], aln = perm ? perm[kk] : kk;
      int64_t e = qmap_delete_or_store(&pf, b, g.lib_of[aln], g.lib_of, aln);
      if (e >= 0) done_with[kk] = e;
    }
    qmap_free(&pf);
  }
  /* completed duplicate pairs -> their pair-key shard (sorted order kept) */
#pragma omp parallel for schedule(static) num_threads(n_threads)
  for (uint64_t kk = 0; kk < n; kk++) {
    sh_d[kk] = -1;
    if (done_with[kk] < 0) continue;
    uint64_t a1 = perm ? perm[kk] : kk, a2 = (uint64_t)done_with[kk];
    pair_key k;
    order_pair(&g, &a1, &a2, &k);
    sh_d[kk] = (int32_t)(hash_bytes((const uint8_t *)&k, sizeof k) % (uint64_t)n_shards);
  }
  free_shards(&ld);
  if (build_shards(n, sh_d, n_shards, n_threads, &ld)) return -2;
  int64_t *pairdup = (int64_t *)calloc((size_t)n_shards * nl, sizeof(int64_t)), *opt = (int64_t *)calloc((size_t)n_shards * nl, sizeof(int64_t));
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int sh = 0; sh < n_shards; sh++) {
    md_state *s = &ps[sh];
    uint64_t cnt = ld.start[sh + 1] - ld.start[sh];
    free(s->cons);
    s->cons = (aln_cons *)malloc((cnt + 1) * sizeof(aln_cons));
    s->n_cons = 0;
    for (uint64_t k = ld.start[sh]; k < ld.start[sh + 1]; k++) {
      uint64_t kk = ld.idx[k], aln1 = perm ? perm[kk] : kk, aln2 = (uint64_t)done_with[kk];
      int lib = g.lib_of[aln1] == ORC_NIL16 ? h->n_lib : g.lib_of[aln1];
      pairdup[(size_t)sh * nl + lib]++; /* ctr.ReadPairDuplicates++ (:192), on the read that completes the pair */
      pair_key key;
      order_pair(&g, &aln1, &aln2, &key);
      int found;
      int64_t *slot = fmap_get(&s->pairs, &key, 0, 0, &found);
      if (!found) { rc_all = -4; continue; }
      pair_rec *best = &s->prec[*slot];
      if ((uint64_t)best->aln1 != aln1) {
        aln_cons *c = &s->cons[s->n_cons];
        c->aln = (int64_t)((flag_out[aln1] & ORC_FIRST) ? aln1 : aln2);
        c->next = best->opt_head;
        best->opt_head = (int64_t)s->n_cons++;
      }
    }
    /* countOpticalDuplicatesPairs :370-431 over this shard's origins */
    tinfo *fw = NULL, *rv = NULL;
    size_t cap_f = 0, cap_r = 0;
    for (uint64_t p = 0; p < s->n_prec; p++) {
      pair_rec *origin = &s->prec[p];
      size_t nf = 0, nr = 0;
      uint64_t oa = (flag_out[origin->aln1] & ORC_FIRST) ? (uint64_t)origin->aln1 : (uint64_t)origin->aln2;
      for (int pass = 0; pass < 2; pass++) {
        size_t cf = 0, cr = 0;
        if (flag_out[oa] & ORC_REVERSED) { if (pass) get_tinfo(b, oa, &rv[cr]); cr++; }
        else { if (pass) get_tinfo(b, oa, &fw[cf]); cf++; }
        for (int64_t e = origin->opt_head; e >= 0; e = s->cons[e].next) {
          uint64_t a = (uint64_t)s->cons[e].aln;
          if (flag_out[a] & ORC_REVERSED) { if (cr <= 300000) { if (pass) get_tinfo(b, a, &rv[cr]); cr++; } }
          else { if (cf <= 300000) { if (pass) get_tinfo(b, a, &fw[cf]); cf++; } }
        }
        if (!pass) {
          nf = cf; nr = cr;
          if (nf > cap_f) { cap_f = nf * 2; fw = (tinfo *)realloc(fw, cap_f * sizeof(tinfo)); }
          if (nr > cap_r) { cap_r = nr * 2; rv = (tinfo *)realloc(rv, cap_r * sizeof(tinfo)); }
        }
      }
      int lib = g.lib_of[origin->aln1] == ORC_NIL16 ? h->n_lib : g.lib_of[origin->aln1];
      opt[(size_t)sh * nl + lib] += count_from_slice(fw, (int)nf, pixel_dist) + count_from_slice(rv, (int)nr, pixel_dist);
    }
    free(fw); free(rv);
  }
  for (int sh = 0; sh < n_shards; sh++) {
    for (int l = 0; l < nl; l++) {
      counters[(size_t)l * ORC_NCTR + 5] += pairdup[(size_t)sh * nl + l];
      counters[(size_t)l * ORC_NCTR + 6] += opt[(size_t)sh * nl + l];
    }
    fmap_free(&ps[sh].pairs); free(ps[sh].prec); free(ps[sh].cons);
  }
  for (int l = 0; l < nl; l++) counters[(size_t)l * ORC_NCTR + 1] /= 2; /* :504-506 */
  free(pairdup); free(opt); free(ps); free(sh_f); free(sh_q); free(sh_p); free(sh_d); free(pair_of); free(done_with);
  free_shards(&lf); free_shards(&lq); free_shards(&lp); free_shards(&ld);
  free(g.lib_of); free(g.upos); free(g.score);
  return rc_all;
}

/* :537-569 */
#include <math.h>
static double f_lib(double x, double c, double n) { return c / x - 1 + orc_go_exp(-n / x); } /* (math.Exp: the pure-Go function, orc_gomath.c) */
int64_t orc_estimate_library_size(int64_t n_pairs, int64_t n_unique_pairs) {
  double n = (double)n_pairs, c = (double)n_unique_pairs;
  int64_t dups = n_pairs - n_unique_pairs;
  if (n_pairs > 0 && dups > 0) {
    double m = 1.0, M = 100.0;
    double fd = f_lib(M * c, c, n);
    while (fd >= 0.0) { M *= 10.0; fd = f_lib(M * c, c, n); }
    for (int i = 0; i < 40; i++) {
      double r = (m + M) / 2.0;
      double u = f_lib(r * c, c, n);
      if (u == 0.0) break;
      if (u > 0.0) m = r;
      if (u < 0.0) M = r;
    }
    return (int64_t)(c * ((m + M) / 2.0));
  }
  return 0;
}
