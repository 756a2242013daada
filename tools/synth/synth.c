/*
 * tools/synth/synth.c — deterministic synthetic 150 bp paired-end data for tests and bench.py
 * (SURVEY.md §8(d) / BASELINE.md §3).  Test/bench infrastructure, not product code.
 *
 * Everything is a pure function of (seed, pair index, field) through a counter-based hash, so any
 * sub-range of pairs can be generated independently (OpenMP here; per-rank shards in bench.py).
 *
 * Layout produced = the column-wise batch of include/elprep_hip.h (elp_batch) in queryname-grouped
 * input order: mates adjacent, pairs in generation order, optional supplementary/secondary records
 * directly after their pair.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct synth_cfg {
  uint64_t seed;
  int32_t n_ref;
  const int32_t *ref_len;
  int32_t read_len;        /* 150 */
  double p_dup;            /* 0.10  pair copies an earlier pair's unclipped ends/strands */
  double p_optical;        /* 0.15  of duplicates: same lane+tile, |dx|,|dy| <= 80 */
  double p_unmapped_pair;  /* 0.01 */
  double p_mate_unmapped;  /* 0.01 */
  double p_supp;           /* 0.005 */
  double p_sec;            /* 0.003 */
  double p_spread;         /* 0.02 mates on different contigs */
  double p_frag;           /* 0.0  unpaired single-end records (flag 0/16) — extra coverage for fragment logic */
  int32_t n_lanes;         /* 4 read groups, one per lane; lanes 1..n/2 -> lib 0, rest -> lib 1 */
  int32_t qual_mode;       /* 0 = binned qualities (2, 12, 23, 25, 27, 32, 37: NovaSeq-style), 1 = full range (~40 distinct values 2..41, as an
                              unbinned HiSeq run reports them) */
  int32_t home_lo, home_hi; /* the fragment's contig is drawn from [home_lo, home_hi) (both 0 = all contigs); the mate of a spread pair
                              is drawn from the whole genome.  Used to generate the reads of one contig group directly (sfm-style shards). */
  uint64_t ref_seed;       /* seed of the reference sequence and the known sites (0 = `seed`): shards generated with seeds of their own
                              must still be reads of ONE genome */
} synth_cfg;
static inline uint64_t rseed(const struct synth_cfg *c) { return c->ref_seed ? c->ref_seed : c->seed; }

typedef struct synth_sizes { uint64_t n_records, qname_bytes, cigar_ops, seq_bytes, qual_bytes; } synth_sizes;

typedef struct synth_out {
  int32_t *refid, *pos, *next_refid, *pnext, *tlen;
  uint16_t *flag; uint8_t *mapq; uint16_t *rgid; uint8_t *has_sr; uint32_t *l_seq;
  uint64_t *qname_off; uint8_t *qname;
  uint64_t *cigar_off; uint32_t *cigar;
  uint64_t *seq_off; uint8_t *seq4;
  uint64_t *qual_off; uint8_t *qual;
} synth_out;

static inline uint64_t sm64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
static inline uint64_t hsh(uint64_t seed, uint64_t a, uint64_t stream) { return sm64(seed ^ sm64(a * 0xD1B54A32D192ED03ull + stream * 0x8CB92BA72F3D8DD7ull)); }
static inline double u01(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }

/* ---- reference genome: uniform ACGT with ~0.1 % N runs ---- */
static inline uint8_t ref_base(const synth_cfg *c, int refid, int64_t pos0) {
  uint64_t blk = (uint64_t)pos0 / 4096;
  uint64_t hb = hsh(rseed(c), ((uint64_t)refid << 40) ^ blk, 101);
  if ((hb & 0xFF) < 10) { /* ~4 % of 4 KiB blocks carry one N run of 20..120 bases => ~0.07 % N */
    uint32_t off = (uint32_t)((hb >> 8) % 3900), len = 20 + (uint32_t)((hb >> 24) % 101);
    uint32_t in = (uint32_t)((uint64_t)pos0 % 4096);
    if (in >= off && in < off + len) return 'N';
  }
  uint64_t h = hsh(rseed(c), ((uint64_t)refid << 40) ^ ((uint64_t)pos0 >> 5), 102);
  return "ACGT"[(h >> (2 * ((uint64_t)pos0 & 31))) & 3];
}
void synth_reference(const synth_cfg *c, int refid, uint8_t *out) {
  int64_t n = c->ref_len[refid];
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) out[i] = ref_base(c, refid, i);
}

/* ---- known sites: one 1-bp site per 1000 bp + one 3..20 bp interval per 50 kbp (1-based inclusive) ---- */
static inline int32_t site1_pos(const synth_cfg *c, int refid, int64_t k) { /* 1-based position of the site in block k */
  return (int32_t)(k * 1000 + 1 + (int64_t)(hsh(rseed(c), ((uint64_t)refid << 40) ^ (uint64_t)k, 201) % 1000));
}
static inline void site2_iv(const synth_cfg *c, int refid, int64_t k, int32_t *s, int32_t *e) {
  uint64_t h = hsh(rseed(c), ((uint64_t)refid << 40) ^ (uint64_t)k, 202);
  *s = (int32_t)(k * 50000 + 1 + (int64_t)(h % 49000));
  *e = *s + 2 + (int32_t)((h >> 32) % 18);
}
static inline int is_known_site(const synth_cfg *c, int refid, int64_t pos1) {
  int64_t k = (pos1 - 1) / 1000;
  if (site1_pos(c, refid, k) == pos1) return 1;
  int64_t k2 = (pos1 - 1) / 50000;
  int32_t s, e;
  site2_iv(c, refid, k2, &s, &e);
  return pos1 >= s && pos1 <= e;
}
/* writes unsorted, possibly overlapping raw intervals (as a VCF-derived .elsites would hold); returns count */
int64_t synth_known_sites(const synth_cfg *c, int refid, int32_t *start, int32_t *end, int64_t cap) {
  int64_t len = c->ref_len[refid], n = 0;
  for (int64_t k = 0; k * 1000 < len; k++) {
    int32_t p = site1_pos(c, refid, k);
    if (p <= len) { if (n < cap) { start[n] = p; end[n] = p; } n++; }
  }
  for (int64_t k = 0; k * 50000 < len; k++) {
    int32_t s, e;
    site2_iv(c, refid, k, &s, &e);
    if (e <= len) { if (n < cap) { start[n] = s; end[n] = e; } n++; }
  }
  return n;
}

/* ---- pair layout ---- */
typedef struct {
  int unmapped_pair, mate_unmapped, spread, frag;
  int refid, refid2;
  int32_t start;     /* fragment start, 1-based (unclipped 5' end of the forward read) */
  int32_t insert;    /* fragment length */
  int first_rev;     /* first mate is the reverse read */
  int32_t start2;    /* spread pairs: position of the second read on refid2 */
  int lane, tile, x, y;
} layout;

static int pick_contig(const synth_cfg *c, uint64_t h, int lo, int hi, int32_t *start_out, int32_t span) {
  /* contig of [lo, hi) proportional to length, start uniform such that [start, start+span) fits */
  int64_t total = 0;
  for (int i = lo; i < hi; i++) total += c->ref_len[i];
  int64_t r = (int64_t)(u01(h) * (double)total);
  int id = hi - 1;
  for (int i = lo; i < hi; i++) { if (r < c->ref_len[i]) { id = i; break; } r -= c->ref_len[i]; }
  int64_t room = (int64_t)c->ref_len[id] - span - 400;
  if (room < 1) room = 1;
  *start_out = (int32_t)(201 + (int64_t)(sm64(h) % (uint64_t)room));
  return id;
}

static void base_layout(const synth_cfg *c, uint64_t t, layout *L) {
  memset(L, 0, sizeof *L);
  double u = u01(hsh(c->seed, t, 1));
  double a = c->p_unmapped_pair, b = a + c->p_mate_unmapped, s = b + c->p_spread, f = s + c->p_frag;
  if (u < a) L->unmapped_pair = 1;
  else if (u < b) L->mate_unmapped = 1;
  else if (u < s) L->spread = 1;
  else if (u < f) L->frag = 1;
  /* insert ~ round(N(400, 90^2)) clamped to [60, 1000] (Box-Muller) */
  double u1 = u01(hsh(c->seed, t, 2)), u2 = u01(hsh(c->seed, t, 3));
  if (u1 < 1e-12) u1 = 1e-12;
  double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
  int32_t ins = (int32_t)lround(400.0 + 90.0 * z);
  if (ins < 60) ins = 60;
  if (ins > 1000) ins = 1000;
  L->insert = ins;
  const int hlo = (c->home_hi > c->home_lo) ? c->home_lo : 0, hhi = (c->home_hi > c->home_lo) ? c->home_hi : c->n_ref;
  L->refid = pick_contig(c, hsh(c->seed, t, 4), hlo, hhi, &L->start, ins > c->read_len ? ins : c->read_len);
  L->first_rev = (int)(hsh(c->seed, t, 5) & 1);
  if (L->spread) {
    if (c->n_ref < 2) L->spread = 0;
    else {
      L->refid2 = pick_contig(c, hsh(c->seed, t, 6), 0, c->n_ref, &L->start2, c->read_len);
      if (L->refid2 == L->refid) L->refid2 = (L->refid + 1) % c->n_ref, L->start2 = 201 + (int32_t)(hsh(c->seed, t, 7) % (uint64_t)(c->ref_len[L->refid2] > 1000 ? c->ref_len[L->refid2] - 800 : 1));
    }
  }
  uint64_t hq = hsh(c->seed, t, 8);
  L->lane = 1 + (int)(hq % (uint64_t)c->n_lanes);
  L->tile = 1101 + (int)((hq >> 8) % 1128);
  L->x = 1000 + (int)((hq >> 24) % 19001);
  L->y = 1000 + (int)((hq >> 44) % 19001);
}

static int lane_lib(const synth_cfg *c, int lane) { return (lane - 1) < (c->n_lanes + 1) / 2 ? 0 : 1; }

static void pair_layout(const synth_cfg *c, uint64_t p, layout *L, int *is_dup) {
  base_layout(c, p, L);
  *is_dup = 0;
  if (p == 0 || L->unmapped_pair) return;
  double u = u01(hsh(c->seed, p, 20));
  if (u >= c->p_dup) return;
  uint64_t src = hsh(c->seed, p, 21) % p;
  layout S;
  base_layout(c, src, &S);
  if (S.unmapped_pair || S.mate_unmapped != L->mate_unmapped || S.spread != L->spread || S.frag != L->frag) return;
  *is_dup = 1;
  int own_lane = L->lane, own_tile = L->tile, own_x = L->x, own_y = L->y;
  *L = S;
  uint64_t ho = hsh(c->seed, p, 22);
  if (u01(ho) < c->p_optical) { /* optical: same lane + tile, nearby x/y */
    int dx = (int)((ho >> 20) % 161) - 80, dy = (int)((ho >> 40) % 161) - 80;
    L->x = S.x + dx; L->y = S.y + dy;
    if (L->x < 0) L->x = 0;
    if (L->y < 0) L->y = 0;
  } else { /* PCR duplicate elsewhere on the flowcell, same library */
    int lib = lane_lib(c, S.lane);
    int half = (c->n_lanes + 1) / 2;
    int nl = lib == 0 ? half : c->n_lanes - half;
    if (nl < 1) nl = 1;
    L->lane = (lib == 0 ? 1 : half + 1) + (own_lane % nl);
    L->tile = own_tile; L->x = own_x; L->y = own_y;
  }
}

/* ---- one read ---- */
typedef struct { uint32_t ops[8]; int n; int32_t lead_clip, trail_clip, ref_len; int32_t hard_lead; } cig;

/* CIGAR: 150M 84 %, leading and/or trailing soft clip 1..30 8 %, one insertion 3 %, one deletion 3 %, xH prefix 0.5 %, two indels 1.5 % */
static void make_cigar(const synth_cfg *c, uint64_t h, int L, cig *g) {
  memset(g, 0, sizeof *g);
  double u = u01(h);
  uint64_t h2 = sm64(h);
#define OP(len, op) (g->ops[g->n++] = ((uint32_t)(len) << 4) | (op))
  enum { M = 0, I = 1, D = 2, S = 4, H = 5 };
  if (u < 0.84) { OP(L, M); g->ref_len = L; }
  else if (u < 0.92) {
    int which = (int)(h2 % 3); /* 0 lead, 1 trail, 2 both */
    int a = 1 + (int)((h2 >> 8) % 30), b = 1 + (int)((h2 >> 16) % 30);
    if (which == 1) a = 0;
    if (which == 0) b = 0;
    if (a) OP(a, S);
    OP(L - a - b, M);
    if (b) OP(b, S);
    g->lead_clip = a; g->trail_clip = b; g->ref_len = L - a - b;
  } else if (u < 0.95) {
    int il = 1 + (int)(h2 % 5), at = 10 + (int)((h2 >> 8) % (uint64_t)(L - 30));
    OP(at, M); OP(il, I); OP(L - at - il, M);
    g->ref_len = L - il;
  } else if (u < 0.98) {
    int dl = 1 + (int)(h2 % 5), at = 10 + (int)((h2 >> 8) % (uint64_t)(L - 30));
    OP(at, M); OP(dl, D); OP(L - at, M);
    g->ref_len = L + dl;
  } else if (u < 0.985) {
    int hl = 1 + (int)(h2 % 40);
    OP(hl, H); OP(L, M);
    g->hard_lead = hl; g->lead_clip = hl; g->ref_len = L;
  } else {
    int il = 1 + (int)(h2 % 5), dl = 1 + (int)((h2 >> 4) % 5);
    int at = 10 + (int)((h2 >> 8) % 40), at2 = 10 + (int)((h2 >> 16) % 40);
    OP(at, M); OP(il, I); OP(at2, M); OP(dl, D); OP(L - at - il - at2, M);
    g->ref_len = L - il + dl;
  }
#undef OP
  (void)c;
}

typedef struct {
  const synth_cfg *c;
  synth_out *o;       /* NULL in counting mode */
  synth_sizes z;      /* running offsets */
} emit;

static int qname_fmt(char *buf, uint64_t p, const layout *L) {
  return sprintf(buf, "SIM%llx:1:FC1:%d:%d:%d:%d", (unsigned long long)p, L->lane, L->tile, L->x, L->y);
}

/* emits one record; bases/quals generated from (p, rec) */
static void emit_record(emit *e, uint64_t p, int rec, const layout *L, const char *qn, int qn_len, int32_t refid, int32_t pos,
                        int32_t next_refid, int32_t pnext, int32_t tlen, uint16_t flag, uint8_t mapq, const cig *g, int reversed_decay) {
  const synth_cfg *c = e->c;
  int RL = c->read_len;
  uint64_t r = e->z.n_records;
  if (e->o) {
    synth_out *o = e->o;
    o->refid[r] = refid; o->pos[r] = pos; o->next_refid[r] = next_refid; o->pnext[r] = pnext; o->tlen[r] = tlen;
    o->flag[r] = flag; o->mapq[r] = mapq; o->rgid[r] = (uint16_t)(L->lane - 1); o->has_sr[r] = 0; o->l_seq[r] = (uint32_t)RL;
    o->qname_off[r] = e->z.qname_bytes; memcpy(o->qname + e->z.qname_bytes, qn, (size_t)qn_len);
    o->cigar_off[r] = e->z.cigar_ops; for (int k = 0; k < g->n; k++) o->cigar[e->z.cigar_ops + k] = g->ops[k];
    o->seq_off[r] = e->z.seq_bytes; o->qual_off[r] = e->z.qual_bytes;
    /* qualities */
    uint8_t *q = o->qual + e->z.qual_bytes;
    uint64_t hm = hsh(c->seed, p * 8 + (uint64_t)rec, 30);
    double um = u01(hm);
    int mean = um < 0.7 ? 37 : (um < 0.9 ? 30 : 20);
    double extra = mean == 37 ? 0.0 : (mean == 30 ? 0.2 : 0.4);
    uint64_t hq = 0;
    for (int i = 0; i < RL; i++) {
      if ((i & 3) == 0) hq = hsh(c->seed, (p * 8 + (uint64_t)rec) * 64 + (uint64_t)(i >> 2), 31);
      double u = (double)((hq >> (16 * (i & 3))) & 0xFFFF) / 65536.0;
      int cyc = reversed_decay ? RL - 1 - i : i;
      double pd = 0.05 + 0.25 * ((double)cyc / RL) + extra;
      int qq;
      if (u < pd * 0.1) qq = 2;
      else if (u < pd * 0.4) qq = 12;
      else if (u < pd * 0.7) qq = 23;
      else if (u < pd) qq = 27;
      else qq = mean == 37 ? 37 : (mean == 30 ? 32 : 25);
      if (c->qual_mode == 1 && qq != 2) { /* spread every bin over +-4 around its centre */
        uint64_t hd = sm64(hq ^ (0x9E3779B97F4A7C15ull * (uint64_t)(i + 1)));
        qq += (int)(hd % 9) - 4;
        if (qq > 41) qq = 41;
        if (qq < 3) qq = 3;
      }
      q[i] = (uint8_t)qq;
    }
    if (u01(sm64(hm)) < 0.1) { /* low-quality tail at the 3' end of the sequencing direction */
      int tl = (int)(sm64(hm + 1) % 6);
      for (int i = 0; i < tl; i++) q[reversed_decay ? i : RL - 1 - i] = 2;
    }
    /* bases, reference orientation */
    uint8_t *s4 = o->seq4 + e->z.seq_bytes;
    memset(s4, 0, (size_t)(RL + 1) / 2);
    int ri = 0;
    int64_t rp = (int64_t)pos; /* 1-based ref position of next aligned base */
    uint64_t hb = 0;
    int hbi = 0;
    for (int k = 0; k <= g->n; k++) {
      int op = k < g->n ? (int)(g->ops[k] & 0xF) : -1, len = k < g->n ? (int)(g->ops[k] >> 4) : 0;
      if ((refid < 0 || g->n == 0) && k == 0) { op = 4; len = RL; } /* unmapped: random bases */
      else if (refid < 0 || g->n == 0) break;
      if (op < 0) break;
      if (op == 5 || op == 6) continue;
      if (op == 2 || op == 3) { rp += len; continue; }
      for (int t = 0; t < len && ri < RL; t++, ri++) {
        if ((hbi & 7) == 0) hb = hsh(c->seed, (p * 8 + (uint64_t)rec) * 64 + (uint64_t)(hbi >> 3), 32);
        uint32_t rnd = (uint32_t)((hb >> (8 * (hbi & 7))) & 0xFF);
        hbi++;
        uint8_t base;
        if (op == 0 || op == 7 || op == 8) {
          uint8_t rb = (rp >= 1 && rp <= c->ref_len[refid]) ? ref_base(c, refid, rp - 1) : 'N';
          base = rb;
          /* sequencing error w.p. 10^(-q/10); true SNP at known sites w.p. 0.5 */
          double pe = pow(10.0, -(double)q[ri] / 10.0);
          uint64_t he = hsh(c->seed, (p * 8 + (uint64_t)rec) * 256 + (uint64_t)ri, 33);
          int mutate = u01(he) < pe;
          if (!mutate && is_known_site(c, refid, rp) && ((he >> 5) & 1)) mutate = 1;
          if (rb == 'N') base = "ACGT"[rnd & 3];
          else if (mutate) { const char *alt = "ACGT"; int bi = (int)(strchr(alt, rb) - alt); base = (uint8_t)alt[(bi + 1 + (int)(rnd % 3)) & 3]; }
          rp++;
        } else {
          base = "ACGT"[rnd & 3];
        }
        if ((rnd & 0xFF) == 0xFF && (hb & 0x300) == 0) base = 'N'; /* rare no-call (~0.1 %) */
        uint8_t nib = base == 'A' ? 1 : base == 'C' ? 2 : base == 'G' ? 4 : base == 'T' ? 8 : 15;
        s4[ri >> 1] |= (ri & 1) ? nib : (uint8_t)(nib << 4);
      }
    }
    for (; ri < RL; ri++) { uint8_t nib = 1; s4[ri >> 1] |= (ri & 1) ? nib : (uint8_t)(nib << 4); }
  }
  e->z.n_records++;
  e->z.qname_bytes += (uint64_t)qn_len;
  e->z.cigar_ops += (uint64_t)g->n;
  e->z.seq_bytes += (uint64_t)(RL + 1) / 2;
  e->z.qual_bytes += (uint64_t)RL;
}

static uint8_t make_mapq(uint64_t h) {
  double u = u01(h);
  if (u < 0.88) return 60;
  if (u < 0.92) return 0;
  return (uint8_t)(1 + (sm64(h) % 59));
}

static void gen_pair(emit *e, uint64_t p) {
  const synth_cfg *c = e->c;
  int RL = c->read_len;
  layout L;
  int is_dup;
  pair_layout(c, p, &L, &is_dup);
  char qn[96];
  int ql = qname_fmt(qn, p, &L);
  cig g1, g2, gz;
  memset(&gz, 0, sizeof gz);
  if (L.unmapped_pair) {
    emit_record(e, p, 0, &L, qn, ql, -1, 0, -1, 0, 0, 77, 0, &gz, 0);
    emit_record(e, p, 1, &L, qn, ql, -1, 0, -1, 0, 0, 141, 0, &gz, 0);
    return;
  }
  make_cigar(c, hsh(c->seed, p, 40), RL, &g1); /* forward read */
  make_cigar(c, hsh(c->seed, p, 41), RL, &g2); /* reverse read */
  /* forward read: unclipped start = L.start  =>  POS = start + leading clip */
  int32_t pos_f = L.start + g1.lead_clip;
  /* reverse read: unclipped end = start + insert - 1  =>  POS = end + 1 - ref_len - trailing clip */
  int32_t end_r = L.start + L.insert - 1;
  int32_t pos_r = end_r + 1 - g2.ref_len - g2.trail_clip;
  if (pos_r < 1) pos_r = 1;
  uint8_t mq_f = make_mapq(hsh(c->seed, p, 42)), mq_r = make_mapq(hsh(c->seed, p, 43));
  if (L.frag) { /* single-end record */
    if (L.first_rev) emit_record(e, p, 0, &L, qn, ql, L.refid, pos_r, -1, 0, 0, 16, mq_r, &g2, 1);
    else emit_record(e, p, 0, &L, qn, ql, L.refid, pos_f, -1, 0, 0, 0, mq_f, &g1, 0);
    return;
  }
  if (L.mate_unmapped) {
    /* mapped read (first) + unmapped mate placed at the mapped read's RNAME/POS */
    if (L.first_rev) {
      emit_record(e, p, 0, &L, qn, ql, L.refid, pos_r, L.refid, pos_r, 0, 0x1 | 0x8 | 0x10 | 0x40, mq_r, &g2, 1);       /* 89 */
      emit_record(e, p, 1, &L, qn, ql, L.refid, pos_r, L.refid, pos_r, 0, 0x1 | 0x4 | 0x20 | 0x80, 0, &gz, 0);          /* 165 */
    } else {
      emit_record(e, p, 0, &L, qn, ql, L.refid, pos_f, L.refid, pos_f, 0, 0x1 | 0x8 | 0x40, mq_f, &g1, 0);              /* 73 */
      emit_record(e, p, 1, &L, qn, ql, L.refid, pos_f, L.refid, pos_f, 0, 0x1 | 0x4 | 0x80, 0, &gz, 0);                 /* 133 */
    }
    return;
  }
  if (L.spread) {
    /* first mate forward on refid, second mate reverse on refid2 */
    int32_t pos2 = L.start2;
    emit_record(e, p, 0, &L, qn, ql, L.refid, pos_f, L.refid2, pos2, 0, 0x1 | 0x20 | 0x40, mq_f, &g1, 0);  /* 97 */
    emit_record(e, p, 1, &L, qn, ql, L.refid2, pos2, L.refid, pos_f, 0, 0x1 | 0x10 | 0x80, mq_r, &g2, 1);  /* 145 */
  } else {
    int32_t tl = L.insert;
    uint16_t ff, fr;
    if (L.first_rev) { fr = 83; ff = 163; } else { ff = 99; fr = 147; }
    int rec_f = L.first_rev ? 1 : 0, rec_r = L.first_rev ? 0 : 1;
    /* emit in template order: first mate, then last mate */
    if (L.first_rev) {
      emit_record(e, p, rec_r, &L, qn, ql, L.refid, pos_r, L.refid, pos_f, -tl, fr, mq_r, &g2, 1);
      emit_record(e, p, rec_f, &L, qn, ql, L.refid, pos_f, L.refid, pos_r, tl, ff, mq_f, &g1, 0);
    } else {
      emit_record(e, p, rec_f, &L, qn, ql, L.refid, pos_f, L.refid, pos_r, tl, ff, mq_f, &g1, 0);
      emit_record(e, p, rec_r, &L, qn, ql, L.refid, pos_r, L.refid, pos_f, -tl, fr, mq_r, &g2, 1);
    }
  }
  double ux = u01(hsh(c->seed, p, 50));
  if (ux < c->p_supp) { /* supplementary record of the forward read somewhere else */
    int32_t sp; int sref = pick_contig(c, hsh(c->seed, p, 51), 0, c->n_ref, &sp, RL);
    cig gs; memset(&gs, 0, sizeof gs);
    int cl = 40 + (int)(hsh(c->seed, p, 52) % 60);
    gs.ops[0] = ((uint32_t)cl << 4) | 5; gs.ops[1] = ((uint32_t)RL << 4) | 0; gs.n = 2; gs.ref_len = RL;
    emit_record(e, p, 2, &L, qn, ql, sref, sp, L.refid, pos_r, 0, (uint16_t)((L.first_rev ? 163 : 99) | 0x800), make_mapq(hsh(c->seed, p, 53)), &gs, 0);
  } else if (ux < c->p_supp + c->p_sec) { /* secondary record */
    int32_t sp; int sref = pick_contig(c, hsh(c->seed, p, 54), 0, c->n_ref, &sp, RL);
    cig gs; memset(&gs, 0, sizeof gs);
    gs.ops[0] = ((uint32_t)RL << 4) | 0; gs.n = 1; gs.ref_len = RL;
    emit_record(e, p, 3, &L, qn, ql, sref, sp, L.refid, pos_r, 0, (uint16_t)((L.first_rev ? 163 : 99) | 0x100), 0, &gs, 0);
  }
}

#define CHUNK 8192ull

int synth_plan(const synth_cfg *c, uint64_t pair_lo, uint64_t pair_hi, synth_sizes *out) {
  uint64_t nchunk = (pair_hi - pair_lo + CHUNK - 1) / CHUNK;
  synth_sizes tot = {0, 0, 0, 0, 0};
#pragma omp parallel
  {
    synth_sizes loc = {0, 0, 0, 0, 0};
#pragma omp for schedule(dynamic, 4)
    for (uint64_t ch = 0; ch < nchunk; ch++) {
      emit e; e.c = c; e.o = NULL; memset(&e.z, 0, sizeof e.z);
      uint64_t lo = pair_lo + ch * CHUNK, hi = lo + CHUNK < pair_hi ? lo + CHUNK : pair_hi;
      for (uint64_t p = lo; p < hi; p++) gen_pair(&e, p);
      loc.n_records += e.z.n_records; loc.qname_bytes += e.z.qname_bytes; loc.cigar_ops += e.z.cigar_ops;
      loc.seq_bytes += e.z.seq_bytes; loc.qual_bytes += e.z.qual_bytes;
    }
#pragma omp critical
    {
      tot.n_records += loc.n_records; tot.qname_bytes += loc.qname_bytes; tot.cigar_ops += loc.cigar_ops;
      tot.seq_bytes += loc.seq_bytes; tot.qual_bytes += loc.qual_bytes;
    }
  }
  *out = tot;
  return 0;
}

int synth_fill(const synth_cfg *c, uint64_t pair_lo, uint64_t pair_hi, synth_out *o) {
  uint64_t nchunk = (pair_hi - pair_lo + CHUNK - 1) / CHUNK;
  if (nchunk == 0) { o->qname_off[0] = o->cigar_off[0] = o->seq_off[0] = o->qual_off[0] = 0; return 0; }
  synth_sizes *base = (synth_sizes *)calloc(nchunk + 1, sizeof(synth_sizes));
  if (!base) return -1;
#pragma omp parallel for schedule(dynamic, 4)
  for (uint64_t ch = 0; ch < nchunk; ch++) {
    emit e; e.c = c; e.o = NULL; memset(&e.z, 0, sizeof e.z);
    uint64_t lo = pair_lo + ch * CHUNK, hi = lo + CHUNK < pair_hi ? lo + CHUNK : pair_hi;
    for (uint64_t p = lo; p < hi; p++) gen_pair(&e, p);
    base[ch + 1] = e.z;
  }
  for (uint64_t ch = 0; ch < nchunk; ch++) {
    base[ch + 1].n_records += base[ch].n_records; base[ch + 1].qname_bytes += base[ch].qname_bytes;
    base[ch + 1].cigar_ops += base[ch].cigar_ops; base[ch + 1].seq_bytes += base[ch].seq_bytes;
    base[ch + 1].qual_bytes += base[ch].qual_bytes;
  }
#pragma omp parallel for schedule(dynamic, 4)
  for (uint64_t ch = 0; ch < nchunk; ch++) {
    emit e; e.c = c; e.o = o; e.z = base[ch];
    uint64_t lo = pair_lo + ch * CHUNK, hi = lo + CHUNK < pair_hi ? lo + CHUNK : pair_hi;
    for (uint64_t p = lo; p < hi; p++) gen_pair(&e, p);
  }
  synth_sizes t = base[nchunk];
  o->qname_off[t.n_records] = t.qname_bytes; o->cigar_off[t.n_records] = t.cigar_ops;
  o->seq_off[t.n_records] = t.seq_bytes; o->qual_off[t.n_records] = t.qual_bytes;
  free(base);
  return 0;
}
