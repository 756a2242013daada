/*
 * oracle/orc_bam.c — CPU oracle (test infrastructure; see orc.h): BAM alignment records.
 *
 * Restates formatBamAlignment (sam/bam-files.go:635-737), Alignment.bin() (:443-468) and formatBamTag's integer rule (:492-525).
 * One function writes the records of a batch in a given order; it serves two purposes in the tests:
 *   - the INPUT of elp_stage_bam: staging order, optional fields in deliberately non-minimal integer types (what other writers
 *     produce), the sr:i tag on tagged copies;
 *   - the EXPECTED OUTPUT of elp_emit_sorted_bam: sorted order, new FLAG / QUAL columns, optional fields as elPrep re-encodes
 *     them after parsing (parseBamAlignment :373-396 turns every integer into int64; formatBamTag picks the smallest type).
 * The optional fields of record i are a deterministic function of i and of the record's fields, so both calls agree on them.
 */
#include "orc.h"
#include <string.h>

static void put16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void put32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); }

/* sam/bam-files.go:443-468 */
static uint16_t bam_bin(int32_t pos, uint16_t flag, const uint32_t *cigar, uint32_t n_cigar) {
  int32_t beg = pos - 1, end = beg;
  if (!(flag & ORC_UNMAPPED)) {
    for (uint32_t i = 0; i < n_cigar; i++) {
      uint32_t op = cigar[i] & 0xF;
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) end += (int32_t)(cigar[i] >> 4); /* M D N = X */
    }
    end--;
  }
  if (beg >> 14 == end >> 14) return (uint16_t)(((1 << 15) - 1) / 7 + (beg >> 14));
  if (beg >> 17 == end >> 17) return (uint16_t)(((1 << 12) - 1) / 7 + (beg >> 17));
  if (beg >> 20 == end >> 20) return (uint16_t)(((1 << 9) - 1) / 7 + (beg >> 20));
  if (beg >> 23 == end >> 23) return (uint16_t)(((1 << 6) - 1) / 7 + (beg >> 23));
  if (beg >> 26 == end >> 26) return (uint16_t)(((1 << 3) - 1) / 7 + (beg >> 26));
  return 0;
}

/* one integer optional field: `raw_type` as an arbitrary writer chose it, or (normalize) as formatBamTag :492-525 writes it */
static size_t put_int_tag(uint8_t *o, const char *tag, int64_t v, char raw_type, int normalize) {
  char t = raw_type;
  if (normalize) {
    if (v < 0) t = v >= -128 ? 'c' : (v >= -32768 ? 's' : 'i');
    else t = v <= 255 ? 'C' : (v <= 65535 ? 'S' : 'I');
  }
  size_t sz = (t == 'c' || t == 'C') ? 1 : ((t == 's' || t == 'S') ? 2 : 4);
  if (o) {
    o[0] = (uint8_t)tag[0]; o[1] = (uint8_t)tag[1]; o[2] = (uint8_t)t;
    for (size_t b = 0; b < sz; b++) o[3 + b] = (uint8_t)((uint64_t)v >> (8 * b));
  }
  return 3 + sz;
}
static size_t put_str_tag(uint8_t *o, const char *tag, const char *s, size_t l) {
  if (o) { o[0] = (uint8_t)tag[0]; o[1] = (uint8_t)tag[1]; o[2] = 'Z'; memcpy(o + 3, s, l); o[3 + l] = 0; }
  return 3 + l + 1;
}

/* optional fields of record i */
static size_t put_tags(uint8_t *o, const orc_batch *b, uint64_t i, const char *const *rg_ids, int normalize) {
  size_t n = 0;
#define AT (o ? o + n : NULL)
  int64_t nm = (int64_t)((i * 2654435761u) % 7);
  n += put_int_tag(AT, "NM", nm, 'i', normalize);                               /* small value in a 4-byte type */
  if (b->rgid[i] != ORC_NIL16) n += put_str_tag(AT, "RG", rg_ids[b->rgid[i]], strlen(rg_ids[b->rgid[i]]));
  int64_t as = (int64_t)b->l_seq[i] - (int64_t)((i * 40503u) % 400);             /* positive and negative, one and two bytes */
  n += put_int_tag(AT, "AS", as, (as >= -32768 && as <= 32767) ? 's' : 'i', normalize);
  if (i % 3 == 0) {
    if (o) { o[n] = 'X'; o[n + 1] = 'T'; o[n + 2] = 'A'; o[n + 3] = (uint8_t)('A' + i % 26); }
    n += 4;
  }
  if (i % 5 == 0) n += put_int_tag(AT, "XL", (int64_t)70000 + (int64_t)(i % 1000), 'I', normalize);
  if (i % 7 == 0) { /* numeric array: copied as it is (:552-630) */
    if (o) {
      o[n] = 'X'; o[n + 1] = 'B'; o[n + 2] = 'B'; o[n + 3] = 's';
      put32(o + n + 4, 3);
      put16(o + n + 8, (uint32_t)(i & 0xFFFF)); put16(o + n + 10, 0xFFFF); put16(o + n + 12, 7);
    }
    n += 14;
  }
  if (i % 4 == 1) {
    char md[24];
    size_t l = 0;
    uint64_t v = b->l_seq[i];
    char tmp[24];
    size_t k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) md[l++] = tmp[--k];
    n += put_str_tag(AT, "MD", md, l);
  }
  if (b->has_sr && b->has_sr[i]) n += put_int_tag(AT, "sr", 1, 'i', normalize);  /* aln.TAGS.Set(sr, 1), sam/split-merge.go:291 */
#undef AT
  return n;
}

/* Writes the records order[0 .. n_order) of b (order NULL: 0 .. b->n) behind each other; flags / qual (may be NULL) replace the
 * batch's FLAG and QUAL columns.  out NULL: only the size is computed.  Returns the number of bytes. */
size_t orc_bam_encode(const orc_batch *b, const char *const *rg_ids, const uint32_t *order, uint64_t n_order, const uint16_t *flags,
                      const uint8_t *qual, int normalize_tags, uint8_t *out) {
  size_t at = 0;
  if (!order) n_order = b->n;
  for (uint64_t k = 0; k < n_order; k++) {
    uint64_t i = order ? order[k] : k;
    uint32_t lq = (uint32_t)(b->qname_off[i + 1] - b->qname_off[i]);
    uint32_t nc = (uint32_t)(b->cigar_off[i + 1] - b->cigar_off[i]);
    uint32_t ls = b->l_seq[i], sb = (ls + 1) >> 1;
    const uint32_t *cg = b->cigar + b->cigar_off[i];
    uint16_t f = flags ? flags[i] : b->flag[i];
    size_t body = 32 + (size_t)lq + 1 + 4 * (size_t)nc + sb + ls;
    size_t tags = put_tags(NULL, b, i, rg_ids, normalize_tags);
    if (out) {
      uint8_t *o = out + at;
      put32(o, (uint32_t)(body + tags));
      put32(o + 4, (uint32_t)b->refid[i]);              /* dictTable miss ("*") = 0xFFFFFFFF = -1 */
      put32(o + 8, (uint32_t)(b->pos[i] - 1));
      o[12] = (uint8_t)(lq + 1);
      o[13] = b->mapq[i];
      put16(o + 14, bam_bin(b->pos[i], f, cg, nc));
      put16(o + 16, nc);
      put16(o + 18, f);
      put32(o + 20, ls);
      put32(o + 24, (uint32_t)b->next_refid[i]);
      put32(o + 28, (uint32_t)(b->pnext[i] - 1));
      put32(o + 32, (uint32_t)b->tlen[i]);
      memcpy(o + 36, b->qname + b->qname_off[i], lq);
      o[36 + lq] = 0;
      uint8_t *w = o + 36 + lq + 1;
      for (uint32_t c = 0; c < nc; c++) put32(w + 4 * c, cg[c]);
      w += 4 * (size_t)nc;
      memcpy(w, b->seq4 + b->seq_off[i], sb);
      w += sb;
      memcpy(w, (qual ? qual : b->qual) + b->qual_off[i], ls);
      w += ls;
      put_tags(w, b, i, rg_ids, normalize_tags);
    }
    at += 4 + body + tags;
  }
  return at;
}

/* byte offset of every record of the stream orc_bam_encode writes (n_order + 1 values) */
void orc_bam_offsets(const orc_batch *b, const char *const *rg_ids, const uint32_t *order, uint64_t n_order, int normalize_tags, uint64_t *off_out) {
  size_t at = 0;
  if (!order) n_order = b->n;
  for (uint64_t k = 0; k < n_order; k++) {
    uint64_t i = order ? order[k] : k;
    off_out[k] = at;
    at += 4 + 32 + (size_t)(b->qname_off[i + 1] - b->qname_off[i]) + 1 + 4 * (size_t)(b->cigar_off[i + 1] - b->cigar_off[i]) + ((b->l_seq[i] + 1) >> 1) + b->l_seq[i] +
          put_tags(NULL, b, i, rg_ids, normalize_tags);
  }
  off_out[n_order] = at;
}
