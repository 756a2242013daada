/* tools/synth/bam_writer.c - writes the records of a generated batch as uncompressed BAM alignment records (SAMv1 4.2), the bytes a BAM
 * reader hands over after inflating the BGZF blocks.  Test / bench infrastructure standing in for that reader (bench.py's PCIe-inclusive
 * side measurement): the generator's output format, not the oracle and not the product.  Optional fields: NM:i, RG:Z, and sr:i on the
 * tagged copies of `elprep split`. */
#include <stdint.h>
#include <stddef.h>
#include <string.h>
#include "../../include/elprep_hip.h"

static void w16(uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
static void w32(uint8_t *p, uint32_t v) { w16(p, v); w16(p + 2, v >> 16); }

/* UCSC binning scheme of the SAM specification (5.3) */
static uint32_t reg2bin(int32_t beg, int32_t end) {
  --end;
  if (beg >> 14 == end >> 14) return ((1u << 15) - 1) / 7 + (uint32_t)(beg >> 14);
  if (beg >> 17 == end >> 17) return ((1u << 12) - 1) / 7 + (uint32_t)(beg >> 17);
  if (beg >> 20 == end >> 20) return ((1u << 9) - 1) / 7 + (uint32_t)(beg >> 20);
  if (beg >> 23 == end >> 23) return ((1u << 6) - 1) / 7 + (uint32_t)(beg >> 23);
  if (beg >> 26 == end >> 26) return ((1u << 3) - 1) / 7 + (uint32_t)(beg >> 26);
  return 0;
}

static size_t record_size(const elp_batch *b, uint64_t i, const char *const *rg_ids) {
  size_t s = 4 + 32 + (size_t)(b->qname_off[i + 1] - b->qname_off[i]) + 1 + 4 * (size_t)(b->cigar_off[i + 1] - b->cigar_off[i]) +
             ((size_t)b->l_seq[i] + 1) / 2 + b->l_seq[i];
  s += 3 + 1;                                                          /* NM:C */
  if (b->rgid[i] != ELP_NIL16) s += 3 + strlen(rg_ids[b->rgid[i]]) + 1; /* RG:Z */
  if (b->has_sr && b->has_sr[i]) s += 3 + 1;                           /* sr:C */
  return s;
}

/* off_out (n + 1 values, may be NULL): byte offset of every record; out NULL: sizes only.  Returns the total number of bytes. */
uint64_t synth_bam_write(const elp_batch *b, const char *const *rg_ids, uint8_t *out, uint64_t *off_out) {
  uint64_t at = 0;
  for (uint64_t i = 0; i < b->n; i++) {
    const size_t sz = record_size(b, i, rg_ids);
    if (off_out) off_out[i] = at;
    if (out) {
      uint8_t *o = out + at;
      const uint32_t lq = (uint32_t)(b->qname_off[i + 1] - b->qname_off[i]), nc = (uint32_t)(b->cigar_off[i + 1] - b->cigar_off[i]);
      const uint32_t ls = b->l_seq[i];
      const uint32_t *cg = b->cigar + b->cigar_off[i];
      int32_t beg = b->pos[i] - 1, end = beg;
      if (!(b->flag[i] & 0x4))
        for (uint32_t c = 0; c < nc; c++) { uint32_t op = cg[c] & 15u; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) end += (int32_t)(cg[c] >> 4); }
      if (end == beg) end = beg + 1;
      w32(o, (uint32_t)(sz - 4));
      w32(o + 4, (uint32_t)b->refid[i]);
      w32(o + 8, (uint32_t)beg);
      o[12] = (uint8_t)(lq + 1);
      o[13] = b->mapq[i];
      w16(o + 14, reg2bin(beg, end));
      w16(o + 16, nc);
      w16(o + 18, b->flag[i]);
      w32(o + 20, ls);
      w32(o + 24, (uint32_t)b->next_refid[i]);
      w32(o + 28, (uint32_t)(b->pnext[i] - 1));
      w32(o + 32, (uint32_t)b->tlen[i]);
      uint8_t *w = o + 36;
      memcpy(w, b->qname + b->qname_off[i], lq); w[lq] = 0; w += lq + 1;
      for (uint32_t c = 0; c < nc; c++) w32(w + 4 * c, cg[c]);
      w += 4 * (size_t)nc;
      memcpy(w, b->seq4 + b->seq_off[i], (ls + 1) / 2); w += (ls + 1) / 2;
      memcpy(w, b->qual + b->qual_off[i], ls); w += ls;
      w[0] = 'N'; w[1] = 'M'; w[2] = 'C'; w[3] = (uint8_t)(i % 5); w += 4;
      if (b->rgid[i] != ELP_NIL16) {
        const char *id = rg_ids[b->rgid[i]];
        const size_t l = strlen(id);
        w[0] = 'R'; w[1] = 'G'; w[2] = 'Z'; memcpy(w + 3, id, l + 1); w += 3 + l + 1;
      }
      if (b->has_sr && b->has_sr[i]) { w[0] = 's'; w[1] = 'r'; w[2] = 'C'; w[3] = 1; w += 4; }
    }
    at += sz;
  }
  if (off_out) off_out[b->n] = at;
  return at;
}
