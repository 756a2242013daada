// deflate_core.hpp — the compressing half of the BGZF writer: one BGZF block = one DEFLATE block over an LZ77 parse made by 256 parts in
// parallel, with the block's OWN Huffman codes (RFC 1951 3.2.7, BTYPE 10: round 6, the second half of this file) or - where those are not
// shorter - the FIXED ones (3.2.6, BTYPE 01: round 5).
//
// Reference: utils/bgzf/bgzf-files.go:324-383 (Writer.Write / writeBlock: flate.NewWriter at level -1, 65280 payload bytes per block,
// CRC-32 + ISIZE trailer).  The reference's bytes are whatever Go's compress/flate emits; any valid DEFLATE stream that inflates to the
// same payload is the same BAM file to every reader, so what is checked here is that zlib inflates every member to exactly the record
// stream (tests/test_deflate_cpu.py on the host emulation below, tests/test_gpu_round5.py on the device).
//
// Shape (why it is not zlib's loop).  zlib finds matches and parses in ONE sequential loop over the block; here the two are separate and
// both are parallel over the block's 65280 bytes, 256 threads per block:
//   1. MATCH FINDING, a strip of 256 consecutive positions at a time, one position per thread: the position's 4-byte hash is looked up
//      in a two-way hash table in LDS that holds EVERY position in front of the strip (so a look-up sees the whole history of the block
//      up to the strip: a first form - every thread parsing its own part against a table all threads fill in lockstep - saw half of
//      it and compressed the BAM records to 0.72 instead of 0.55), the candidates and the byte in front of the position (runs: the
//      quality strings) are compared, the longest match (length, distance) is noted per position; then the strip's positions go into
//      the table.  The table is read and written without ordering inside a strip: whatever a look-up returns is only a CANDIDATE - it
//      counts only if it lies in front of the position, inside the 32 KB window, and its bytes match - so a lost or torn entry costs
//      compression, never correctness.
//   2. PARSE: the block is cut into 256 parts of 255 bytes; thread p walks part p greedily over the noted matches (a match is cut at
//      the part's end), leaves its tokens in place of the notes it has consumed, and counts their bits.
//   3. a scan gives every part its bit offset; every thread writes its tokens' codes at their offsets (bit-OR into the zeroed output:
//      neighbouring parts share words).  A block whose code would not be shorter than its payload is stored (BTYPE 00).
//
// Everything that decides a bit is in this header as host-and-device code: the device kernel (bgzf.hip) and the host emulation the CPU
// tests compile (tests/deflate_host.cpp) run the same functions in the same order of phases.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define ELP_DFL_HD __host__ __device__ __forceinline__
#else
#define ELP_DFL_HD inline
#endif

namespace elp {
namespace dfl {

#ifndef ELP_DFL_NT
#define ELP_DFL_NT 256
#endif
constexpr int NT = ELP_DFL_NT;        // parts (threads) per block
constexpr uint32_t PAYLOAD = 65280;   // BGZF payload bytes per block (bgzf-files.go:33)
constexpr uint32_t PART = (PAYLOAD + NT - 1) / NT;  // 255 bytes per part with 256 parts
#ifndef ELP_DFL_HBITS
#define ELP_DFL_HBITS 12
#endif
#ifndef ELP_DFL_WAYS
#define ELP_DFL_WAYS 2
#endif
constexpr int HBITS = ELP_DFL_HBITS, WAYS = ELP_DFL_WAYS;   // hash table: 4096 buckets of two u16 positions (16 KB of LDS: two workgroups per CU; round 5's
                                                             // 8192 buckets compress the bench's records to 0.4198 instead of 0.4220 and leave room for one)
#ifndef ELP_DFL_MAXL
#define ELP_DFL_MAXL 255
#endif
constexpr uint32_t MINM = 4, MAXM = 258, WINDOW = 32768, MAXL = ELP_DFL_MAXL;  // MAXL: the longest match the finder reports
constexpr uint16_t NOPOS = 0xFFFF;
constexpr uint32_t IN_PAD = 16;       // readable bytes behind the payload in the input buffer (4-byte loads at its end)

ELP_DFL_HD uint32_t load4(const uint8_t *p) {
#if defined(__HIP_DEVICE_COMPILE__)
  // (the payload lives in LDS, 16-byte aligned and padded: the two aligned words around the four bytes in ONE ds_read2_b32 and a funnel
  // shift, instead of four byte reads and three shift-ors - the match finder is this function, 2/3 of the kernel's time)
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  const uint32_t *w = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
  return __builtin_amdgcn_alignbit(w[1], w[0], (uint32_t)(a & 3u) * 8u);
#else
  return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
#endif
}
ELP_DFL_HD uint32_t hash4(uint32_t w) { return (w * 2654435761u) >> (32 - HBITS); }
ELP_DFL_HD uint32_t ilog2(uint32_t x) { return 31u - (uint32_t)__builtin_clz(x); }  // x > 0

// token: a literal byte, or 0x80000000 | (length - 3) << 16 | (distance - 1)
ELP_DFL_HD uint32_t tok_match(uint32_t len, uint32_t dist) { return 0x80000000u | ((len - 3u) << 16) | (dist - 1u); }

// length 3..258 -> literal/length symbol 257..285, number and value of its extra bits (RFC 1951 3.2.5)
ELP_DFL_HD uint32_t len_symbol(uint32_t len, uint32_t &eb, uint32_t &ev) {
  const uint32_t l = len - 3u;
  if (len == 258u) { eb = 0; ev = 0; return 285u; }
  if (l < 8u) { eb = 0; ev = 0; return 257u + l; }
  eb = ilog2(l) - 2u;
  ev = l & ((1u << eb) - 1u);
  return 257u + 4u * (eb + 1u) + ((l >> eb) & 3u);
}
// distance 1..32768 -> distance code 0..29, number and value of its extra bits
ELP_DFL_HD uint32_t dist_symbol(uint32_t dist, uint32_t &eb, uint32_t &ev) {
  const uint32_t d = dist - 1u;
  if (d < 4u) { eb = 0; ev = 0; return d; }
  const uint32_t n = ilog2(d);
  eb = n - 1u;
  ev = d & ((1u << eb) - 1u);
  return 2u * n + ((d >> eb) & 1u);
}
// fixed Huffman code of a literal/length symbol (RFC 1951 3.2.6): code value (MSB first) and length
ELP_DFL_HD uint32_t fixed_code(uint32_t sym, uint32_t &nbits) {
  if (sym < 144u) { nbits = 8; return 0x30u + sym; }
  if (sym < 256u) { nbits = 9; return 0x190u + (sym - 144u); }
  if (sym < 280u) { nbits = 7; return sym - 256u; }
  nbits = 8;
  return 0xC0u + (sym - 280u);
}
// Huffman codes go into the stream starting with their most significant bit, everything else LSB first (RFC 1951 3.1.1)
ELP_DFL_HD uint32_t bit_reverse(uint32_t v, uint32_t nbits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __brev(v) >> (32u - nbits);
#endif
  uint32_t r = 0;
  for (uint32_t k = 0; k < nbits; k++) r |= ((v >> k) & 1u) << (nbits - 1u - k);
  return r;
}
ELP_DFL_HD uint32_t token_bits(uint32_t tok) {
  if (!(tok & 0x80000000u)) return tok < 144u ? 8u : 9u;
  uint32_t eb, ev, nb, deb, dev;
  const uint32_t sym = len_symbol(((tok >> 16) & 0xFFu) + 3u, eb, ev);
  (void)fixed_code(sym, nb);
  (void)dist_symbol((tok & 0x7FFFu) + 1u, deb, dev);
  return nb + eb + 5u + deb;
}

// ---- 1. match finding.  table: [1 << HBITS][WAYS] positions (NOPOS = empty), way 0 the most recent
// equal bytes at c.. and i.., at most maxl
ELP_DFL_HD uint32_t match_len(const uint8_t *in, uint32_t c, uint32_t i, uint32_t maxl) {
  uint32_t l = 0;
#if defined(__HIP_DEVICE_COMPILE__)
  // four bytes a step to the end (the bytes behind the payload are padding: reading them is safe, counting them is not)
  for (;;) {
    const uint32_t x = load4(in + c + l) ^ load4(in + i + l);
    const uint32_t eq = x ? (uint32_t)__builtin_ctz(x) >> 3 : 4u, left = maxl - l;
    if (eq < 4u || left <= 4u) return l + (eq < left ? eq : left);
    l += 4u;
  }
#else
  while (l + 4u <= maxl && load4(in + c + l) == load4(in + i + l)) l += 4u;
  while (l < maxl && in[c + l] == in[i + l]) l++;
  return l;
#endif
}
// the note of position i: length (0 = no match of MINM bytes or more; at most 255) | distance << 8
#if defined(__HIPCC__) && ELP_DFL_WAYS == 2
// the device's form of the function below, same result: the first words of the three candidates (two ways of the bucket, the byte in front)
// are read TOGETHER - one LDS round trip instead of three, one after the other - and only candidates whose first four bytes match go on
ELP_DFL_HD uint32_t match_rest(const uint8_t *in, uint32_t c, uint32_t i, uint32_t maxl, uint32_t x /* first words xor-ed */) {
  const uint32_t eq = x ? (uint32_t)__builtin_ctz(x) >> 3 : 4u;
  if (eq < 4u || maxl <= 4u) return eq < maxl ? eq : maxl;
  return 4u + match_len(in, c + 4u, i + 4u, maxl - 4u);
}
ELP_DFL_HD uint32_t find_match(const uint8_t *in, uint32_t n, uint32_t i, const uint16_t *table, uint32_t w /* load4(in + i) */) {
  const uint32_t maxl = n - i < MAXL ? n - i : MAXL;
  if (maxl < MINM) return 0u;
  const uint32_t both = *reinterpret_cast<const uint32_t *>(table + hash4(w) * 2u);
  const uint32_t c0 = both & 0xFFFFu, c1 = both >> 16;
  const bool ok0 = c0 < i && i - c0 <= WINDOW, ok1 = c1 < i && i - c1 <= WINDOW, okr = i > 0;
  const uint32_t x0 = ok0 ? load4(in + c0) ^ w : 1u, x1 = ok1 ? load4(in + c1) ^ w : 1u, xr = okr ? load4(in + i - 1u) ^ w : 1u;
  uint32_t best = 0, bd = 0;
  if (ok0) { const uint32_t l = match_rest(in, c0, i, maxl, x0); if (l > best) { best = l; bd = i - c0; } }
  if (ok1) { const uint32_t l = match_rest(in, c1, i, maxl, x1); if (l > best) { best = l; bd = i - c1; } }
  if (okr) { const uint32_t l = match_rest(in, i - 1u, i, maxl, xr); if (l > best) { best = l; bd = 1u; } }
  return best >= MINM ? (best | (bd << 8)) : 0u;
}
// (the position's first four bytes are read once for the look-up and the insert)
ELP_DFL_HD void table_insert(uint16_t *table, uint32_t n, uint32_t i, uint32_t w /* load4(in + i) */) {
  if (i + MINM > n) return;
  uint32_t *e = reinterpret_cast<uint32_t *>(table + hash4(w) * 2u);
  *e = (*e << 16) | i;  // (way 0 moves to way 1; unordered against the other threads of the strip, like the two stores of the host's form)
}
#else
ELP_DFL_HD uint32_t find_match(const uint8_t *in, uint32_t n, uint32_t i, const uint16_t *table) {
  const uint32_t maxl = n - i < MAXL ? n - i : MAXL;
  uint32_t best = 0, bd = 0;
  if (maxl >= MINM) {
    const uint32_t h = hash4(load4(in + i));
    for (int k = 0; k < WAYS; k++) {
      const uint32_t c = table[h * WAYS + k];
      if (c < i && i - c <= WINDOW) {
        const uint32_t l = match_len(in, c, i, maxl);
        if (l > best) { best = l; bd = i - c; }
      }
    }
    if (i > 0) {  // the byte in front: a run
      const uint32_t l = match_len(in, i - 1u, i, maxl);
      if (l > best) { best = l; bd = 1u; }
    }
  }
  return best >= MINM ? (best | (bd << 8)) : 0u;
}
#endif
ELP_DFL_HD void table_insert(uint16_t *table, const uint8_t *in, uint32_t n, uint32_t i) {
  if (i + MINM > n) return;
  const uint32_t h = hash4(load4(in + i));
  table[h * WAYS + 1] = table[h * WAYS];
  table[h * WAYS] = (uint16_t)i;
}

// ---- 2. the greedy parse of part [lo, hi): reads the notes ld[i], writes token k of the part to ld[lo + k] (k <= i - lo: a note is
// overwritten only after it has been read); returns the number of tokens, *bits = their size in a fixed-Huffman block
// count(symbol index): called for the token's literal / length symbol s (index s) and distance symbol d (index DOFF_ + d) - the dynamic
// codes' histogram, taken while the token is at hand
template <class Count>
ELP_DFL_HD uint32_t parse_part(const uint8_t *in, uint32_t *ld, uint32_t lo, uint32_t hi, uint32_t *bits, Count count) {
  uint32_t i = lo, k = 0, nb = 0;
  while (i < hi) {
    const uint32_t note = ld[i];
    uint32_t len = note & 0xFFu;
    if (len > hi - i) len = hi - i;
    uint32_t tok;
    if (len >= MINM) {
      tok = tok_match(len, note >> 8);
      i += len;
      uint32_t eb, ev, deb, dev, nbl;
      const uint32_t sym = len_symbol(len, eb, ev);
      (void)fixed_code(sym, nbl);
      const uint32_t ds = dist_symbol(note >> 8, deb, dev);
      nb += nbl + eb + 5u + deb;
      count(sym);
      count(288u + ds);
    } else {
      tok = in[i];
      i += 1;
      nb += tok < 144u ? 8u : 9u;
      count(tok);
    }
    ld[lo + k++] = tok;
  }
  *bits = nb;
  return k;
}

// Bit writer into zeroed 32-bit words shared with the neighbouring parts: words go out through orw(word index, value) (an atomic OR on
// the device)
template <class OrW>
struct BitWriter {
  OrW orw;
  uint32_t w, fill;
  unsigned long long acc;
  ELP_DFL_HD BitWriter(OrW o, uint32_t bit_offset) : orw(o), w(bit_offset >> 5), fill(bit_offset & 31u), acc(0) {}
  ELP_DFL_HD void put(uint32_t v, uint32_t nbits) {  // nbits <= 16
    acc |= (unsigned long long)v << fill;
    fill += nbits;
    if (fill >= 32u) { orw(w++, (uint32_t)acc); acc >>= 32; fill -= 32u; }
  }
  ELP_DFL_HD void finish() { if (fill) orw(w, (uint32_t)acc); }
};
template <class OrW>
ELP_DFL_HD void emit_token(BitWriter<OrW> &bw, uint32_t tok) {
  uint32_t nb;
  if (!(tok & 0x80000000u)) {
    const uint32_t c = fixed_code(tok, nb);
    bw.put(bit_reverse(c, nb), nb);
    return;
  }
  uint32_t eb, ev, deb, dev;
  const uint32_t sym = len_symbol(((tok >> 16) & 0xFFu) + 3u, eb, ev);
  const uint32_t c = fixed_code(sym, nb);
  bw.put(bit_reverse(c, nb), nb);
  if (eb) bw.put(ev, eb);
  const uint32_t ds = dist_symbol((tok & 0x7FFFu) + 1u, deb, dev);
  bw.put(bit_reverse(ds, 5u), 5u);
  if (deb) bw.put(dev, deb);
}

// ------------------------------------------------------------------ dynamic Huffman codes (round 6; RFC 1951 3.2.7, BTYPE 10)
// The reference's writer is compress/flate at its default level (utils/bgzf/bgzf-files.go:324-383): dynamic codes per block.  Here, behind
// the parse: the tokens' symbols are counted (286 literal / length symbols, 30 distance symbols), both alphabets get Huffman code lengths
// (<= 15 bits), the lengths are run-length coded and themselves Huffman coded (<= 7 bits) into the block's header, and the parts write their
// tokens with the block's own codes.  A block whose dynamic form is not shorter keeps the fixed codes.
// Who does what on the device (256 threads; the host emulation runs the same functions in the same phases):
//   count        every part over its tokens (atomic adds)
//   rank         every thread for its symbols: the symbol's place in the order by (count, index) - a sort by counting, 286 reads each
//   merge        ONE thread per alphabet: the two-queue Huffman merge over the sorted counts (the one serial loop: ~2 LDS round trips a node)
//   depths       every thread for its leaves: steps to the root, counted per depth (atomic adds)
//   limit        one thread per alphabet: the counts repaired where the tree is deeper than 15; canonical bases
//   lengths      every thread for its leaves: the length of its rank; then the canonical code = base of its length + rank among the
//                symbols of that length
//   header       one thread: run-length symbols 16 / 17 / 18, their counts, their 7-bit-limited code, the header's bits
//   bits, write  every part, as with the fixed codes
constexpr int NLL = 286, NDIST = 30, NCL = 19, DOFF = 288;  // the distance alphabet's arrays start at DOFF
struct DynCodes {
  uint32_t freq[320];    // [0, 286) literal / length, [DOFF, DOFF + 30) distance
  uint16_t order[320];   // the used symbols in ascending order of (count, index), per alphabet
  uint8_t len[320];      // code lengths (0: unused)
  uint16_t code[320];    // the codes, bit-reversed (ready for the LSB-first stream)
  uint32_t w[2 * 320];   // weights of the merge's nodes (leaves, then internal nodes), per alphabet at 2 * its offset
  uint16_t up[2 * 320];  // parent of a node
  uint16_t base[2][16];  // canonical base code per length
  uint32_t cnt[2][17];   // codes per length
  uint32_t over[2];      // leaves deeper than the limit
  uint16_t rle[320];     // the header's code-length symbols: symbol | extra bits' value << 8
  uint32_t n_rle, hlit, hdist, hclen, header_bits;
  uint32_t m[2];         // used symbols per alphabet
  uint8_t cl_len[NCL];
  uint16_t cl_code[NCL];
  // the code of the code lengths is built by one thread: its work arrays (here, not on the thread's stack: on the device that is HBM)
  uint32_t cl_freq[NCL], cl_w[2 * NCL], cl_cnt[17];
  uint16_t cl_ord[NCL], cl_up[2 * NCL], cl_base[16];
};
// the literal / length symbol and the distance symbol of a token (distance symbol: 0xFFFF for a literal)
ELP_DFL_HD uint32_t token_symbols(uint32_t tok, uint32_t &dsym, uint32_t &extra_bits) {
  if (!(tok & 0x80000000u)) { dsym = 0xFFFFu; extra_bits = 0; return tok; }
  uint32_t eb, ev, deb, dev;
  const uint32_t sym = len_symbol(((tok >> 16) & 0xFFu) + 3u, eb, ev);
  dsym = dist_symbol((tok & 0x7FFFu) + 1u, deb, dev);
  extra_bits = eb + deb;
  return sym;
}
// place of symbol s among the used symbols of freq[0, n) in the order by (count, index); -1: unused
ELP_DFL_HD int symbol_rank(const uint32_t *freq, int n, int s) {
  const uint32_t f = freq[s];
  if (!f) return -1;
  int r = 0;
  for (int j = 0; j < n; j++) {
    const uint32_t g = freq[j];
    r += (g != 0u && (g < f || (g == f && j < s))) ? 1 : 0;
  }
  return r;
}
// Huffman code lengths of the m used symbols order[0, m) (ascending counts) of an alphabet, at most maxbits long, in pieces: the merge is
// serial (one thread), the depths of the leaves and the lengths by rank are not.
// 1. the merge (one thread; m >= 2).  Leaves 0 .. m-1 in ascending weight, internal nodes m .. 2m-2 in the order they are made (ascending
//    too): two queues, their heads in registers.  up[node] = its parent; the root is node 2m-2.  w: room for 2 m weights.
ELP_DFL_HD void huffman_merge(const uint32_t *freq, const uint16_t *order, int m, uint32_t *w, uint16_t *up) {
  for (int k = 0; k < m; k++) w[k] = freq[order[k]];
  const uint32_t NONE = 0xFFFFFFFFu;
  int leaf = 0, node = m;
  uint32_t wl = w[0], wn = NONE;  // weights at the heads (NONE: the queue is empty)
  for (int made = m; made < 2 * m - 1; made++) {
    uint32_t sum = 0;
    for (int pick = 0; pick < 2; pick++) {
      if (wl <= wn) { sum += wl; up[leaf] = (uint16_t)made; leaf++; wl = leaf < m ? w[leaf] : NONE; }
      else { sum += wn; up[node] = (uint16_t)made; node++; wn = node < made ? w[node] : NONE; }
    }
    w[made] = sum;
    if (wn == NONE && node == made) wn = sum;  // (the node just made is the internal queue's new head)
  }
}
// 2. the depth of leaf k (any thread)
ELP_DFL_HD uint32_t leaf_depth(const uint16_t *up, int k, int m) {
  uint32_t d = 0;
  for (int node = k; node != 2 * m - 2; node = up[node]) d++;
  return d;
}
// 3. count[d] = leaves of depth d with the depths beyond the limit clamped to it (one thread).  The clamped code is over-subscribed by
//    `excess` units of 2^-maxbits (less than one unit per clamped leaf); one step of the classic repair (as in zlib's gen_bitlen) takes
//    exactly one unit away and keeps the number of leaves: a leaf of the deepest level above the limit's that has one moves a level down
//    and takes a leaf of the limit's level as its sibling.  (`overflow` = the number of clamped leaves: 0 means nothing to do.)
//    Then base[L] = first canonical code of length L (RFC 1951 3.2.2).
ELP_DFL_HD void limit_counts(uint32_t *count, int maxbits, int overflow, uint16_t *base) {
  if (overflow > 0) {
    unsigned long long kraft = 0;
    for (int l = 1; l <= maxbits; l++) kraft += (unsigned long long)count[l] << (maxbits - l);
    for (unsigned long long excess = kraft - (1ull << maxbits); excess > 0; excess--) {
      int bits = maxbits - 1;
      while (count[bits] == 0) bits--;
      count[bits]--;
      count[bits + 1] += 2;
      count[maxbits]--;
    }
  }
  uint32_t code = 0;
  base[0] = 0;
  for (int l = 1; l < 16; l++) { code = (code + (l > 1 ? count[l - 1] : 0u)) << 1; base[l] = (uint16_t)code; }
}
// 4. the length of the leaf of rank k: the rarest symbols get the longest codes (any thread)
ELP_DFL_HD uint32_t length_of_rank(const uint32_t *count, int maxbits, int k) {
  uint32_t acc = 0;
  for (int bits = maxbits; bits > 1; bits--) {
    acc += count[bits];
    if ((uint32_t)k < acc) return (uint32_t)bits;
  }
  return 1u;
}
// fewer than two used symbols (one thread): two codes of one bit - a complete code every decoder accepts
ELP_DFL_HD void trivial_lengths(const uint16_t *order, int m, int n, uint8_t *len, uint32_t *count, uint16_t *base) {
  const int s = m ? order[0] : 0;
  len[s] = 1;
  len[s == 0 ? (n > 1 ? 1 : 0) : 0] = 1;
  for (int l = 0; l <= 16; l++) count[l] = 0;
  count[1] = 2;
  limit_counts(count, 15, 0, base);
}
// all of it by one thread (the code of the code lengths; the tests' cross-check).  len[] zeroed by the caller; count: 17 words
ELP_DFL_HD void huffman_lengths_serial(const uint32_t *freq, const uint16_t *order, int m, int n, int maxbits, uint8_t *len, uint32_t *w, uint16_t *up,
                                       uint32_t *count, uint16_t *base) {
  if (m < 2) { trivial_lengths(order, m, n, len, count, base); return; }
  huffman_merge(freq, order, m, w, up);
  for (int l = 0; l <= 16; l++) count[l] = 0;
  int overflow = 0;
  for (int k = 0; k < m; k++) {
    uint32_t d = leaf_depth(up, k, m);
    if (d > (uint32_t)maxbits) { d = (uint32_t)maxbits; overflow++; }
    count[d]++;
  }
  limit_counts(count, maxbits, overflow, base);
  for (int k = 0; k < m; k++) len[order[k]] = (uint8_t)length_of_rank(count, maxbits, k);
}
// the code of symbol s, bit-reversed; any thread
ELP_DFL_HD uint16_t canonical_code(const uint8_t *len, int s, const uint16_t *base) {
  const uint32_t l = len[s];
  if (!l) return 0;
  uint32_t r = 0;
  for (int j = 0; j < s; j++) r += len[j] == l ? 1u : 0u;
  return (uint16_t)bit_reverse(base[l] + r, l);
}
// the header: HLIT, HDIST, the run-length coded code lengths, their code; D.header_bits = everything in front of the first token
// (3 block-header bits included).  One thread.
ELP_DFL_HD void build_header(DynCodes &D) {
  const uint8_t order19[NCL] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  int nll = NLL, nd = NDIST;
  while (nll > 257 && D.len[nll - 1] == 0) nll--;
  while (nd > 1 && D.len[DOFF + nd - 1] == 0) nd--;
  D.hlit = (uint32_t)nll - 257u;
  D.hdist = (uint32_t)nd - 1u;
  // the nll + nd lengths as one sequence, run-length coded (RFC 1951 3.2.7: 16 = repeat the previous 3..6 times, 17 = 3..10 zeros,
  // 18 = 11..138 zeros)
  uint32_t *cf = D.cl_freq;
  for (int k = 0; k < NCL; k++) cf[k] = 0;
  uint32_t nr = 0;
  const int total = nll + nd;
  int i = 0;
  while (i < total) {
    const uint32_t v = i < nll ? D.len[i] : D.len[DOFF + (i - nll)];
    int run = 1;
    while (i + run < total && (i + run < nll ? D.len[i + run] : D.len[DOFF + (i + run - nll)]) == v) run++;
    int left = run;
    if (v == 0) {
      while (left >= 11) { const int r = left > 138 ? 138 : left; D.rle[nr++] = (uint16_t)(18u | ((uint32_t)(r - 11) << 8)); cf[18]++; left -= r; }
      if (left >= 3) { D.rle[nr++] = (uint16_t)(17u | ((uint32_t)(left - 3) << 8)); cf[17]++; left = 0; }
      while (left > 0) { D.rle[nr++] = 0; cf[0]++; left--; }
    } else {
      D.rle[nr++] = (uint16_t)v; cf[v]++; left--;
      while (left >= 3) { const int r = left > 6 ? 6 : left; D.rle[nr++] = (uint16_t)(16u | ((uint32_t)(r - 3) << 8)); cf[16]++; left -= r; }
      while (left > 0) { D.rle[nr++] = (uint16_t)v; cf[v]++; left--; }
    }
    i += run;
  }
  D.n_rle = nr;
  // the code of the code lengths: 19 symbols, at most 7 bits (sorted here: the alphabet is tiny)
  uint16_t *ord = D.cl_ord;
  int m = 0;
  for (int s = 0; s < NCL; s++) if (cf[s]) ord[m++] = (uint16_t)s;
  for (int a = 1; a < m; a++) {
    const uint16_t v = ord[a];
    int b = a;
    for (; b > 0 && (cf[ord[b - 1]] > cf[v] || (cf[ord[b - 1]] == cf[v] && ord[b - 1] > v)); b--) ord[b] = ord[b - 1];
    ord[b] = v;
  }
  for (int s = 0; s < NCL; s++) D.cl_len[s] = 0;
  huffman_lengths_serial(cf, ord, m, NCL, 7, D.cl_len, D.cl_w, D.cl_up, D.cl_cnt, D.cl_base);
  for (int s = 0; s < NCL; s++) D.cl_code[s] = canonical_code(D.cl_len, s, D.cl_base);
  int ncl = NCL;
  while (ncl > 4 && D.cl_len[order19[ncl - 1]] == 0) ncl--;
  D.hclen = (uint32_t)ncl - 4u;
  uint32_t bits = 3u + 5u + 5u + 4u + 3u * (uint32_t)ncl;
  for (uint32_t k = 0; k < nr; k++) {
    const uint32_t sym = D.rle[k] & 0xFFu;
    bits += D.cl_len[sym] + (sym == 16u ? 2u : sym == 17u ? 3u : sym == 18u ? 7u : 0u);
  }
  D.header_bits = bits;
}
template <class OrW>
ELP_DFL_HD void emit_dyn_header(BitWriter<OrW> &bw, const DynCodes &D) {
  const uint8_t order19[NCL] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
  bw.put(5u, 3u);  // BFINAL = 1, BTYPE = 10
  bw.put(D.hlit, 5u);
  bw.put(D.hdist, 5u);
  bw.put(D.hclen, 4u);
  for (uint32_t k = 0; k < D.hclen + 4u; k++) bw.put(D.cl_len[order19[k]], 3u);
  for (uint32_t k = 0; k < D.n_rle; k++) {
    const uint32_t sym = D.rle[k] & 0xFFu, ev = D.rle[k] >> 8;
    bw.put(D.cl_code[sym], D.cl_len[sym]);
    if (sym == 16u) bw.put(ev, 2u);
    else if (sym == 17u) bw.put(ev, 3u);
    else if (sym == 18u) bw.put(ev, 7u);
  }
}
ELP_DFL_HD uint32_t token_bits_dyn(uint32_t tok, const DynCodes &D) {
  uint32_t ds, xb;
  const uint32_t sym = token_symbols(tok, ds, xb);
  return D.len[sym] + xb + (ds != 0xFFFFu ? D.len[DOFF + ds] : 0u);
}
template <class OrW>
ELP_DFL_HD void emit_token_dyn(BitWriter<OrW> &bw, uint32_t tok, const DynCodes &D) {
  if (!(tok & 0x80000000u)) { bw.put(D.code[tok], D.len[tok]); return; }
  uint32_t eb, ev, deb, dev;
  const uint32_t sym = len_symbol(((tok >> 16) & 0xFFu) + 3u, eb, ev);
  bw.put(D.code[sym], D.len[sym]);
  if (eb) bw.put(ev, eb);
  const uint32_t ds = dist_symbol((tok & 0x7FFFu) + 1u, deb, dev);
  bw.put(D.code[DOFF + ds], D.len[DOFF + ds]);
  if (deb) bw.put(dev, deb);
}
// size in bytes of a dynamic block: header, tokens, the end-of-block code
ELP_DFL_HD uint32_t deflate_bytes_dyn(const DynCodes &D, unsigned long long token_bits_total) {
  return (uint32_t)((D.header_bits + token_bits_total + D.len[256] + 7ull) >> 3);
}

// size in bytes of the block's DEFLATE data given the parts' bits: 3 header bits + the tokens + the 7-bit end-of-block code
ELP_DFL_HD uint32_t deflate_bytes(unsigned long long token_bits_total) { return (uint32_t)((3ull + token_bits_total + 7ull + 7ull) >> 3); }

}  // namespace dfl
}  // namespace elp
