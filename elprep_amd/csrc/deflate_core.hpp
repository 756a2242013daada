// deflate_core.hpp — the compressing half of the BGZF writer (round 5): one BGZF block = one DEFLATE block with FIXED Huffman codes
// (RFC 1951 3.2.6, BTYPE 01) over an LZ77 parse made by 256 parts in parallel.
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

constexpr int NT = 256;               // parts (threads) per block
constexpr uint32_t PAYLOAD = 65280;   // BGZF payload bytes per block (bgzf-files.go:33)
constexpr uint32_t PART = 255;        // PAYLOAD / NT
constexpr int HBITS = 13, WAYS = 2;   // hash table: 8192 buckets of two u16 positions (32 KB of LDS)
constexpr uint32_t MINM = 4, MAXM = 258, WINDOW = 32768;
constexpr uint16_t NOPOS = 0xFFFF;
constexpr uint32_t IN_PAD = 16;       // readable bytes behind the payload in the input buffer (4-byte loads at its end)

ELP_DFL_HD uint32_t load4(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
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
ELP_DFL_HD uint32_t match_len(const uint8_t *in, uint32_t c, uint32_t i, uint32_t maxl) {
  uint32_t l = 0;
  while (l + 4u <= maxl && load4(in + c + l) == load4(in + i + l)) l += 4u;
  while (l < maxl && in[c + l] == in[i + l]) l++;
  return l;
}
// the note of position i: length (0 = no match of MINM bytes or more; at most 255) | distance << 8
ELP_DFL_HD uint32_t find_match(const uint8_t *in, uint32_t n, uint32_t i, const uint16_t *table) {
  const uint32_t maxl = n - i < 255u ? n - i : 255u;
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
ELP_DFL_HD void table_insert(uint16_t *table, const uint8_t *in, uint32_t n, uint32_t i) {
  if (i + MINM > n) return;
  const uint32_t h = hash4(load4(in + i));
  table[h * WAYS + 1] = table[h * WAYS];
  table[h * WAYS] = (uint16_t)i;
}

// ---- 2. the greedy parse of part [lo, hi): reads the notes ld[i], writes token k of the part to ld[lo + k] (k <= i - lo: a note is
// overwritten only after it has been read); returns the number of tokens, *bits = their size in a fixed-Huffman block
ELP_DFL_HD uint32_t parse_part(const uint8_t *in, uint32_t *ld, uint32_t lo, uint32_t hi, uint32_t *bits) {
  uint32_t i = lo, k = 0, nb = 0;
  while (i < hi) {
    const uint32_t note = ld[i];
    uint32_t len = note & 0xFFu;
    if (len > hi - i) len = hi - i;
    uint32_t tok;
    if (len >= MINM) { tok = tok_match(len, note >> 8); i += len; }
    else { tok = in[i]; i += 1; }
    ld[lo + k++] = tok;
    nb += token_bits(tok);
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

// size in bytes of the block's DEFLATE data given the parts' bits: 3 header bits + the tokens + the 7-bit end-of-block code
ELP_DFL_HD uint32_t deflate_bytes(unsigned long long token_bits_total) { return (uint32_t)((3ull + token_bits_total + 7ull + 7ull) >> 3); }

}  // namespace dfl
}  // namespace elp
