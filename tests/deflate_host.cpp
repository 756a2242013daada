// tests/deflate_host.cpp — host emulation of the device's DEFLATE block compressor (elprep_amd/csrc/deflate_core.hpp): the same
// functions the kernel k_bgzf_deflate runs, its 256 threads executed on one CPU thread, phase by phase, in index order (order 0) or in reverse
// (order 1) inside every phase.
// Test infrastructure: built by tests/test_deflate_cpu.py with g++, never loaded by the product.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../elprep_amd/csrc/deflate_core.hpp"

using namespace elp::dfl;

// in: n <= PAYLOAD bytes; out: the block's DEFLATE data (cap >= n + 5 + 8).  Returns its size; *stored = 1 if the block was stored, 2 if it
// has dynamic codes, 0 fixed codes.
// order 0: the threads of a strip / the parts run in index order; 1: in reverse order (another legal schedule: the result must inflate too)
extern "C" uint32_t dfl_emulate_block(const uint8_t *in_bytes, uint32_t n, uint8_t *out, int order, int *stored, uint32_t *n_tokens) {
  const int mode = order >> 4;  // 0: dynamic codes where they are shorter (the product), 1: fixed codes only
  order &= 15;
  std::vector<uint8_t> in(n + IN_PAD, 0);
  memcpy(in.data(), in_bytes, n);
  std::vector<uint16_t> table((size_t)WAYS << HBITS, NOPOS);
  std::vector<uint32_t> ld(n + NT, 0);
  auto tid = [&](int k) { return order ? NT - 1 - k : k; };
  // 1. match finding, strip by strip: look-ups of the whole strip, then its inserts (the device has a barrier between the two)
  for (uint32_t base = 0; base < n; base += NT) {
    for (int k = 0; k < NT; k++) { const uint32_t i = base + (uint32_t)tid(k); if (i < n) ld[i] = find_match(in.data(), n, i, table.data()); }
    for (int k = 0; k < NT; k++) { const uint32_t i = base + (uint32_t)tid(k); if (i < n) table_insert(table.data(), in.data(), n, i); }
  }
  // 2. parse (the symbols are counted on the way: the dynamic codes' histogram)
  static DynCodes D;
  for (int k = 0; k < 320; k++) D.freq[k] = 0;
  uint32_t bits[NT], ntok[NT], lo[NT], hi[NT];
  for (int k = 0; k < NT; k++) {
    const int p = tid(k);
    lo[p] = (uint32_t)p * PART < n ? (uint32_t)p * PART : n;
    hi[p] = lo[p] + PART < n ? lo[p] + PART : n;
    ntok[p] = parse_part(in.data(), ld.data(), lo[p], hi[p], &bits[p], [&](uint32_t s) { D.freq[s]++; });
  }
  unsigned long long total = 0;
  uint32_t off[NT], nt = 0;
  for (int p = 0; p < NT; p++) { off[p] = (uint32_t)total; total += bits[p]; nt += ntok[p]; }
  if (n_tokens) *n_tokens = nt;
  // 2b. dynamic codes (the device's phases: count | rank | build (one thread per alphabet) | codes | header | bits)
  D.freq[256]++;
  D.m[0] = D.m[1] = 0;
  for (int k = 0; k < 2 * NT; k++) {  // (two symbols per thread)
    const int s = tid(k % NT) + NT * (k / NT);
    if (s < NLL) { const int r = symbol_rank(D.freq, NLL, s); if (r >= 0) D.order[r] = (uint16_t)s; }
    if (s < NDIST) { const int r = symbol_rank(D.freq + DOFF, NDIST, s); if (r >= 0) D.order[DOFF + r] = (uint16_t)s; }
  }
  for (int s = 0; s < NLL; s++) D.m[0] += D.freq[s] != 0;
  for (int s = 0; s < NDIST; s++) D.m[1] += D.freq[DOFF + s] != 0;
  for (int k = 0; k < 320; k++) D.len[k] = 0;
  for (int a = 0; a < 2; a++) {  // a: 0 literal / length, 1 distance - the device's phases merge | depths | limit | lengths
    const int at = a ? DOFF : 0, n_sym = a ? NDIST : NLL, m = (int)D.m[a];
    for (int l = 0; l <= 16; l++) D.cnt[a][l] = 0;
    D.over[a] = 0;
    if (m < 2) { trivial_lengths(D.order + at, m, n_sym, D.len + at, D.cnt[a], D.base[a]); continue; }
    huffman_merge(D.freq + at, D.order + at, m, D.w + 2 * at, D.up + 2 * at);
    for (int k = 0; k < 2 * NT; k++) {
      const int leaf = tid(k % NT) + NT * (k / NT);
      if (leaf >= m) continue;
      uint32_t d = leaf_depth(D.up + 2 * at, leaf, m);
      if (d > 15u) { d = 15u; D.over[a]++; }
      D.cnt[a][d]++;
    }
    limit_counts(D.cnt[a], 15, (int)D.over[a], D.base[a]);
    for (int k = 0; k < 2 * NT; k++) {
      const int leaf = tid(k % NT) + NT * (k / NT);
      if (leaf < m) D.len[at + D.order[at + leaf]] = (uint8_t)length_of_rank(D.cnt[a], 15, leaf);
    }
  }
  for (int k = 0; k < 2 * NT; k++) {
    const int s = tid(k % NT) + NT * (k / NT);
    if (s < NLL) D.code[s] = canonical_code(D.len, s, D.base[0]);
    if (s < NDIST) D.code[DOFF + s] = canonical_code(D.len + DOFF, s, D.base[1]);
  }
  build_header(D);
  uint32_t dbits[NT], doff[NT];
  unsigned long long dtotal = 0;
  for (int k = 0; k < NT; k++) {
    const int p = tid(k);
    dbits[p] = 0;
    for (uint32_t j = 0; j < ntok[p]; j++) dbits[p] += token_bits_dyn(ld[lo[p] + j], D);
  }
  for (int p = 0; p < NT; p++) { doff[p] = (uint32_t)dtotal; dtotal += dbits[p]; }
  const uint32_t dyn_bytes = deflate_bytes_dyn(D, dtotal);
  const bool dynamic = mode != 1 && dyn_bytes < deflate_bytes(total);
  const uint32_t cbytes = dynamic ? dyn_bytes : deflate_bytes(total);
  if (cbytes >= n + 5u) {
    *stored = 1;
    out[0] = 0x01; out[1] = (uint8_t)(n & 0xFF); out[2] = (uint8_t)(n >> 8); out[3] = (uint8_t)(~n & 0xFF); out[4] = (uint8_t)((~n >> 8) & 0xFF);
    memcpy(out + 5, in_bytes, n);
    return n + 5u;
  }
  *stored = 0;
  std::vector<uint32_t> words((cbytes + 3) / 4 + 2, 0u);
  auto orw = [&](uint32_t w, uint32_t v) { words[w] |= v; };
  if (dynamic) {
    *stored = 2;
    {
      BitWriter<decltype(orw)> bw(orw, 0);
      emit_dyn_header(bw, D);
      bw.finish();
    }
    for (int k = 0; k < NT; k++) {
      const int p = tid(k);
      BitWriter<decltype(orw)> bw(orw, D.header_bits + doff[p]);
      for (uint32_t j = 0; j < ntok[p]; j++) emit_token_dyn(bw, ld[lo[p] + j], D);
      bw.finish();
    }
    {
      BitWriter<decltype(orw)> bw(orw, D.header_bits + (uint32_t)dtotal);
      bw.put(D.code[256], D.len[256]);
      bw.finish();
    }
    memcpy(out, words.data(), cbytes);
    return cbytes;
  }
  {  // BFINAL = 1, BTYPE = 01
    BitWriter<decltype(orw)> bw(orw, 0);
    bw.put(3u, 3u);
    bw.finish();
  }
  for (int k = 0; k < NT; k++) {
    const int p = tid(k);
    BitWriter<decltype(orw)> bw(orw, 3u + off[p]);
    for (uint32_t j = 0; j < ntok[p]; j++) emit_token(bw, ld[lo[p] + j]);
    bw.finish();
  }
  // (the end-of-block code is seven zero bits: counted in cbytes, nothing to set)
  memcpy(out, words.data(), cbytes);
  return cbytes;
}

// code lengths of an alphabet of n <= 320 symbols with the counts freq[], at most maxbits long, by the functions the kernel uses (rank by
// counting, merge, depths, limit, lengths by rank): for the tests' comparison with a textbook Huffman construction
extern "C" void dfl_code_lengths(const uint32_t *freq, int n, int maxbits, uint8_t *len_out) {
  uint16_t order[320], up[2 * 320], base[16];
  uint32_t w[2 * 320], count[17];
  int m = 0;
  for (int s = 0; s < n; s++) {
    len_out[s] = 0;
    const int r = symbol_rank(freq, n, s);
    if (r >= 0) { order[r] = (uint16_t)s; m++; }
  }
  huffman_lengths_serial(freq, order, m, n, maxbits, len_out, w, up, count, base);
}

// the dynamic block header for given code lengths (286 literal / length + 30 distance, each a complete code or the two-codes-of-one-bit
// form), written LSB first into out (cap bytes); returns the header's bits (the 3 block-header bits included)
extern "C" uint32_t dfl_header_bits(const uint8_t *len_ll, const uint8_t *len_d, uint8_t *out, uint32_t cap) {
  static DynCodes D;
  for (int s = 0; s < 320; s++) D.len[s] = 0;
  for (int s = 0; s < NLL; s++) D.len[s] = len_ll[s];
  for (int s = 0; s < NDIST; s++) D.len[DOFF + s] = len_d[s];
  build_header(D);
  std::vector<uint32_t> words(cap / 4 + 2, 0u);
  auto orw = [&](uint32_t w, uint32_t v) { words[w] |= v; };
  BitWriter<decltype(orw)> bw(orw, 0);
  emit_dyn_header(bw, D);
  bw.finish();
  memcpy(out, words.data(), cap);
  return D.header_bits;
}
