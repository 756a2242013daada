// tests/deflate_host.cpp — host emulation of the device's DEFLATE block compressor (elprep_amd/csrc/deflate_core.hpp): the same
// functions the kernel k_bgzf_deflate runs, its 256 threads executed on one CPU thread, phase by phase, in index order (order 0) or in reverse
// (order 1) inside every phase.
// Test infrastructure: built by tests/test_deflate_cpu.py with g++, never loaded by the product.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../elprep_amd/csrc/deflate_core.hpp"

using namespace elp::dfl;

// in: n <= PAYLOAD bytes; out: the block's DEFLATE data (cap >= n + 5 + 8).  Returns its size; *stored = 1 if the block was stored.
// order 0: the threads of a strip / the parts run in index order; 1: in reverse order (another legal schedule: the result must inflate too)
extern "C" uint32_t dfl_emulate_block(const uint8_t *in_bytes, uint32_t n, uint8_t *out, int order, int *stored, uint32_t *n_tokens) {
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
  // 2. parse
  uint32_t bits[NT], ntok[NT], lo[NT], hi[NT];
  for (int k = 0; k < NT; k++) {
    const int p = tid(k);
    lo[p] = (uint32_t)p * PART < n ? (uint32_t)p * PART : n;
    hi[p] = lo[p] + PART < n ? lo[p] + PART : n;
    ntok[p] = parse_part(in.data(), ld.data(), lo[p], hi[p], &bits[p]);
  }
  unsigned long long total = 0;
  uint32_t off[NT], nt = 0;
  for (int p = 0; p < NT; p++) { off[p] = (uint32_t)total; total += bits[p]; nt += ntok[p]; }
  if (n_tokens) *n_tokens = nt;
  const uint32_t cbytes = deflate_bytes(total);
  if (cbytes >= n + 5u) {
    *stored = 1;
    out[0] = 0x01; out[1] = (uint8_t)(n & 0xFF); out[2] = (uint8_t)(n >> 8); out[3] = (uint8_t)(~n & 0xFF); out[4] = (uint8_t)((~n >> 8) & 0xFF);
    memcpy(out + 5, in_bytes, n);
    return n + 5u;
  }
  *stored = 0;
  std::vector<uint32_t> words((cbytes + 3) / 4 + 2, 0u);
  auto orw = [&](uint32_t w, uint32_t v) { words[w] |= v; };
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
