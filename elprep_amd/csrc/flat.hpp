// flat.hpp — "flat" streaming over the ragged per-base columns (QUAL, with SEQ / skip bits / reference alongside it).
//
// The per-base kernels (Phred-sum score, BQSR gather, BQSR apply) do not assign a thread to a read.  A workgroup takes a
// 32 KiB tile of the QUAL byte column; every lane loads aligned 16-byte chunks (1 KiB per wave instruction, fully
// coalesced) and owns the 16 consecutive bases of a chunk.  Which reads overlap a tile comes from a tile index built once
// per staged column (k_flat_index: one binary search per tile, all tiles in parallel); the reads' QUAL offsets (relative to
// the tile) and whatever per-read data the kernel wants sit in LDS, and a lane finds the read of its chunk with a guess from
// the mean read length plus a short LDS walk.  A chunk that straddles read boundaries is seen as several *segments*.
//
// Everything per-base is done on whole chunks with SWAR arithmetic in "nibble space": a 64-bit word holds one 4-bit field
// per base of the chunk (base b of the chunk at bits 4b..4b+3), so masks, base codes, context keys and mismatch flags of
// 16 bases cost a handful of 64-bit ALU operations and no per-base memory access.
#pragma once
#include "common.hpp"

namespace elp {

constexpr int FL_THREADS = 512;                                               // default workgroup size (kernels use blockDim.x)
constexpr int FL_CHUNK = 16;
constexpr uint64_t FL_TILE = 32768;                                          // QUAL bytes per tile (2048 chunks)
constexpr int FL_RMAX = 384;                                                 // reads held in LDS at a time (150-base reads: ~220 per tile)
constexpr uint32_t FL_MAX_READ = 0x3FFFFFFFu;                                // per-read QUAL length limit of the tile-relative int32 offsets

struct FlatLds {
  int32_t off[FL_RMAX + 4];  // (n+1 used; sized to keep the dynamic-LDS base 16-byte aligned) QUAL offsets of the group's reads minus the tile's first byte (clamped; only in-tile values matter)
};

int ensure_flat_index(elp_ctx *c);  // builds c->tile_first for the staged QUAL column (sort.hip)

// ------------------------------------------------------------------ nibble-space helpers
constexpr uint64_t NIB1 = 0x1111111111111111ull;
constexpr uint64_t NIBF = 0xFFFFFFFFFFFFFFFFull;

// flag bit (bit 4b) for every b in [lo, hi); requires 0 <= lo < hi <= 16
__device__ __forceinline__ uint64_t nib_range(int lo, int hi) { return (NIB1 << (4 * lo)) & (NIBF >> (64 - 4 * hi)); }
// same, but tolerant: clamps to [0,16] and returns 0 for empty ranges
__device__ __forceinline__ uint64_t nib_range_clamped(int lo, int hi) {
  lo = lo < 0 ? 0 : lo;
  hi = hi > 16 ? 16 : hi;
  return lo < hi ? nib_range(lo, hi) : 0ull;
}
// flag bits (bit 4b) -> full nibbles (0xF)
__device__ __forceinline__ uint64_t nib_fill(uint64_t flags) { return flags * 15ull; }
// bit i of a 16-bit word -> bit 4i
__device__ __forceinline__ uint64_t nib_spread16(uint32_t x16) {
  uint64_t x = x16 & 0xFFFFu;
  x = (x | (x << 24)) & 0x000000FF000000FFull;
  x = (x | (x << 12)) & 0x000F000F000F000Full;
  x = (x | (x << 6)) & 0x0303030303030303ull;
  x = (x | (x << 3)) & NIB1;
  return x;
}
// swap the two nibbles of every byte (BAM packs the first base of a byte into the HIGH nibble)
__device__ __forceinline__ uint64_t nib_swap(uint64_t v) { return ((v & 0x0F0F0F0F0F0F0F0Full) << 4) | ((v >> 4) & 0x0F0F0F0F0F0F0F0Full); }
// nibble b of the result = nibble (b + sn) of the 128-bit value v1:v0, zero where b + sn < 0;  sn in [-16, 15]
__device__ __forceinline__ uint64_t nib_ext(uint64_t v0, uint64_t v1, int sn) {
  const int shr = 4 * (sn < 0 ? 0 : sn);          // 0..60
  const int shl = 4 * (sn < 0 ? -sn : 0);         // 0..64
  const uint64_t pos = (v0 >> shr) | ((v1 << 1) << (63 - shr));
  const uint64_t neg = shl >= 64 ? 0ull : (v0 << (shl & 63));
  return sn < 0 ? neg : pos;
}
// flag where the nibble is non-zero
__device__ __forceinline__ uint64_t nib_nonzero(uint64_t x) { return (x | (x >> 1) | (x >> 2) | (x >> 3)) & NIB1; }
// BAM base nibbles: onehot = flag where the nibble is A(1) C(2) G(4) T(8); code = 2-bit A0 C1 G2 T3 (valid where onehot)
__device__ __forceinline__ void nib_classify(uint64_t x, uint64_t &onehot, uint64_t &code) {
  const uint64_t a = x & NIB1, b = (x >> 1) & NIB1, c = (x >> 2) & NIB1, d = (x >> 3) & NIB1;
  onehot = (a ^ b ^ c ^ d) & ~((a & b) | (c & d));
  code = (b | d) | ((c | d) << 1);
}

// bases kb .. kb+15 of a record (S) and their neighbours kb+dir .. kb+15+dir (N) as nibbles; bases before the record's first
// read 0, bases past its end are garbage (callers mask).  `sp` = the record's packed bases (BAM order), kb in [-15, l_seq).
__device__ __forceinline__ void seq_nibbles(const uint8_t *__restrict__ sp, int kb, int dir, uint64_t &S, uint64_t &N) {
  int wb = kb - 1;
  wb = (wb < 0 ? 0 : wb) & ~1;
  uint64_t v0, v1;
  __builtin_memcpy(&v0, sp + (wb >> 1), 8);
  __builtin_memcpy(&v1, sp + (wb >> 1) + 8, 8);
  v0 = nib_swap(v0);
  v1 = nib_swap(v1);
  const int s = kb - wb;  // -15 .. 2
  S = nib_ext(v0, v1, s);
  N = nib_ext(v0, v1, s + dir);
}

// 2-mer context keys of a chunk (computeStrandedClippedSeq + contextWith, filters/bqsr.go:87-146,312-362), local form:
// key(b) = prev | cur << 2 with prev = the previous base in sequencing direction, complemented on reverse reads; valid only
// where both bases are ACGT and inside [lo, hi) (the caller folds the low-quality-tail bounds into that range).
// S / N from seq_nibbles with dir = reversed ? +1 : -1.  Returns the valid flags; ctx = 4-bit keys (zero where invalid).
__device__ __forceinline__ uint64_t context_nibbles(uint64_t S, uint64_t N, bool reversed, uint64_t range, uint64_t &ctx) {
  uint64_t ohS, cS, ohN, cN;
  nib_classify(S, ohS, cS);
  nib_classify(N, ohN, cN);
  const uint64_t valid = ohS & ohN & range;
  uint64_t k = cN | (cS << 2);
  k ^= reversed ? NIBF : 0ull;
  ctx = k & nib_fill(valid);
  return valid;
}

// The lane's 16 loaded bytes (four scalar members on purpose: an array member keeps the enclosing body object in scratch memory)
struct Chunk {
  uint32_t w0, w1, w2, w3;
  __device__ __forceinline__ void load(const uint8_t *__restrict__ p) {  // p is 16-byte aligned (column base 256-B aligned, padded)
    const uint4 v = *reinterpret_cast<const uint4 *>(p);
    w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w;
  }
  __device__ __forceinline__ void store(uint8_t *__restrict__ p) const { *reinterpret_cast<uint4 *>(p) = make_uint4(w0, w1, w2, w3); }
  template <int I>
  __device__ __forceinline__ uint32_t word() const { return I == 0 ? w0 : (I == 1 ? w1 : (I == 2 ? w2 : w3)); }
  template <int I>
  __device__ __forceinline__ uint32_t get() const { return (word<(I >> 2)>() >> (8 * (I & 3))) & 0xFFu; }
};

// Drives one workgroup over its tiles.  Body provides:
//   void stage(uint32_t g0, uint32_t ng)                      all threads: put per-read data of reads [g0, g0+ng) into LDS
//   void chunk_begin(uint64_t p)                              per lane chunk at byte offset p
//   void round_begin()                                        per lane, before a batch of segments
//   int  segment(uint32_t rl, int k0, int nb, int o)          bases [k0, k0+nb) of read g0+rl sit at bytes [o, o+nb) of the chunk;
//                                                             returns the number of parameter slots it used (0 or 1)
//   void round_end()                                          per lane, after at most Body::MAX_SEG slot-using segments: the per-base
//                                                             work of the batch (one code site, in a loop that almost always runs once)
//   void chunk_end(uint64_t p, int lo, int hi)                bytes [lo, hi) of the chunk belong to this group's reads
//   void group_end(uint32_t g0, uint32_t ng)                  all threads, after a barrier
//   void tile_end(uint32_t nreads)                            all threads (uniform), may contain barriers
template <class Body>
__device__ __forceinline__ void flat_run(const uint64_t *__restrict__ qual_off, uint64_t n_reads, uint64_t qual_bytes,
                                         const uint32_t *__restrict__ tile_first, FlatLds &L, Body &B) {
  const uint64_t ntiles = (qual_bytes + FL_TILE - 1) / FL_TILE;
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t tb = t * FL_TILE;
    const int32_t tlen = (int32_t)((tb + FL_TILE < qual_bytes ? tb + FL_TILE : qual_bytes) - tb);
    const uint32_t r_first = tile_first[t];
    uint32_t r_last = tile_first[t + 1];
    if (r_last >= n_reads) r_last = (uint32_t)n_reads - 1;
    for (uint32_t g0 = r_first; g0 <= r_last; g0 += FL_RMAX) {
      const uint32_t g1 = (g0 + FL_RMAX <= r_last + 1) ? g0 + FL_RMAX : r_last + 1;  // reads [g0, g1)
      const uint32_t ng = g1 - g0;
      for (uint32_t k = threadIdx.x; k <= ng; k += blockDim.x) {
        int64_t d = (int64_t)qual_off[g0 + k] - (int64_t)tb;
        const int64_t dmin = -(int64_t)FL_MAX_READ, dmax = (int64_t)FL_MAX_READ + (int64_t)FL_TILE;
        d = d < dmin ? dmin : (d > dmax ? dmax : d);
        L.off[k] = (int32_t)d;
      }
      B.stage(g0, ng);
      __syncthreads();
      const int32_t o0 = L.off[0], on = L.off[ng];
      const int32_t rb = o0 > 0 ? o0 : 0;
      const int32_t re = on < tlen ? on : tlen;
      if (rb < re) {
        const float inv_avg = (float)ng / ((float)on - (float)o0);
#pragma unroll 1
        for (int ck = (int)threadIdx.x; ck < (int)(FL_TILE / FL_CHUNK); ck += (int)blockDim.x) {
          const int32_t pr = ck * FL_CHUNK;  // chunk start relative to the tile
          const int32_t lo = pr > rb ? pr : rb, hi = (pr + FL_CHUNK) < re ? (pr + FL_CHUNK) : re;
          if (lo >= hi) continue;
          const uint64_t p = tb + (uint64_t)pr;
          B.chunk_begin(p);
          // read holding byte `lo`: guess from the mean length, then walk (exact for uniform read lengths)
          int r = (int)(((float)lo - (float)o0) * inv_avg);
          r = r < 0 ? 0 : (r >= (int)ng ? (int)ng - 1 : r);
          while (L.off[r] > lo) r--;
          while (L.off[r + 1] <= lo) r++;
          int32_t cur = lo;
#pragma unroll 1
          do {
            B.round_begin();
            int taken = 0;
            while (cur < hi && taken < Body::MAX_SEG) {
              const int32_t rs = L.off[r], rend = L.off[r + 1];
              if (rend <= cur) { r++; continue; }  // zero-length read
              const int32_t send = rend < hi ? rend : hi;
              taken += B.segment((uint32_t)r, cur - rs, send - cur, cur - pr);
              cur = send;
              r++;
            }
            B.round_end();
          } while (cur < hi);
          B.chunk_end(p, lo - pr, hi - pr);
        }
      }
      __syncthreads();
      B.group_end(g0, ng);
      __syncthreads();
    }
    B.tile_end(r_last - r_first + 1);
  }
}

}  // namespace elp
