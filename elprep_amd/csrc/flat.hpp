// flat.hpp — "flat" streaming over the ragged per-base columns (QUAL, and SEQ alongside it).
//
// The per-base kernels (Phred-sum score, BQSR gather, BQSR apply) do not assign a thread to a read.  A workgroup takes a
// 16 KiB tile of the QUAL byte column; every lane loads one aligned 16-byte chunk (1 KiB per wave instruction, fully
// coalesced) and works on the 16 consecutive bases it holds.  Which reads overlap a tile comes from a tile index built once
// per staged column (k_flat_index: one binary search per tile, all tiles in parallel); the reads' QUAL offsets (and whatever
// per-read data the kernel wants) sit in LDS, and a lane finds the read of its chunk with a guess from the mean read length
// plus a short LDS walk.  A chunk that straddles a read boundary is handled as two (or more) segments by the same lane.
#pragma once
#include "common.hpp"

namespace elp {

constexpr int FL_THREADS = 256;
constexpr int FL_CHUNK = 16;
constexpr uint64_t FL_TILE = (uint64_t)FL_THREADS * FL_CHUNK * 4;  // 16 KiB of QUAL bytes per tile
constexpr int FL_RMAX = 256;                                       // reads held in LDS at a time (a tile of 150-base reads has ~110)

// the lane's 16 loaded bytes, kept in two 64-bit registers (no dynamically indexed private array => no scratch memory)
struct Chunk {
  uint64_t w0, w1;
  __device__ __forceinline__ uint32_t get(int i) const { return (uint32_t)((i < 8 ? (w0 >> (8 * i)) : (w1 >> (8 * (i - 8)))) & 0xFF); }
  __device__ __forceinline__ void set(int i, uint32_t v) {
    // branch-free (a conditional store to w0 or w1 would be lowered to a dynamically indexed private array = scratch memory)
    const uint64_t m = 0xFFull << (8 * (i & 7)), b = (uint64_t)(v & 0xFF) << (8 * (i & 7));
    const uint64_t m0 = i < 8 ? m : 0ull, m1 = i < 8 ? 0ull : m;
    w0 = (w0 & ~m0) | (b & m0);
    w1 = (w1 & ~m1) | (b & m1);
  }
};

struct FlatLds {
  uint64_t off[FL_RMAX + 1];
};

int ensure_flat_index(elp_ctx *c);  // builds c->tile_first for the staged QUAL column (sort.hip)

// Calls, for every group of <= FL_RMAX reads overlapping the tile:
//   gbegin(g0, ng)                      by all threads after the group's offsets are in LDS (followed by a barrier)
//   fn(rl, k0, k1, bytes, o, p)         per lane, for every maximal run of bases [k0, k1) of read g0 + rl that lies in the lane's
//                                       16-byte chunk; o = index of base k0 inside the chunk; p = byte offset of the chunk
//   done(p, lo, hi, bytes)              per lane chunk after its segments; [lo, hi) = byte sub-range of the chunk covered
//   gend(g0, ng)                        by all threads after a barrier that follows the chunk loop
// r_first / r_last come from the tile index (tile_first[t], tile_first[t + 1]).  Returns the number of reads overlapping the
// tile (uniform across the workgroup).
template <class GBegin, class SegFn, class DoneFn, class GEnd>
__device__ __forceinline__ uint32_t flat_tile(const uint64_t *__restrict__ qual_off, uint64_t n_reads, const uint8_t *__restrict__ qual,
                                     uint64_t tile_begin, uint64_t tile_end, uint32_t r_first, uint32_t r_last, FlatLds &L, GBegin gbegin,
                                     SegFn fn, DoneFn done, GEnd gend) {
  if (r_last >= n_reads) r_last = (uint32_t)n_reads - 1;
  for (uint32_t g0 = r_first; g0 <= r_last; g0 += FL_RMAX) {
    const uint32_t g1 = (g0 + FL_RMAX <= r_last + 1) ? g0 + FL_RMAX : r_last + 1;  // reads [g0, g1)
    const uint32_t ng = g1 - g0;
    for (uint32_t k = threadIdx.x; k <= ng; k += FL_THREADS) L.off[k] = qual_off[g0 + k];
    __syncthreads();
    gbegin(g0, ng);
    __syncthreads();
    const uint64_t o0 = L.off[0], on = L.off[ng];
    const uint64_t rb = o0 > tile_begin ? o0 : tile_begin;
    const uint64_t re = on < tile_end ? on : tile_end;
    if (rb < re) {
      const float inv_avg = (float)ng / (float)(on - o0);
      const uint64_t first_chunk = rb & ~(uint64_t)(FL_CHUNK - 1);
      for (uint64_t p = first_chunk + (uint64_t)threadIdx.x * FL_CHUNK; p < re; p += (uint64_t)FL_THREADS * FL_CHUNK) {
        const uint64_t lo = p > rb ? p : rb, hi = (p + FL_CHUNK) < re ? (p + FL_CHUNK) : re;
        if (lo >= hi) continue;
        Chunk bytes;
        {
          const uint4 v = *reinterpret_cast<const uint4 *>(qual + p);  // column base is 256-B aligned and padded by 16 B
          bytes.w0 = (uint64_t)v.x | ((uint64_t)v.y << 32);
          bytes.w1 = (uint64_t)v.z | ((uint64_t)v.w << 32);
        }
        // read holding byte `lo`: guess from the mean length, then walk (exact for uniform read lengths)
        uint32_t r = (uint32_t)((float)(lo - o0) * inv_avg);
        if (r >= ng) r = ng - 1;
        while (L.off[r] > lo) r--;
        while (L.off[r + 1] <= lo) r++;
        uint64_t cur = lo;
        while (cur < hi) {
          const uint64_t rs = L.off[r], rend = L.off[r + 1];
          if (rend <= cur) { r++; continue; }  // zero-length read
          const uint64_t send = rend < hi ? rend : hi;
          fn(r, (int)(cur - rs), (int)(send - rs), bytes, (int)(cur - p), p);
          cur = send;
          r++;
        }
        done(p, (int)(lo - p), (int)(hi - p), bytes);
      }
    }
    __syncthreads();
    gend(g0, ng);
    __syncthreads();
  }
  return r_last - r_first + 1;
}

}  // namespace elp
