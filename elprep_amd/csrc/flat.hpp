// flat.hpp — "flat" streaming over the ragged per-base columns (QUAL, with SEQ / skip bits / reference alongside it).
//
// The per-base kernels (Phred-sum score, BQSR gather, BQSR apply) do not assign a thread to a read.  The QUAL column is cut
// into 32 KiB tiles; a workgroup takes a tile and owns the reads that START in it (tile index built once per staged column by
// k_flat_index: one binary search per tile).  Every read is cut into BLOCKS of 16 bases aligned to the read's first base, and a
// lane owns one block at a time: lanes of a wave work on consecutive blocks, so their (unaligned) 16-byte loads still cover
// contiguous memory.  Blocks are numbered without a prefix sum: with o_k the QUAL offset of the k-th read of the group, read k
// owns the slots [(o_k + 15 k) / 16, ...): the ranges of consecutive reads cannot overlap, at most one slot per read stays
// empty, and a lane finds the read of its slot with a guess from the mean slot count plus a short LDS walk.
//
// A block never straddles reads, so there is exactly one read, one cycle/context parameter set and one reference mapping per
// lane, and everything per-base is done on the whole block with SWAR arithmetic in "nibble space": a 64-bit word holds one
// 4-bit field per base (base b of the block at bits 4b..4b+3), so masks, base codes, context keys and mismatch flags of 16
// bases cost a handful of 64-bit ALU operations and no per-base memory access.
#pragma once
#include "common.hpp"

namespace elp {

constexpr int FL_THREADS = 512;                  // default workgroup size; every body states its own as Body::NT (a compile-time constant:
                                                 // blockDim.x is a load from the dispatch packet, and a load inside the block loop drains
                                                 // vmcnt, i.e. waits for the prefetched loads of the next block)
constexpr int FL_CHUNK = 16;                     // bases per block
constexpr uint64_t FL_TILE = 32768;              // QUAL bytes per index tile (tile_first); a workgroup step covers Body::TILES of them
constexpr uint32_t FL_MAX_READ = 0x3FFFFFu;      // per-read QUAL length limit (group-relative offsets < step bytes + FL_MAX_READ stay 32-bit)

template <int RMAX>
struct FlatLds {
  uint32_t off[RMAX + 4];  // QUAL offsets of the group's reads minus the first one's (n+1 used; sized to keep LDS 16-byte aligned)
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
__device__ __forceinline__ uint64_t nib_fill(uint64_t flags) { return (flags << 4) - flags; }
// bit i of a 16-bit word -> bit 4i
__device__ __forceinline__ uint64_t nib_spread16(uint32_t x16) {
  uint64_t x = x16 & 0xFFFFu;
  x = (x | (x << 24)) & 0x000000FF000000FFull;
  x = (x | (x << 12)) & 0x000F000F000F000Full;
  x = (x | (x << 6)) & 0x0303030303030303ull;
  x = (x | (x << 3)) & NIB1;
  return x;
}
// nibble b of the result = nibble (b + sn) of the 128-bit value v1:v0, zero where b + sn < 0;  sn in [-16, 15]
__device__ __forceinline__ uint64_t nib_ext(uint64_t v0, uint64_t v1, int sn) {
  const int shr = 4 * (sn < 0 ? 0 : sn);          // 0..60
  const int shl = 4 * (sn < 0 ? -sn : 0);         // 0..64
  const uint64_t pos = (v0 >> shr) | ((v1 << 1) << (63 - shr));
  const uint64_t neg = shl >= 64 ? 0ull : (v0 << (shl & 63));
  return sn < 0 ? neg : pos;
}
// flag where the XOR of two code nibbles (values 0..3 and 8: bit 2 is never set) is non-zero
__device__ __forceinline__ uint64_t nib_code_differs(uint64_t x) { return (x | (x >> 1) | (x >> 3)) & NIB1; }
// code nibbles (A 0, C 1, G 2, T 3, anything else has bit 3 set; ctx.hip k_recode_seq): acgt = flag where the base is A/C/G/T,
// code = its 2-bit code
__device__ __forceinline__ void nib_classify(uint64_t x, uint64_t &acgt, uint64_t &code) {
  acgt = ~(x >> 3) & NIB1;
  code = x & 0x3333333333333333ull;
}

// Bases k0 .. k0+15 of a record (S) and their neighbours in sequencing direction (N: k0-1 .. k0+14 forward, k0+1 .. k0+16
// reverse) as nibbles; k0 is a multiple of 16.  Bases before the record's first read 0, bases past its end are garbage
// (callers mask).  `sp` = the record's packed bases as code nibbles, first base of a byte in the LOW nibble (k_recode_seq).
// seq_load issues the load of the 32-base window that starts at base k0 - 2 (k0 > 0) or 0; seq_unpack uses it.
__device__ __forceinline__ void seq_load(const uint8_t *__restrict__ sp, int k0, uint64_t &v0, uint64_t &v1) {
  const int lead = k0 ? 2 : 0;
  __builtin_memcpy(&v0, sp + ((k0 - lead) >> 1), 8);
  __builtin_memcpy(&v1, sp + ((k0 - lead) >> 1) + 8, 8);
}
__device__ __forceinline__ void seq_unpack(uint64_t v0, uint64_t v1, int k0, bool reversed, uint64_t &S, uint64_t &N) {
  // 32-bit funnel shifts (v_alignbit_b32: ({hi, lo} >> sh) & 0xFFFFFFFF, sh in 0..31) over the window words with one zero word
  // in front: S = window >> 4 lead bits; N forward = window >> 4 (lead - 1) (a zero nibble comes in at k0 = 0), N reverse =
  // window >> 4 (lead + 1);  lead = 2 nibbles for k0 > 0, else 0
  const uint32_t w0 = (uint32_t)v0, w1 = (uint32_t)(v0 >> 32), w2 = (uint32_t)v1;
  const uint32_t ss = k0 ? 8u : 0u;
  S = (uint64_t)__builtin_amdgcn_alignbit(w1, w0, ss) | ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, ss) << 32);
  // N: total right shift t = 32 + 4 (lead +- 1) over the words {0, w0, w1, w2}: t = 28 (k0 = 0, forward) uses {w0:0, w1:w0}
  const bool lowcase = !k0 && !reversed;
  const uint32_t ns = lowcase ? 28u : (k0 ? (reversed ? 12u : 4u) : 4u);  // k0 = 0 & reversed: >> 4
  const uint32_t a0 = lowcase ? 0u : w0, a1 = lowcase ? w0 : w1, a2 = lowcase ? w1 : w2;
  N = (uint64_t)__builtin_amdgcn_alignbit(a1, a0, ns) | ((uint64_t)__builtin_amdgcn_alignbit(a2, a1, ns) << 32);
}

// The lane's 16 loaded bytes (four scalar members on purpose: an array member keeps the enclosing body object in scratch memory)
struct Chunk {
  uint32_t w0, w1, w2, w3;
  __device__ __forceinline__ void load(const uint8_t *__restrict__ p) {  // any alignment; the columns are padded by >= 32 bytes
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    w0 = v.x; w1 = v.y; w2 = v.z; w3 = v.w;
  }
  template <int I>
  __device__ __forceinline__ uint32_t word() const { return I == 0 ? w0 : (I == 1 ? w1 : (I == 2 ? w2 : w3)); }
  template <int I>
  __device__ __forceinline__ uint32_t get() const { return (word<(I >> 2)>() >> (8 * (I & 3))) & 0xFFu; }
  // stores bytes [0, nb) at p (any alignment) and nothing else: the bytes behind a read's last block belong to another lane.
  // A partial block goes out as its binary pieces - 8, 4, 2, 1 bytes where the bit of nb is set, each taken from what the previous
  // pieces left - i.e. four predicated stores instead of a tree of per-word cases (nearly every wave holds the last block of
  // some read, so this path runs all the time).
  __device__ __forceinline__ void store(uint8_t *__restrict__ p, int nb) const {
    if (nb == FL_CHUNK) {
      const uint4 v = make_uint4(w0, w1, w2, w3);
      __builtin_memcpy(p, &v, 16);
    } else {
      // every piece and address is computed before the first store goes out: a register that an outstanding store still reads
      // can only be overwritten after a wait for that store
      const bool b8 = nb & 8, b4 = nb & 4, b2 = nb & 2, b1 = nb & 1;
      // (the words pass through registers the optimiser cannot look into: a select between two MEMBERS becomes a load from a
      // selected address, and the object that holds this chunk would move to scratch memory)
      uint32_t a0 = w0, a1 = w1, a2 = w2, a3 = w3;
      asm("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
      const uint2 d8 = make_uint2(a0, a1);
      const uint32_t d4 = b8 ? a2 : a0, x1 = b8 ? a3 : a1;     // the eight bytes behind the 8-byte piece
      const uint32_t y = b4 ? x1 : d4;                         // the four bytes behind the 4-byte piece
      const uint16_t d2 = (uint16_t)y;
      const uint8_t d1 = (uint8_t)(b2 ? y >> 16 : y);
      uint8_t *q4 = p + (nb & 8), *q2 = p + (nb & 12), *q1 = p + (nb & 14);
      __builtin_amdgcn_sched_barrier(0);
      if (b8) __builtin_memcpy(p, &d8, 8);
      if (b4) __builtin_memcpy(q4, &d4, 4);
      if (b2) __builtin_memcpy(q2, &d2, 2);
      if (b1) *q1 = d1;
    }
  }
};

// Drives one workgroup over its tiles.  Body provides:
//   static constexpr int NT                                   workgroup size the kernel is launched with
//   static constexpr int TILES, RMAX                          index tiles per workgroup step; reads held in LDS at a time (a step's
//                                                             reads are handled in groups of at most RMAX)
//   struct Pre                                                the registers a block's global loads land in (+ what identifies it)
//   void stage(uint32_t g0, uint32_t ng)                      all threads: put per-read data of reads [g0, g0+ng) into LDS
//   void slots(uint32_t nslots)                               all threads (uniform), behind the barrier: the group has nslots slots
//   bool prefetch(uint32_t rl, int k0, int nb, uint64_t qpos, uint32_t slot, Pre &)
//                                                             per lane: ISSUE the global loads of bases [k0, k0+nb) of read g0+rl
//                                                             (k0 a multiple of 16, 1 <= nb <= 16; first QUAL byte at column offset
//                                                             qpos; the block is slot `slot` of the group) without using their results;
//                                                             false = nothing to do for the block
//   void process(Pre &)                                       per lane: the block's work, except its global stores
//   void retire()                                             per lane: issue the global stores of the block processed last.  Called
//                                                             in front of the next prefetch: a store issued behind the prefetch loads
//                                                             would sit in the same in-order counter (vmcnt) and be waited for, with
//                                                             its full latency, as soon as the loaded registers are needed
//   void group_end(uint32_t g0, uint32_t ng)                  all threads, after a barrier
//   void tile_end(uint32_t nreads, uint64_t nbases)           all threads (uniform), may contain barriers
// The lane's next block is prefetched before the current one is processed, so the HBM latency of block i+1 hides behind the ALU
// and LDS work of block i (the loads stay in flight across the loop's back edge).
// tile_first[t] = first read whose QUAL offset is >= t * FL_TILE (tile_first[ntiles] = n_reads).
// flat_steps(): number of workgroup steps of a body over a column = the most workgroups worth launching.
template <class Body>
__device__ __forceinline__ void flat_run(const uint64_t *__restrict__ qual_off, uint64_t n_reads, uint64_t qual_bytes,
                                         const uint32_t *__restrict__ tile_first, FlatLds<Body::RMAX> &L, Body &B) {
  constexpr uint32_t RMAX = Body::RMAX;
  const uint64_t ntiles = (qual_bytes + FL_TILE - 1) / FL_TILE, nsteps = (ntiles + Body::TILES - 1) / Body::TILES;
  for (uint64_t t = blockIdx.x; t < nsteps; t += gridDim.x) {
    const uint64_t t1 = (t + 1) * Body::TILES < ntiles ? (t + 1) * Body::TILES : ntiles;
    const uint32_t r_first = tile_first[t * Body::TILES], r_end = tile_first[t1];
    for (uint32_t g0 = r_first; g0 < r_end; g0 += RMAX) {
      const uint32_t ng = (r_end - g0 < RMAX) ? r_end - g0 : RMAX;  // reads [g0, g0 + ng)
      const uint64_t base = qual_off[g0];
      const uint32_t len0 = (uint32_t)(qual_off[g0 + 1] - base);
      int same = 1;  // every read of the group has len0 bases (what a sequencer writes): slots are found by arithmetic alone
      for (uint32_t k = threadIdx.x; k <= ng; k += Body::NT) {
        const uint32_t o = (uint32_t)(qual_off[g0 + k] - base);
        L.off[k] = o;
        same &= o == k * len0;
      }
      B.stage(g0, ng);
      const bool uniform = __syncthreads_and(same) != 0 && len0 != 0;
      const uint32_t nslots = (L.off[ng] + 15u * ng) >> 4;  // slot of read k: (off[k] + 15 k) >> 4
      B.slots(nslots);
      const float inv_avg = nslots ? (float)ng / (float)nslots : 0.f, inv_step = 1.0f / (float)(len0 + 15u);
      // Uniform groups: slot s belongs to read k = (16 s + 15) / step (step = len0 + 15: read k starts at slot (step k) >> 4), and with
      // r = (16 s + 15) % step the block starts at base r & ~15 of the read.  A lane's slots are NT apart, so (k, r) of its next
      // slot follow from the current ones by adding the (uniform) quotient and remainder of 16 NT / step - a handful of VALU
      // instructions per block instead of a division.
      const uint32_t step = len0 + 15u;  // < 2^24, like every read index: 24-bit multiplies
      uint32_t uk = 0, ur = 0, udk = 0, udr = 0;
      if (uniform) {
        udk = (16u * Body::NT) / step;
        udr = (16u * Body::NT) - udk * step;
        const uint32_t q = 16u * threadIdx.x + 15u;  // < 2^24: the float quotient is off by at most one
        int k = (int)((float)q * inv_step);
        k = (uint32_t)k * step > q ? k - 1 : k;
        k = ((uint32_t)k + 1u) * step <= q ? k + 1 : k;
        uk = (uint32_t)k;
        ur = q - uk * step;
      }
      // locate slot s and issue its loads (the uniform path must be called for s = threadIdx.x, then s + NT, s + 2 NT, ... in turn)
      auto fetch = [&](uint32_t s, typename Body::Pre &pre) __attribute__((always_inline)) -> bool {
        if (uniform) {
          const uint32_t k = uk, k0 = ur & ~15u;
          // state of the lane's next slot
          ur += udr;
          const bool wrap = ur >= step;
          uk += udk + (wrap ? 1u : 0u);
          ur = wrap ? ur - step : ur;
          if (k0 >= len0) return false;  // the (at most one) empty slot behind a read
          const uint32_t nb = len0 - k0 < 16u ? len0 - k0 : 16u;
          return B.prefetch(k, (int)k0, (int)nb, base + __umul24(len0, k) + k0, s, pre);
        }
        // read owning slot s: guess g from the mean; the four offsets around g are read at once (one LDS latency) and decide
        // among g-1, g, g+1; otherwise walk
        int k = (int)((float)s * inv_avg);
        k = k >= (int)ng - 1 ? (int)ng - 2 : k;
        k = k < 1 ? 1 : k;
        // straight-line on purpose (bitwise &, selects): with && the compiler nests branches and reads the offsets in two
        // dependent LDS round trips.  For ng < 3 the four words are read all the same (the array is longer) and not used.
        const uint32_t a0 = L.off[k - 1], a1 = L.off[k], a2 = L.off[k + 1], a3 = L.off[k + 2];
        const uint32_t v0 = (a0 + 15u * (uint32_t)(k - 1)) >> 4, v1 = (a1 + 15u * (uint32_t)k) >> 4, v2 = (a2 + 15u * (uint32_t)(k + 1)) >> 4,
                       v3 = (a3 + 15u * (uint32_t)(k + 2)) >> 4;
        const bool found = (ng >= 3u) & (s >= v0) & (s < v3);
        const bool hi = s >= v2, mid = s >= v1;
        uint32_t o = hi ? a2 : (mid ? a1 : a0);
        uint32_t nxt = hi ? a3 : (mid ? a2 : a1);
        k = found ? (hi ? k + 1 : (mid ? k : k - 1)) : k;
        if (!found) {
          k = k >= (int)ng ? (int)ng - 1 : k;
          while (((L.off[k] + 15u * (uint32_t)k) >> 4) > s) k--;
          while (((L.off[k + 1] + 15u * (uint32_t)(k + 1)) >> 4) <= s) k++;
          o = L.off[k];
          nxt = L.off[k + 1];
        }
        const uint32_t len = nxt - o;
        const uint32_t k0 = (s - ((o + 15u * (uint32_t)k) >> 4)) << 4;
        if (k0 >= len) return false;  // the (at most one) empty slot behind a read
        const uint32_t nb = len - k0 < 16u ? len - k0 : 16u;
        return B.prefetch((uint32_t)k, (int)k0, (int)nb, base + o + k0, s, pre);
      };
      uint32_t s = threadIdx.x;
      typename Body::Pre cur;
      bool have = s < nslots ? fetch(s, cur) : false;
#pragma unroll 1
      while (s < nslots) {
        B.retire();
        const uint32_t sn = s + Body::NT;
        typename Body::Pre nxt;
        const bool have_n = sn < nslots ? fetch(sn, nxt) : false;
        if (have) B.process(cur);
        cur = nxt;
        have = have_n;
        s = sn;
      }
      B.retire();
      __syncthreads();
      B.group_end(g0, ng);
      __syncthreads();
    }
    B.tile_end(r_end - r_first, qual_off[r_end] - qual_off[r_first]);
  }
}

template <class Body>
inline uint64_t flat_steps(uint64_t qual_bytes) {
  const uint64_t ntiles = (qual_bytes + FL_TILE - 1) / FL_TILE;
  return (ntiles + Body::TILES - 1) / Body::TILES;
}

}  // namespace elp
