// apply3.hip — ApplyBQSR for read sets of one length (round 3): the loop shape of count3.hip on the apply side.
//
// Reference: BaseRecalibratorTables.ApplyBQSR (filters/bqsr.go:936-1005): every base with quality >= 6 of a record whose read group is
// in the tables is replaced by the memo value of (read group, quality, cycle, context); cycle and context are taken on the full,
// unclipped read.  Same bytes as k_bqsr_apply_flat (bqsr.hip), which stays as the general kernel (ragged lengths, reads longer than
// --max-cycle, LUTs with more than ~250 distinct rows).
//
//   - A workgroup trip covers RPI = 1024 / (blocks per read) whole reads; lane t is block t % bpr of read slot t / bpr for the whole
//     kernel: addresses are a wave-uniform base plus a constant lane offset; no groups, no barriers.
//   - Every block is a full 16 bytes: the last block of a read whose length is not a multiple of 16 is moved back to END at the read's
//     end and overlaps its neighbour (both lanes compute the same bytes from the same original values: the two blocks sit in one
//     wave - the layout is chosen so - and the wave's loads of a read come a trip before its stores).  One 16-byte load and ONE
//     16-byte store per lane and trip, the store issued by every lane (lanes without work write to a dump area), so that the trip's
//     wait can leave exactly that store in flight (gload.hpp).
//   - Per read an 8-byte record (context window, covariate, direction, cycle origin) made by k_apply_records from the columns.
//   - Two-level LUT in LDS as in k_bqsr_apply_flat (level 1: (covariate, quality, cycle) -> id of one of the few distinct 17-byte LUT
//     rows; level 2: the rows), with two changes: level 1 starts at quality 0 - qualities 0..5 map to IDENTITY rows (their bytes equal
//     the quality), so "ApplyBQSR leaves qualities below 6 alone" needs no byte masks behind the look-ups - and every quality from 6
//     up to the largest one seen in the sample of the column is resident (a quality between 6 and the smallest sampled one was read
//     from the wrong row by the older kernel's clamp); qualities above the resident range read the 0x80 row and take the fix-up loop.
#include <algorithm>

#include "apply_rec.hpp"
#include "bqsr_common.hpp"
#include "gload.hpp"

namespace elp {

constexpr uint32_t A3_N1 = 0x11111111u, A3_C3 = 0x33333333u;
constexpr int A3_ROW = 20;  // bytes between two level-2 rows in LDS (17 used)
constexpr int A3_NT = 512;  // three workgroups per CU around three copies of the LUT (~50 KB each): six waves per SIMD
__global__ __launch_bounds__(256) void k_apply_records(uint64_t n, uint32_t len, int lmax, const uint16_t *__restrict__ flag, const uint16_t *__restrict__ rgid,
                                                       const uint16_t *__restrict__ rg_cov, const uint64_t *__restrict__ qbounds,
                                                       const uint8_t *__restrict__ cov_present, uint2 *__restrict__ recs, uint32_t *err) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint16_t rg = rgid[i], f = flag[i];
  const uint64_t qb = qbounds[i];
  uint2 r = make_uint2(0u, 0u);
  if (rg == ELP_NIL16) {
    atomicOr(&err[0], 32u);  // readGroupCovariate panics, bqsr.go:38
  } else {
    const uint32_t cov = rg_cov[rg];
    if (cov_present[cov]) r = apply_record(len, lmax, f, qb, cov);  // else: read group absent from the tables, read untouched (:953-955)
  }
  recs[i] = r;
}

// Round 5 - the records SPLIT BY COVARIATE (many read groups: the level-1 tables of all covariates do not fit one workgroup's LDS, or the
// LUT has more distinct rows than one-byte ids hold): only the reads ApplyBQSR touches get a record, the records of one covariate lie
// together (counts per covariate, offsets, a scatter: the shape of k_c3_other_* in bqsr.hip), the read's staging index next to the
// record; a workgroup of k_bqsr_apply3<true> then holds ONE covariate's level 1 and that covariate's own row dictionary at a time.
constexpr int A3_MAXCOV = 256, A3_RTILE = 1024;
__device__ __forceinline__ int apply_read_cov(uint16_t rg, const uint16_t *__restrict__ rg_cov, const uint8_t *__restrict__ cov_present, uint32_t *err) {
  if (rg == ELP_NIL16) { if (err) atomicOr(&err[0], 32u); return -1; }
  const uint32_t cov = rg_cov[rg] & 0xFFu;
  return cov_present[cov] ? (int)cov : -1;
}
// Both passes run the SAME few workgroups over the same chunks of tiles (workgroup b: tiles [b * per, (b + 1) * per)): the counters all
// workgroups add to share one cache line, and a line takes about one atomic per clock - with a workgroup per tile (15 700 of them at
// 16 M reads, 16 to 32 counters each) those flushes were the two kernels' time (0.19 + 0.21 ms; 0.03 of it the loads).
constexpr int A3_RBLOCKS = 1024;
__global__ __launch_bounds__(256) void k_apply_cov_hist(uint64_t n, uint32_t per, const uint16_t *__restrict__ rgid, const uint16_t *__restrict__ rg_cov,
                                                        const uint8_t *__restrict__ cov_present, uint32_t *__restrict__ cnt /* [A3_MAXCOV] */,
                                                        uint32_t *__restrict__ blk /* [gridDim.x][A3_MAXCOV] */, uint32_t *err) {
  __shared__ uint32_t h[A3_MAXCOV];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t t = 0; t < per; t++) {
    const uint64_t i0 = ((uint64_t)blockIdx.x * per + t) * A3_RTILE + threadIdx.x;
    if (i0 - threadIdx.x >= n) break;
    // (the tile's four read-group ids first, then the covariates they lead to: one round trip per level, not four)
    uint16_t rg[A3_RTILE / 256];
#pragma unroll
    for (int j = 0; j < A3_RTILE / 256; j++) rg[j] = i0 + j * 256 < n ? rgid[i0 + j * 256] : (uint16_t)ELP_NIL16;
    int cv[A3_RTILE / 256];
#pragma unroll
    for (int j = 0; j < A3_RTILE / 256; j++) cv[j] = i0 + j * 256 < n ? apply_read_cov(rg[j], rg_cov, cov_present, err) : -1;
    // (every lane adds its one: lanes that share a counter serialise in the LDS - 30 cycles for eight lanes per counter -, cheaper than
    // forming the wave's groups by covariate first)
#pragma unroll
    for (int j = 0; j < A3_RTILE / 256; j++)
      if (cv[j] >= 0) atomicAdd(&h[cv[j]], 1u);
  }
  __syncthreads();
  blk[(size_t)blockIdx.x * A3_MAXCOV + threadIdx.x] = h[threadIdx.x];
  if (h[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_apply_cov_offsets(const uint32_t *__restrict__ cnt, uint32_t *__restrict__ off /* [A3_MAXCOV + 1] */, uint32_t *__restrict__ cursor) {
  uint32_t at = 0;
  for (int c = 0; c < A3_MAXCOV; c++) { off[c] = at; cursor[c] = at; at += cnt[c]; }
  off[A3_MAXCOV] = at;
}
__global__ __launch_bounds__(256) void k_apply_records_split(uint64_t n, uint32_t per, uint32_t len, int lmax, const uint16_t *__restrict__ flag,
                                                             const uint16_t *__restrict__ rgid, const uint16_t *__restrict__ rg_cov, const uint64_t *__restrict__ qbounds,
                                                             const uint8_t *__restrict__ cov_present, const uint32_t *__restrict__ blk, uint32_t *cursor,
                                                             uint2 *__restrict__ recs, uint32_t *__restrict__ ridx, uint32_t *err) {
  // the workgroup's places in every covariate's range (its counts of the first pass), then a running counter per covariate in the LDS
  __shared__ uint32_t run[A3_MAXCOV], base[A3_MAXCOV];
  {
    const uint32_t mine = blk[(size_t)blockIdx.x * A3_MAXCOV + threadIdx.x];
    base[threadIdx.x] = mine ? atomicAdd(&cursor[threadIdx.x], mine) : 0u;
    run[threadIdx.x] = 0;
  }
  __syncthreads();
  for (uint32_t t = 0; t < per; t++) {
    const uint64_t i0 = ((uint64_t)blockIdx.x * per + t) * A3_RTILE + threadIdx.x;
    if (i0 - threadIdx.x >= n) break;
    int cv[A3_RTILE / 256];
    uint16_t rg[A3_RTILE / 256], fl[A3_RTILE / 256];
    uint64_t qb[A3_RTILE / 256];
#pragma unroll
    for (int j = 0; j < A3_RTILE / 256; j++) {  // every column load of the tile up front
      const uint64_t i = i0 + j * 256;
      rg[j] = i < n ? rgid[i] : (uint16_t)ELP_NIL16;
      fl[j] = i < n ? flag[i] : (uint16_t)0;
      qb[j] = i < n ? qbounds[i] : 0ull;
    }
#pragma unroll
    for (int j = 0; j < A3_RTILE / 256; j++) cv[j] = i0 + j * 256 < n ? apply_read_cov(rg[j], rg_cov, cov_present, nullptr) : -1;  // (errors: noted by the first pass)
#pragma unroll
    for (int j = 0; j < A3_RTILE / 256; j++) {
      if (cv[j] >= 0) {
        const size_t to = (size_t)base[cv[j]] + atomicAdd(&run[cv[j]], 1u);
        recs[to] = apply_record(len, lmax, fl[j], qb[j], (uint32_t)cv[j]);
        ridx[to] = (uint32_t)(i0 + j * 256);
      }
    }
  }
}

struct Apply3Args {
  uint64_t n;
  uint32_t len;
  uint8_t *qual;
  const uint8_t *seq4;  // first SEQ byte of read 0
  const uint2 *recs;
  const uint8_t *lut;   // dense [n_cov][94][2*max_cycle+1][17] (fix-up path)
  const uint16_t *t1;   // [n_cov][n_qi + 1][w] row ids for qualities 6 .. qhi (+ the "not resident" row), w = 2 lmax + 1
  const uint8_t *t2;    // [n_dict + 1][17]
  const uint32_t *n_dict;  // number of distinct rows, on the device (no read-back in front of the launch)
  int n_cov, n_qi, lmax, max_cycle;
  uint32_t *err;
  uint8_t *dump;   // 16 bytes per lane of the launch: where lanes without a block store
  uint32_t group;  // lanes that share reads: A3_NT (reads packed over the whole workgroup) or 64 (whole reads per wave)
  // SPLIT (k_bqsr_apply3<true>): recs / ridx hold the records sorted by covariate; t1's ids and t2 [n_cov][256][17] / n_dict [n_cov] are
  // per covariate
  const uint32_t *ridx;     // staging index of record k
  const uint32_t *cov_cnt;  // [A3_MAXCOV] records per covariate
  const uint32_t *cov_off;  // [A3_MAXCOV + 1] first record of a covariate
  const uint8_t *cov_present;  // not null: the records are the adapt stage's (apply_rec.hpp) - whether a read's group is in the tables is tested here
};

struct A3Data { u32x4 q, s; };  // QUAL bytes; SEQ window (three words used): the asm loads of gload.hpp write these registers

__device__ __forceinline__ uint32_t lds_u8(uint32_t at) { return *reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>((uintptr_t)at); }

template <int BYTE>
__device__ __forceinline__ uint32_t byte_min(uint32_t w, uint32_t hi) {  // min((w >> 8 BYTE) & 0xFF, hi) in one instruction
  uint32_t r;
  if (BYTE == 0) asm("v_min_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(hi));
  else if (BYTE == 1) asm("v_min_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(hi));
  else if (BYTE == 2) asm("v_min_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(hi));
  else asm("v_min_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(hi));
  return r;
}
template <int BYTE>
__device__ __forceinline__ uint32_t byte_add(uint32_t w, uint32_t b) {  // ((w >> 8 BYTE) & 0xFF) + b in one instruction
  uint32_t r;
  if (BYTE == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(b));
  else if (BYTE == 1) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(b));
  else if (BYTE == 2) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(b));
  else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(w), "v"(b));
  return r;
}

// nibble word of four bases (bits 4 b .. 4 b + 3, b = 0..3, in the low 16 bits of x) -> one byte per base
__device__ __forceinline__ uint32_t nib4_to_bytes(uint32_t x) {
  x = (x | (x << 8)) & 0x00FF00FFu;
  return (x | (x << 4)) & 0x0F0F0F0Fu;
}

template <bool SPLIT>
struct Apply3 {
  uint32_t k0, qoff, soff, ssh;  // first base of the lane's block; byte offsets into the trip's QUAL / SEQ span; SEQ shift
  uint32_t t1_at, t2_at;  // LDS byte addresses of the two levels
  uint32_t w, rows_w;     // level-1 entries per (covariate, quality) row / per covariate
  uint32_t qhi1;          // qualities above qhi1 - 1 read row qhi1: "not resident"
  int lmax, max_cycle;
  const uint8_t *__restrict__ lut;
  uint32_t err;

  // one base: level 1 (covariate, min(quality, qhi1), cycle) -> row id, level 2 (row, context) -> new quality
  template <int I>
  __device__ __forceinline__ uint32_t base(uint32_t qw, uint32_t cxw, uint32_t c1, uint32_t t2r) const {
    const uint32_t qc = byte_min<(I & 3)>(qw, qhi1);
    const uint32_t id = lds_u8(__umul24(qc, w) + c1);
    if (A3_ROW == 32) return lds_u8(lshl_add_u32<5>(id, byte_add<(I & 3)>(cxw, t2r)));
    return lds_u8(__umul24(id, (uint32_t)A3_ROW) + byte_add<(I & 3)>(cxw, t2r));
  }

  // bases whose look-up hit the 0x80 row: quality above the resident range -> dense LUT; quality > 93 -> error
  __device__ __forceinline__ void fixup(uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3, const A3Data &d, uint32_t cov, int cyc0, int ci, uint32_t FV_lo,
                                        uint32_t FV_hi, uint32_t CX_lo, uint32_t CX_hi) {
    const int ncyc = 2 * max_cycle + 1;
    uint64_t lo = (uint64_t)o0 | ((uint64_t)o1 << 32), hi = (uint64_t)o2 | ((uint64_t)o3 << 32);
    const uint64_t qlo = (uint64_t)d.q.x | ((uint64_t)d.q.y << 32), qhi = (uint64_t)d.q.z | ((uint64_t)d.q.w << 32);
#pragma unroll 1
    for (int i = 0; i < 16; i++) {
      const int bs = 8 * (i & 7);
      const uint32_t q = (uint32_t)(((i & 8) ? qhi : qlo) >> bs) & 0xFFu;
      if (q < qhi1) continue;  // resident: done by the straight-line code
      uint64_t v = q;
      if (q >= (uint32_t)ELP_NQUAL) {
        err |= 8u;
      } else {
        const uint32_t fv = ((i & 8) ? FV_hi : FV_lo) >> (4 * (i & 7)), cxn = ((i & 8) ? CX_hi : CX_lo) >> (4 * (i & 7));
        const uint32_t cx = (fv & 1u) ? (cxn & 15u) : 16u;
        v = (uint64_t)lut[(((size_t)cov * ELP_NQUAL + q) * ncyc + (size_t)(cyc0 + i * ci + max_cycle)) * 17 + cx];
      }
      const uint64_t m = ~(0xFFull << bs);
      lo = (i & 8) ? lo : ((lo & m) | (v << bs));
      hi = (i & 8) ? ((hi & m) | (v << bs)) : hi;
    }
    o0 = (uint32_t)lo; o1 = (uint32_t)(lo >> 32); o2 = (uint32_t)hi; o3 = (uint32_t)(hi >> 32);
  }

  __device__ __forceinline__ u32x4 process(u32x2 rec, const A3Data &d) {
    const uint32_t fl = rec.y;
    const bool rev = fl & AR_REV;
    const uint32_t cov = fl & 0xFFu;
    // SEQ: S = the block's bases, N = their predecessors in sequencing direction
    // (the window starts at the byte in front of the one that holds base k0: 8 bits to base k0 if k0 is even, 12 if it is odd)
    const uint32_t ns = rev ? ssh + 4u : ssh - 4u;
    const uint32_t S_lo = __builtin_amdgcn_alignbit(d.s.y, d.s.x, ssh), S_hi = __builtin_amdgcn_alignbit(d.s.z, d.s.y, ssh);
    const uint32_t N_lo = __builtin_amdgcn_alignbit(d.s.y, d.s.x, ns), N_hi = __builtin_amdgcn_alignbit(d.s.z, d.s.y, ns);
    const uint32_t o_lo = ((S_lo | N_lo) >> 3) & A3_N1, o_hi = ((S_hi | N_hi) >> 3) & A3_N1;  // base or predecessor not A / C / G / T
    const uint64_t cw = nib_range_clamped((int)(rec.x & 0xFFFFu) - (int)k0, (int)(rec.x >> 16) - (int)k0);
    const uint32_t FV_lo = (uint32_t)cw & ~o_lo, FV_hi = (uint32_t)(cw >> 32) & ~o_hi;  // context valid
    const uint32_t rm = rev ? 0xFFFFFFFFu : 0u;
    const uint32_t CX_lo = ((N_lo & A3_C3) | ((S_lo & A3_C3) << 2)) ^ rm, CX_hi = ((N_hi & A3_C3) | ((S_hi & A3_C3) << 2)) ^ rm;
    // context index per base as a byte: 0 .. 15, 16 = no context
    const uint32_t X_lo = (CX_lo & (FV_lo * 15u)), X_hi = (CX_hi & (FV_hi * 15u));
    const uint32_t nv_lo = ~FV_lo & A3_N1, nv_hi = ~FV_hi & A3_N1;
    const uint32_t c0 = nib4_to_bytes(X_lo & 0xFFFFu) | (nib4_to_bytes(nv_lo & 0xFFFFu) << 4), c1w = nib4_to_bytes(X_lo >> 16) | (nib4_to_bytes(nv_lo >> 16) << 4);
    const uint32_t c2 = nib4_to_bytes(X_hi & 0xFFFFu) | (nib4_to_bytes(nv_hi & 0xFFFFu) << 4), c3 = nib4_to_bytes(X_hi >> 16) | (nib4_to_bytes(nv_hi >> 16) << 4);
    const int ci = (fl & AR_NEG) ? -1 : 1;
    const int cyc0l = (int)(fl >> 16) + ci * (int)k0;  // cycle of the block's first base + lmax
    const uint32_t l1 = t1_at + (SPLIT ? 0u : __umul24(cov, rows_w)) + (uint32_t)cyc0l;  // (SPLIT: the one covariate's level 1 is resident)
    const uint32_t uci = (uint32_t)ci;
    const uint32_t t2r = t2_at;
#define ELP_A3(I, QW, CW) base<I>(QW, CW, l1 + (uint32_t)(I) * uci, t2r)
    const uint32_t b0 = ELP_A3(0, d.q.x, c0), b1 = ELP_A3(1, d.q.x, c0), b2 = ELP_A3(2, d.q.x, c0), b3 = ELP_A3(3, d.q.x, c0);
    const uint32_t b4 = ELP_A3(4, d.q.y, c1w), b5 = ELP_A3(5, d.q.y, c1w), b6 = ELP_A3(6, d.q.y, c1w), b7 = ELP_A3(7, d.q.y, c1w);
    const uint32_t b8 = ELP_A3(8, d.q.z, c2), b9 = ELP_A3(9, d.q.z, c2), b10 = ELP_A3(10, d.q.z, c2), b11 = ELP_A3(11, d.q.z, c2);
    const uint32_t b12 = ELP_A3(12, d.q.w, c3), b13 = ELP_A3(13, d.q.w, c3), b14 = ELP_A3(14, d.q.w, c3), b15 = ELP_A3(15, d.q.w, c3);
#undef ELP_A3
    uint32_t o0 = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24), o1 = b4 | (b5 << 8) | (b6 << 16) | (b7 << 24);
    uint32_t o2 = b8 | (b9 << 8) | (b10 << 16) | (b11 << 24), o3 = b12 | (b13 << 8) | (b14 << 16) | (b15 << 24);
    if (((o0 | o1) | (o2 | o3)) & 0x80808080u) fixup(o0, o1, o2, o3, d, cov, cyc0l - lmax, ci, FV_lo, FV_hi, CX_lo, CX_hi);
    return (u32x4){o0, o1, o2, o3};
  }
};

template <bool SPLIT>
__global__ __launch_bounds__(A3_NT, 6) void k_bqsr_apply3(Apply3Args A) {  // six waves per SIMD = three workgroups per CU: <= 80 vector registers
  extern __shared__ __attribute__((aligned(16))) uint8_t llut[];
  // level 1 in LDS: [n_cov][qhi1 + 1][w] bytes (SPLIT: ONE covariate's [qhi1 + 1][w]), rows for qualities 0 .. 5 (identity), 6 .. qhi (from
  // A.t1), qhi1 ("not resident"); level 2: rows A3_ROW bytes apart: 0 .. n_dict - 1 the distinct LUT rows, n_dict the 0x80 row,
  // n_dict + 1 + q the identity row of q < 6
  const int w = 2 * A.lmax + 1, qhi1 = 6 + A.n_qi, rows_w = (qhi1 + 1) * w, n1 = (SPLIT ? 1 : A.n_cov) * rows_w;
  const int t1_bytes = (n1 + 15) & ~15;
  // more distinct rows than one-byte ids (and the LDS the launch reserved) hold - SPLIT: in any covariate's dictionary -: nothing is
  // touched, the host takes another form
  {
    int over = 0;
    if (SPLIT) { for (int k = threadIdx.x; k < A.n_cov; k += A3_NT) over |= (int)A.n_dict[k] + 7 > 256; }
    else over = (int)*A.n_dict + 7 > 256;
    if (__syncthreads_or(over)) {
      if (threadIdx.x == 0 && blockIdx.x == 0) atomicOr(&A.err[0], 512u);
      return;
    }
  }
  // fills the tables: level 1 of covariates [cov_lo, cov_lo + ncv) and the dictionary with `n_dict` rows at t2
  auto fill = [&](int cov_lo, int ncv, int n_dict, const uint8_t *t2) {
    for (int k = threadIdx.x; k < ncv * rows_w; k += A3_NT) {
      const int x = k % w, q = (k / w) % (qhi1 + 1), cov = cov_lo + k / rows_w;
      uint32_t id;
      if (q < 6) id = (uint32_t)(n_dict + 1 + q);
      else id = A.t1[((size_t)cov * (A.n_qi + 1) + (size_t)(q - 6)) * w + x];  // row n_qi of t1 is the "not resident" row (id n_dict)
      llut[k] = (uint8_t)id;
    }
    const int n2 = (n_dict + 7) * 17;
    for (int k = threadIdx.x; k < n2; k += A3_NT) {
      const int row = k / 17, cx = k - 17 * row;
      llut[t1_bytes + A3_ROW * row + cx] = row <= n_dict ? t2[k] : (uint8_t)(row - n_dict - 1);
    }
  };
  __shared__ uint8_t s_present[A3_MAXCOV];
  if (!SPLIT) {
    fill(0, A.n_cov, (int)*A.n_dict, A.t2);
    if (A.cov_present)
      for (int k = threadIdx.x; k < A3_MAXCOV; k += A3_NT) s_present[k] = k < A.n_cov ? A.cov_present[k] : (uint8_t)0;
    __syncthreads();
  }
  const bool adapt_recs = !SPLIT && A.cov_present != nullptr;
  Apply3<SPLIT> B;
  const uint32_t len = A.len, bpr = (len + 15u) >> 4, sbytes = (len + 1u) >> 1;
  // lanes in groups of A.group that share whole reads (apply3_launch picks the group so that a read's last two blocks sit in one wave)
  const uint32_t grp = threadIdx.x / A.group, in_grp = threadIdx.x - grp * A.group, per_grp = A.group / bpr, RPI = per_grp * (A3_NT / A.group);
  const uint32_t slot_g = in_grp / bpr, jb = in_grp - slot_g * bpr, slot = grp * per_grp + slot_g;
  const bool lane_on = slot_g < per_grp;
  B.k0 = (jb == bpr - 1u && (len & 15u)) ? len - 16u : 16u * jb;  // the last block ends at the read's end
  B.qoff = slot * len + B.k0;
  B.soff = slot * sbytes + (B.k0 >> 1);
  B.ssh = 8u + 4u * (B.k0 & 1u);
  B.t1_at = lds_address(llut); B.t2_at = lds_address(llut) + (uint32_t)t1_bytes;
  B.w = (uint32_t)w; B.rows_w = (uint32_t)rows_w; B.qhi1 = (uint32_t)qhi1;
  B.lmax = A.lmax; B.max_cycle = A.max_cycle; B.lut = A.lut; B.err = 0;
  asm volatile("" : "+v"(B.qhi1));  // the SDWA form takes no inline constant / scalar here
  const uint8_t *seq_m1 = A.seq4 - 1;
  // The records of a launch: every staged read in staging order (a trip of a workgroup covers RPI consecutive reads, the workgroups
  // stride over the trips), or - SPLIT - the records sorted by covariate: the launch's trips, covariate after covariate, are shared
  // evenly among the workgroups; a workgroup takes a contiguous range of them and loads the tables of every covariate its range crosses.
  uint64_t n = SPLIT ? 0 : A.n;       // records of the part this workgroup works on (SPLIT: of the current covariate) ...
  uint64_t r_base = 0;                // ... and the first of them
  const uint64_t stride = SPLIT ? (uint64_t)RPI : (uint64_t)gridDim.x * RPI, r_mine = SPLIT ? 0 : (uint64_t)blockIdx.x * RPI;
  auto first_read = [&](uint64_t it) __attribute__((always_inline)) -> uint64_t { return it * stride + r_mine; };
  const uint32_t roff = slot * 8u, ioff = slot * 4u;
  auto rec_load = [&](uint64_t it, u32x2 &z, uint32_t &zi) __attribute__((always_inline)) {
    const uint64_t r0 = first_read(it);
    if (lane_on && r0 + slot < n) {
      gload_x2(z, reinterpret_cast<const uint8_t *>(A.recs) + (r_base + r0) * 8u, roff);
      if (SPLIT) gload_x1(zi, reinterpret_cast<const uint8_t *>(A.ridx) + (r_base + r0) * 4u, ioff);
    } else z = (u32x2){0u, 0u};
  };
  // idx: the read's staging index (SPLIT: out of the record's index word; else the trip's first read + the lane's slot, folded into the
  // wave-uniform base)
  auto data_load = [&](uint64_t it, u32x2 rec, uint32_t idx, A3Data &d) __attribute__((always_inline)) -> bool {
    const uint64_t r0 = first_read(it);
    if (!(lane_on && r0 + slot < n) || !(rec.y & AR_ON)) {
      if (adapt_recs && rec.x == AR_NO_RG) B.err |= 32u;  // readGroupCovariate panics, bqsr.go:38
      return false;
    }
    if (adapt_recs && !s_present[rec.y & 0xFFu]) return false;  // read group absent from the tables, read untouched (:953-955)
    if (SPLIT) {
      gload_x4(d.q, (uint64_t)A.qual + (uint64_t)idx * len + B.k0);
      gload_x4(d.s, (uint64_t)seq_m1 + (uint64_t)idx * sbytes + (B.k0 >> 1));
    } else {
      gload_x4(d.q, A.qual + r0 * len, B.qoff);
      gload_x4(d.s, seq_m1 + r0 * sbytes, B.soff);
    }
    return true;
  };
  // the wait of a loop trip (gload.hpp); the record moves out of its buffer by copies that stay behind the wait.  Inside the loop the
  // youngest vector-memory instruction of the wave is the trip's store: it may stay in flight
  auto landed0 = [&](A3Data &d, const u32x2 &z, const uint32_t &zi, u32x2 &rec, uint32_t &ri) __attribute__((always_inline)) {
    gwait();
    asm volatile("" : "+v"(d.q), "+v"(d.s));
    rec = amov(z);
    if (SPLIT) ri = amov(zi);
  };
  auto landed = [&](A3Data &d, const u32x2 &z, const uint32_t &zi, u32x2 &rec, uint32_t &ri) __attribute__((always_inline)) {
    gwait_but<1>();
    asm volatile("" : "+v"(d.q), "+v"(d.s));
    rec = amov(z);
    if (SPLIT) ri = amov(zi);
  };
  const uint64_t my_dump = (uint64_t)(A.dump + ((uint64_t)blockIdx.x * A3_NT + threadIdx.x) * 16u);
  // one block: the look-ups if the lane has one, then the store EVERY lane issues (exactly one per trip, behind the trip's loads)
  auto work = [&](uint64_t it, bool on, u32x2 rec, uint32_t idx, const A3Data &d) __attribute__((always_inline)) {
    u32x4 o = (u32x4){0u, 0u, 0u, 0u};
    if (on) o = B.process(rec, d);
    const uint64_t mine = SPLIT ? (uint64_t)A.qual + (uint64_t)idx * len + B.k0 : (uint64_t)(A.qual + first_read(it) * len + B.qoff);
    const uint64_t at = on ? mine : my_dump;
    gstore_x4(at, o);
  };
  // records two trips ahead (z), data one trip ahead, two sets X / Y that swap roles
  A3Data dX, dY;
  dX.q = (u32x4){0, 0, 0, 0}; dX.s = dX.q;
  dY = dX;
  u32x2 rX, rY, rN, z = (u32x2){0u, 0u};  // records of the blocks in X / Y, of the next trip's block, the buffer in flight
  uint32_t iX = 0, iY = 0, iN = 0, zi = 0;  // SPLIT: their reads' staging indices
  bool onX, onY = false;
  // one pass of the pipeline over trips [0, n_trips) of the current part
  auto run = [&](uint64_t n_trips) __attribute__((always_inline)) {
    rec_load(0, z, zi);
    landed0(dX, z, zi, rX, iX);
    onX = data_load(0, rX, iX, dX);
    rec_load(1, z, zi);
    landed0(dX, z, zi, rN, iN);
#pragma unroll 1
    for (uint64_t it = 0; it < n_trips; it += 2) {
      {
        rY = rN; iY = iN;
        onY = data_load(it + 1, rY, iY, dY);
        rec_load(it + 2, z, zi);
        work(it, onX, rX, iX, dX);
        landed(dY, z, zi, rN, iN);
      }
      {
        rX = rN; iX = iN;
        onX = data_load(it + 2, rX, iX, dX);
        rec_load(it + 3, z, zi);
        work(it + 1, onY, rY, iY, dY);
        landed(dX, z, zi, rN, iN);
      }
    }
    gwait();
  };
  if (!SPLIT) {
    run((n + stride - 1) / stride);
  } else {
    // trips per covariate, their exclusive prefix sums (s_pre[k] for k >= n_cov = the total)
    __shared__ uint32_t s_cnt[A3_MAXCOV], s_pre[A3_MAXCOV + 1], s_wsum[4];
    {
      const uint32_t my_cnt = (int)threadIdx.x < A.n_cov ? A.cov_cnt[threadIdx.x] : 0u;
      const uint32_t my_trips = (my_cnt + RPI - 1u) / RPI;
      uint32_t incl = my_trips;
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t v = __shfl_up(incl, d, 64);
        if ((int)(threadIdx.x & 63u) >= d) incl += v;
      }
      if (threadIdx.x < (uint32_t)A3_MAXCOV && (threadIdx.x & 63u) == 63u) s_wsum[threadIdx.x >> 6] = incl;
      __syncthreads();
      if (threadIdx.x < (uint32_t)A3_MAXCOV) {
        uint32_t off = 0;
        for (uint32_t wv = 0; wv < (threadIdx.x >> 6); wv++) off += s_wsum[wv];
        s_cnt[threadIdx.x] = my_cnt;
        s_pre[threadIdx.x] = off + incl - my_trips;
        if (threadIdx.x == (uint32_t)A3_MAXCOV - 1u) s_pre[A3_MAXCOV] = off + incl;
      }
      __syncthreads();
    }
    const uint64_t trips_all = to_sgpr(s_pre[A3_MAXCOV]);
    const uint32_t my_lo = to_sgpr((uint32_t)(trips_all * blockIdx.x / gridDim.x)), my_hi = to_sgpr((uint32_t)(trips_all * (blockIdx.x + 1ull) / gridDim.x));
    uint32_t cov = 0;
    for (uint32_t a = 0, b = (uint32_t)A.n_cov; b - a > 1u;) {
      const uint32_t md = (a + b) >> 1;
      if (s_pre[md] <= my_lo) { a = md; cov = md; } else b = md;
    }
#pragma unroll 1
    for (; cov < (uint32_t)A.n_cov; cov++) {
      const uint32_t p0 = to_sgpr(s_pre[cov]);
      if (p0 >= my_hi) break;
      const uint32_t tr = to_sgpr(s_pre[cov + 1]) - p0, cnt_c = to_sgpr(s_cnt[cov]);
      const uint32_t t0 = my_lo > p0 ? my_lo - p0 : 0u, t1 = my_hi - p0 < tr ? my_hi - p0 : tr;
      if (t0 >= t1) continue;
      __syncthreads();  // (every wave is through with the previous covariate's tables)
      fill((int)cov, 1, (int)A.n_dict[cov], A.t2 + (size_t)cov * 256 * 17);
      __syncthreads();
      const uint64_t r_first = (uint64_t)t0 * RPI, r_end = (uint64_t)t1 * RPI < (uint64_t)cnt_c ? (uint64_t)t1 * RPI : (uint64_t)cnt_c;
      n = to_sgpr(r_end - r_first);
      r_base = to_sgpr((uint64_t)A.cov_off[cov] + r_first);
      run(t1 - t0);
    }
  }
  uint32_t my_err = B.err;
  if (__any(my_err != 0)) {
    for (int d = 32; d >= 1; d >>= 1) my_err |= __shfl_xor(my_err, d, 64);
    if ((threadIdx.x & 63) == 0) atomicOr(&A.err[0], my_err);
  }
}

// LDS of a launch: level 1 + room for 256 level-2 rows (their number is not read back in front of the launch); 1 = does not fit
int apply3_bytes(int n_cov, int n_qi, int lmax, size_t *dyn_out) {
  const size_t n1 = (size_t)n_cov * (size_t)(6 + n_qi + 1) * (size_t)(2 * lmax + 1);
  const size_t dyn = ((n1 + 15) & ~(size_t)15) + (size_t)256 * A3_ROW + 16;
  *dyn_out = dyn;
  return dyn + 512 <= 160 * 1024 ? 0 : 1;
}

int apply3_launch(elp_ctx *c, int max_cycle, const uint8_t *d_lut, const uint8_t *d_cov_present, const uint16_t *t1, const uint8_t *t2, const uint32_t *n_dict_dev,
                  int n_qi, int lmax, size_t dyn, bool split) {
  const uint64_t n = c->n;
  const uint32_t len = c->uniform_len, bpr = (len + 15u) >> 4;
  // reads packed over the whole workgroup unless that puts the last two blocks of some read (which overlap when the length is not a
  // multiple of 16) into different waves: then whole reads per wave
  uint32_t group = A3_NT;
  if ((len & 15u) && bpr > 1)
    for (uint32_t slot = 0; slot < A3_NT / bpr; slot++)
      if (((slot * bpr + bpr - 1u) & 63u) == 0) group = 64;
  const uint32_t rpi = (group / bpr) * (A3_NT / group);
  unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(3, (160 * 1024) / (dyn + 512 + (split ? 2304 : 0))));
  if (c->tune.apply_wgs >= 1 && c->tune.apply_wgs <= 3) per_cu = std::min(per_cu, (unsigned)c->tune.apply_wgs);
  const int grid = (int)std::min<uint64_t>((n + rpi - 1) / rpi, (uint64_t)c->n_cu * per_cu);
  uint2 *recs;
  const size_t dump_words = (size_t)grid * A3_NT * 2;  // 16 bytes per lane, in uint2 units
  // records | dump | (split) staging indices | counts per covariate, offsets, cursors
  const size_t rec_words = (n + 4 + 1) & ~(uint64_t)1;
  ELP_TRY(scratch(c, 5, rec_words + dump_words + (split ? (n + 1) / 2 + 2 * A3_MAXCOV + 8 + (size_t)A3_RBLOCKS * A3_MAXCOV / 2 : 0) + 8, &recs));
  uint8_t *dump = reinterpret_cast<uint8_t *>(recs + rec_words);
  uint32_t *ridx = reinterpret_cast<uint32_t *>(recs + rec_words + dump_words), *cw = ridx + ((n + 1) & ~(uint64_t)1);  // cw: [256] counts | [257] offsets | [256] cursors
  if (split) {
    ELP_HIP(c, hipMemsetAsync(cw, 0, (3 * A3_MAXCOV + 1) * sizeof(uint32_t), c->stream));
    // cw: ... | [A3_RBLOCKS][256] the workgroups' counts (written whole by the first pass)
    const uint64_t tiles = blocks_for(n, A3_RTILE);
    const unsigned rg = (unsigned)std::min<uint64_t>(tiles, A3_RBLOCKS);
    const uint32_t per = (uint32_t)((tiles + rg - 1) / rg);
    uint32_t *blk = cw + 3 * A3_MAXCOV + 8;
    ELP_LAUNCH(c, "bqsr_apply_cov_hist", k_apply_cov_hist, dim3(rg), dim3(256), 0, n, per, (const uint16_t *)c->rgid.p, (const uint16_t *)c->rg_cov.p, d_cov_present, cw, blk,
               c->err_flag.p);
    ELP_LAUNCH(c, "bqsr_apply_cov_offsets", k_apply_cov_offsets, dim3(1), dim3(1), 0, (const uint32_t *)cw, cw + A3_MAXCOV, cw + 2 * A3_MAXCOV + 1);
    ELP_LAUNCH(c, "bqsr_apply_records", k_apply_records_split, dim3(rg), dim3(256), 0, n, per, len, lmax, (const uint16_t *)c->flag.p, (const uint16_t *)c->rgid.p,
               (const uint16_t *)c->rg_cov.p, (const uint64_t *)c->qbounds.p, d_cov_present, (const uint32_t *)blk, cw + 2 * A3_MAXCOV + 1, recs, ridx, c->err_flag.p);
  }
  // the adapt stage's score kernel left the records (same length, same lmax: both are the one length of the staged reads)
  const bool adapt_recs = !split && c->adapted && c->apply_recs_valid && c->apply_recs_lmax == lmax && c->apply_recs.cap >= n;
  if (!split && !adapt_recs)
    ELP_LAUNCH(c, "bqsr_apply_records", k_apply_records, dim3(blocks_for(n, 256)), dim3(256), 0, n, len, lmax, (const uint16_t *)c->flag.p,
               (const uint16_t *)c->rgid.p, (const uint16_t *)c->rg_cov.p, (const uint64_t *)c->qbounds.p, d_cov_present, recs, c->err_flag.p);
  Apply3Args A{n, len, c->qual.p, c->seq4.p + elp_ctx::SEQ_FRONT, adapt_recs ? (const uint2 *)c->apply_recs.p : (const uint2 *)recs, d_lut, t1, t2, n_dict_dev, c->n_cov, n_qi,
               lmax, max_cycle, c->err_flag.p, dump, group, ridx, cw, cw + A3_MAXCOV, adapt_recs ? d_cov_present : (const uint8_t *)nullptr};
  if (split) {
    ELP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bqsr_apply3<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    ELP_LAUNCH(c, "bqsr_apply", k_bqsr_apply3<true>, dim3(grid), dim3(A3_NT), dyn, A);
  } else {
    ELP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bqsr_apply3<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    ELP_LAUNCH(c, "bqsr_apply", k_bqsr_apply3<false>, dim3(grid), dim3(A3_NT), dyn, A);
  }
  return 0;
}

}  // namespace elp
