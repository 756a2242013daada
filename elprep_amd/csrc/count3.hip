// count3.hip — the BQSR covariate count for read sets of one length (what a sequencer writes), round 3.
//
// Reference: BaseRecalibrator.Recalibrate's per-base loop (filters/bqsr.go:505-538) over the records the prologue kernels of bqsr.hip
// prepared (recalibrateAln, clipping, known-site bits).  Same tables as k_bqsr_count (bqsr.hip), which stays as the general kernel:
// ragged read lengths, reads longer than --max-cycle or 1022 bases, more quality slots than one table pass holds.
//
// What the measurements of this round said (profiles/r3m_isa_rate_probe.txt, profiles/r3o_count2_ab_and_ablations.txt):
//   - k_bqsr_count issues 0.51 vector instructions per base and lane-block; most of them re-derive, per 16-base block, facts of the
//     READ (window, cycle origin, reference mapping) and split 64-bit nibble words;
//   - a first rewrite that kept the group structure of flat.hpp (a workgroup stages 256-512 reads into LDS behind barriers, every lane
//     then walks ~5 blocks) was SLOWER although it issued fewer instructions per block: with short blocks the per-group costs - the
//     stage step's memory latency, three barriers, a pipeline that starts cold and re-synchronises sixteen waves every five blocks -
//     dominate; its loads alone (everything up to the eligible-base mask) took 2.8 of its 5.4 ms at 24 M reads = 3.8 TB/s: the
//     kernel's floor is the memory system, not the ALUs.
// Hence this shape:
//   - NO groups, no LDS staging, no barriers in the main loop: reads have one length, so offsets are arithmetic.  A workgroup trip
//     covers RPI = 1024 / (blocks per read) WHOLE reads; lane t is block t % bpr of read slot t / bpr for the whole kernel, so a lane's
//     offsets inside a trip's span of the columns are constants and every address is a wave-uniform base (scalar arithmetic) plus a
//     constant 32-bit lane offset; waves run free of each other and drift apart, so that one wave's loads overlap another's arithmetic;
//   - the per-read facts come as a 32-byte record (BqRec) the prologue kernels write: two 16-byte loads per block (ten lanes share a
//     record: L1 hits), fetched two blocks ahead; the block's data (QUAL, known-site bits, SEQ window, a 24-byte reference window that
//     also covers the neighbouring pieces of a read with indels) one block ahead;
//   - per base: one LDS look-up quality -> row, two 32-bit LDS atomics of a 0 / 1 value (cycle cell, context cell); the rare mismatch
//     counts are added by a short loop over the set bits of the block's mismatch word;
//   - the context cells are replicated over the LDS banks (cell * R + lane % R; rows are multiples of 32 words).
//   - a read that is one run of matches takes its known-site bits from the reference window (bit 2 of a nibble, k_ref_mark_sites in
//     bqsr.hip): no skip-column load for it; reads with indels or clipped windows carry RC_SKIPCOL and read the column as before.
// Row of the workgroup-private table (32-bit words), n_q + 3 rows per covariate (extra rows: bad quality, quality without a slot, not
// counted - as k_bqsr_count):
//   [0, 16 R)            context observations, cell cx replica r at cx * R + r
//   [16 R, 16 R + 16)    context mismatches
//   [16 R + 16, rsw)     cycle cells, observations | mismatches << 16, cycle index x at (17 x) >> 4
#include <algorithm>

#include "bqsr_common.hpp"
#include "gload.hpp"

namespace elp {

constexpr uint32_t N1 = 0x11111111u, C3 = 0x33333333u;
constexpr int C3_XROWS = 3, C3_PAD = 64, C3_NT = 1024;

template <int BYTE>
__device__ __forceinline__ uint32_t byte_shl2(uint32_t w, uint32_t two) {  // ((w >> 8 BYTE) & 0xFF) << 2 in one instruction
  uint32_t r;
  if (BYTE == 0) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(two), "v"(w));
  else if (BYTE == 1) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(r) : "v"(two), "v"(w));
  else if (BYTE == 2) asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(two), "v"(w));
  else asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3" : "=v"(r) : "v"(two), "v"(w));
  return r;
}
__device__ __forceinline__ uint32_t lds_read_u32(uint32_t at) { return *reinterpret_cast<const lds_u32_t *>((uintptr_t)at); }

// what a block's work needs from its read's record (BqRec)
struct BlkRec { uint32_t win, ctxw, t0, fl, bpk, dpk; };
// what a block's loads land in (the asm loads of gload.hpp write these registers directly)
struct BlkData {
  u32x4 q;         // QUAL bytes
  uint32_t skipw;  // 32 known-site bits from a byte boundary on
  u32x4 s;         // SEQ window: nibble n = base k0 - 2 + n (three words used)
  u32x4 w03;       // reference window words 0..3: nibble n = reference base E0 - parity + k0 - 16 + n
  u32x2 w45;       // words 4, 5
};

// Reads the record cannot describe: k_bqsr_count's logic on the BqDesc of the read (loads inside the block's work: rare)
__device__ __noinline__ uint64_t c3_ref_general(const uint4 *__restrict__ desc, const uint32_t *__restrict__ cigar, const uint32_t *__restrict__ cig_scratch,
                                               uint8_t *const *__restrict__ ref_seq, const int64_t *__restrict__ ref_seq_len, uint32_t r, uint64_t S, int blo, int bhi,
                                               int cbase) {
  const uint4 dx = desc[2 * (size_t)r], dy = desc[2 * (size_t)r + 1];
  const uint32_t dfl = (dy.w >> 8) & 0xFFu;
  const int32_t refid = (int32_t)dx.w;
  const uint8_t *__restrict__ rp = ref_seq[refid];
  const int64_t rlen = ref_seq_len[refid];
  const int32_t D0 = (int32_t)dx.x, D1 = (int32_t)dx.y, D2 = (int32_t)dx.z;
  const int b1 = (int)(dy.x & 0xFFFFu), b2 = (int)(dy.x >> 16);
  uint64_t R = 0;
  if (!(dfl & BQ_COMPLEX)) {
    const int B1 = b1 - cbase, B2 = b2 - cbase;
    {
      const int hi = bhi < B1 ? bhi : B1;
      if (blo < hi) R |= (D0 == BQ_NOREF ? S : ref_nibbles(rp, rlen, (int64_t)D0 + cbase)) & nib_fill(nib_range(blo, hi));
    }
    if (B1 < bhi) {
      const int lo = blo > B1 ? blo : B1, hi = bhi < B2 ? bhi : B2;
      if (lo < hi) R |= (D1 == BQ_NOREF ? S : ref_nibbles(rp, rlen, (int64_t)D1 + cbase)) & nib_fill(nib_range(lo, hi));
      if (B2 < bhi) {
        const int lo2 = blo > B2 ? blo : B2;
        if (lo2 < bhi) R |= (D2 == BQ_NOREF ? S : ref_nibbles(rp, rlen, (int64_t)D2 + cbase)) & nib_fill(nib_range(lo2, bhi));
      }
    }
  } else {
    const uint32_t *cg = ((dfl & BQ_CIG_SCRATCH) ? cig_scratch : cigar) + (uint32_t)D0;
    R = ref_nibbles_complex(cg, b1, (int64_t)D2, cbase, blo, bhi, rp, rlen, S);
  }
  return R;
}


// OTHER false: the launch over the class-1 segments (every record a run of matches with its known-site bits in the reference window: the
// piece logic, the skip-column path and the descriptor path are not compiled in); true: the launch over the other region
template <int RLOG, bool OTHER>
struct Count3 {
  // kernel arguments
  const uint8_t *__restrict__ qual;
  const uint8_t *__restrict__ seq_m1;  // SEQ column minus one byte (the window of a block starts one byte in front of it)
  const uint8_t *__restrict__ skipbits;
  const uint4 *__restrict__ desc;
  const uint32_t *__restrict__ cigar;
  const uint32_t *__restrict__ cig_scratch;
  uint8_t *const *__restrict__ ref_seq;
  const int64_t *__restrict__ ref_seq_len;
  unsigned long long *cycle_tbl, *ctx_tbl;
  uint32_t k0, nb;      // the lane's block inside its read: first base, number of bases (constant for the whole kernel)
  int n_cov, n_q, lmax, max_cycle, rsw;
  // LDS
  uint32_t qrow_at, spread_at;
  const uint8_t *slot_q;
  uint32_t *tbl;
  uint32_t rpc_bytes;
  int cov_fixed;  // >= 0: this workgroup counts ONE covariate (covariate-split records): the table holds that covariate's rows only
  int t_origin;
  uint32_t rep4, two;
  uint32_t err;

  // Issues the loads of the lane's block of one trip (asm loads: nothing waits here).  The records are compacted (RecOut): the read's
  // staging index comes out of the record and the block's addresses are per lane.  Returns whether the block has any base of the clipped
  // copy (else nothing is loaded).
  __device__ __forceinline__ bool load_data(const u32x4 &ra, const u32x4 &rb, bool on, uint32_t len, uint32_t sbytes, BlkData &d, BlkRec &f, uint32_t &qlow,
                                            uint32_t &idx) const {
    const uint32_t a = ra.z & 0xFFFFu, e = ra.z >> 16;
    f.win = ra.z; f.ctxw = ra.w; f.t0 = rb.x; f.fl = rb.y; f.bpk = rb.z; f.dpk = rb.w;
    idx = rec_idx(ra.y, rb.y, rb.w);
    const uint64_t bit = (uint64_t)idx * len + k0;
    qlow = (uint32_t)bit & 7u;
    if (!on || k0 + nb <= a || k0 >= e) return false;
    gload_x4(d.q, (uint64_t)qual + bit);
    d.skipw = 0u;
    if (OTHER && (rb.y & RC_SKIPCOL)) gload_x1(d.skipw, (uint64_t)skipbits + (bit >> 3));  // else: the flags of the reference window
    gload_x4(d.s, (uint64_t)seq_m1 + (uint64_t)idx * sbytes + (k0 >> 1));
    const uint64_t rp = ((uint64_t)ra.x | ((uint64_t)(ra.y & 0xFFFFu) << 32)) + ((rb.y & RC_GENERAL) ? 0u : (k0 >> 1));
    gload_x4(d.w03, rp);
    gload_x2(d.w45, rp + 16);
    return true;
  }

  // 16 nibbles from nibble n0 (0 .. 31) of the 48-nibble reference window.  Two conditional word shifts (by 16 and by 8 nibbles), then
  // one funnel shift.  (The words pass through registers the optimiser cannot look into: a select between MEMBERS of the block's data
  // becomes an indexed load, and the whole object would move to scratch memory.)
  __device__ __forceinline__ static uint64_t win_extract(const BlkData &d, int n0) {
    uint32_t x0 = d.w03.x, x1 = d.w03.y, x2 = d.w03.z, x3 = d.w03.w, x4 = d.w45.x, x5 = d.w45.y;
    asm("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5));
    const bool s16 = n0 & 16, s8 = n0 & 8;
    const uint32_t y0 = s16 ? x2 : x0, y1 = s16 ? x3 : x1, y2 = s16 ? x4 : x2, y3 = s16 ? x5 : x3;
    const uint32_t z0 = s8 ? y1 : y0, z1 = s8 ? y2 : y1, z2 = s8 ? y3 : y2;
    const uint32_t sh = 4u * (uint32_t)(n0 & 7);
    return (uint64_t)__builtin_amdgcn_alignbit(z1, z0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(z2, z1, sh) << 32);
  }
  // reference nibbles of a block of a read with up to four pieces (record form): piece j covers clipped bases [b_j, b_j+1), its
  // reference base for clipped base c is E0 + delta_j + a + c (delta_0 = 0 along the first piece with a reference), or none
  __device__ __forceinline__ static uint64_t ref_pieces(const BlkData &d, uint32_t fl, uint32_t bpk, uint32_t dpk, uint64_t S, int blo, int bhi, int cbase) {
    const int par = (fl & RC_PAR) ? 1 : 0;
    const uint32_t nr = fl >> 24;
    // the first piece with a reference has delta 0; pieces in front of it are insertions
    const int f = !(nr & 1u) ? 0 : (!(nr & 2u) ? 1 : (!(nr & 4u) ? 2 : 3));
    uint64_t R = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int bj = j == 0 ? 0 : (int)((bpk >> (10 * (j - 1))) & 1023u), bn = j == 3 ? 1023 : (int)((bpk >> (10 * j)) & 1023u);
      if (j > 0 && bj == 1023) break;
      int lo = bj - cbase, hi = bn == 1023 ? bhi : bn - cbase;
      lo = lo > blo ? lo : blo;
      hi = hi < bhi ? hi : bhi;
      if (lo < hi) {
        const int dj = j == 0 ? 0 : (int)(int8_t)((dpk >> (8 * (j - 1))) & 0xFFu);
        // deltas are relative to piece f; for j < f the piece is an insertion (f > 0 means pieces 0 .. f-1 have no reference)
        const uint64_t piece = ((nr >> j) & 1u) ? S : win_extract(d, 16 + par + (j == f ? 0 : dj));
        R |= piece & nib_fill(nib_range(lo, hi));
      }
    }
    return R;
  }
  template <int I>
  __device__ __forceinline__ uint32_t row_of(uint32_t qw) const { return lds_read_u32(qrow_at + byte_shl2<(I & 3)>(qw, two)); }
  // One base: two atomics of a 0 / 1 value into the row `ro` of its quality - a base that is not counted adds zero to whatever cell it
  // points at.  t = cell position of the base in sixteenths of a word (row base of the covariate included).
  template <int I>
  __device__ __forceinline__ void base(uint32_t ro, uint32_t Fh, uint32_t FVh, uint32_t CXh, uint32_t t, uint32_t lrepC) {
    constexpr int sh = 4 * (I & 7);
    lds_add_u32(lshl_add_u32<2>(t >> 4, ro), bfe_u32<sh, 1>(Fh));
    lds_add_u32(ro + lshl_add_u32<RLOG + 2>(bfe_u32<sh, 4>(CXh), lrepC), bfe_u32<sh, 1>(FVh));
  }
  // the mismatches of eight bases (rare): cycle cell += 1 << 16, context-mismatch cell += 1.  Eh / FVh / CXh: the half's nibble words,
  // qa / qb its QUAL words, tb the cell position of the half's first base
  __device__ __forceinline__ void mismatches(uint32_t Eh, uint32_t FVh, uint32_t CXh, uint32_t qa, uint32_t qb, uint32_t tb, int st, uint32_t rowb) {
    constexpr uint32_t cxm = (uint32_t)(16 << RLOG) * 4u;  // byte offset of the context-mismatch cells in a row
    while (Eh) {
      const uint32_t b4 = (uint32_t)__builtin_ctz(Eh);
      const uint32_t b = b4 >> 2;
      Eh &= Eh - 1u;
      const uint32_t q = (((b & 4u) ? qb : qa) >> (8u * (b & 3u))) & 0xFFu;
      if (q < 6u) continue;  // not counted (its row is thrown away): no look-up, no atomics
      const uint32_t ro = lds_read_u32(qrow_at + 4u * q);
      const uint32_t t = tb + (uint32_t)(st * (int)b);
      lds_add_u32(ro + ((t >> 4) << 2), 0x10000u);
      if ((FVh >> b4) & 1u) lds_add_u32(ro + rowb + cxm + (((CXh >> b4) & 15u) << 2), 1u);
    }
  }

  __device__ __forceinline__ void process(const BlkRec &f, uint32_t r, const BlkData &d, uint32_t qlow) {
    const int a = (int)(f.win & 0xFFFFu), e = (int)(f.win >> 16);
    int blo = a - (int)k0, bhi = e - (int)k0;
    blo = blo > 0 ? blo : 0;
    bhi = bhi < (int)nb ? bhi : (int)nb;
    const uint32_t fl = f.fl;
    const bool rev = fl & RC_REV;
    // SEQ: S = the block's bases, N = their predecessors in sequencing direction (base - 1 forward, base + 1 reverse)
    const uint32_t ns = rev ? 12u : 4u;
    const uint32_t S_lo = __builtin_amdgcn_alignbit(d.s.y, d.s.x, 8), S_hi = __builtin_amdgcn_alignbit(d.s.z, d.s.y, 8);
    const uint32_t N_lo = __builtin_amdgcn_alignbit(d.s.y, d.s.x, ns), N_hi = __builtin_amdgcn_alignbit(d.s.z, d.s.y, ns);
    const uint32_t oS_lo = (S_lo >> 3) & N1, oS_hi = (S_hi >> 3) & N1;  // not A / C / G / T
    const uint32_t oN_lo = (N_lo >> 3) & N1, oN_hi = (N_hi >> 3) & N1;
    // reference nibbles of a read that is one run of matches: nibble 16 + parity of the window = words 2, 3, 4.  Bit 2 of a nibble =
    // the base lies in a known site (k_ref_mark_sites, bqsr.hip)
    uint32_t R_lo = 0, R_hi = 0, k_lo = 0, k_hi = 0;
    const bool one_run = !OTHER || !(fl & (RC_MULTI | RC_GENERAL));
    if (one_run) {
      const uint32_t sh = (fl & RC_PAR) ? 4u : 0u;
      R_lo = __builtin_amdgcn_alignbit(d.w03.w, d.w03.z, sh);
      R_hi = __builtin_amdgcn_alignbit(d.w45.x, d.w03.w, sh);
      k_lo = (R_lo >> 2) & N1;
      k_hi = (R_hi >> 2) & N1;
    }
    if (OTHER && (fl & RC_SKIPCOL)) {  // known-site bits from the skip column -> nibble flags (LDS table: bit i of a byte -> bit 4 i)
      const uint32_t sk = d.skipw >> qlow;
      k_lo |= lds_read_u32(spread_at + ((sk & 0xFFu) << 2));
      k_hi |= lds_read_u32(spread_at + ((sk >> 6) & 0x3FCu));
    }
    const uint64_t inw = nib_range(blo, bhi);
    const uint32_t F_lo = (uint32_t)inw & ~(oS_lo | k_lo), F_hi = (uint32_t)(inw >> 32) & ~(oS_hi | k_hi);
    if ((F_lo | F_hi) == 0u) return;
    // SNP events (computeSnpEvents, bqsr.go:254-285): read nibble vs reference nibble
    if (!one_run) {
      const uint64_t S = (uint64_t)S_lo | ((uint64_t)S_hi << 32);
      const uint64_t R = (fl & RC_GENERAL) ? c3_ref_general(desc, cigar, cig_scratch, ref_seq, ref_seq_len, r, S, blo, bhi, (int)k0 - a) : ref_pieces(d, fl, f.bpk, f.dpk, S, blo, bhi, (int)k0 - a);
      R_lo = (uint32_t)R;
      R_hi = (uint32_t)(R >> 32);
    }
    const uint32_t x_lo = S_lo ^ R_lo, x_hi = S_hi ^ R_hi;
    const uint32_t E_lo = (x_lo | (x_lo >> 1) | (x_lo >> 3)) & F_lo & N1, E_hi = (x_hi | (x_hi >> 1) | (x_hi >> 3)) & F_hi & N1;
    // context (bqsr.go:87-146)
    const uint64_t cw = nib_range_clamped((int)(f.ctxw & 0xFFFFu) - (int)k0, (int)(f.ctxw >> 16) - (int)k0);
    const uint32_t FV_lo = F_lo & (uint32_t)cw & ~oN_lo, FV_hi = F_hi & (uint32_t)(cw >> 32) & ~oN_hi;
    const uint32_t rm = rev ? 0xFFFFFFFFu : 0u;
    const uint32_t CX_lo = ((N_lo & C3) | ((S_lo & C3) << 2)) ^ rm, CX_hi = ((N_hi & C3) | ((S_hi & C3) << 2)) ^ rm;
    const int st = (fl & RC_NEG) ? -17 : 17;
    const uint32_t rowb = cov_fixed >= 0 ? 0u : __umul24(fl & 0xFFu, rpc_bytes);
    const uint32_t tb = (uint32_t)((int)f.t0 + t_origin + st * (int)k0) + 4u * rowb;
    const uint32_t lrepC = rowb + rep4;
    const uint32_t ust = (uint32_t)st;
#define ELP_B3(I, RO, FH, FVH, CXH) base<I>(RO, FH, FVH, CXH, tb + (uint32_t)(I) * ust, lrepC)
    {
      const uint32_t r0 = row_of<0>(d.q.x), r1 = row_of<1>(d.q.x), r2 = row_of<2>(d.q.x), r3 = row_of<3>(d.q.x);
      const uint32_t r4 = row_of<4>(d.q.y), r5 = row_of<5>(d.q.y), r6 = row_of<6>(d.q.y), r7 = row_of<7>(d.q.y);
      ELP_B3(0, r0, F_lo, FV_lo, CX_lo); ELP_B3(1, r1, F_lo, FV_lo, CX_lo); ELP_B3(2, r2, F_lo, FV_lo, CX_lo); ELP_B3(3, r3, F_lo, FV_lo, CX_lo);
      ELP_B3(4, r4, F_lo, FV_lo, CX_lo); ELP_B3(5, r5, F_lo, FV_lo, CX_lo); ELP_B3(6, r6, F_lo, FV_lo, CX_lo); ELP_B3(7, r7, F_lo, FV_lo, CX_lo);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const uint32_t r0 = row_of<8>(d.q.z), r1 = row_of<9>(d.q.z), r2 = row_of<10>(d.q.z), r3 = row_of<11>(d.q.z);
      const uint32_t r4 = row_of<12>(d.q.w), r5 = row_of<13>(d.q.w), r6 = row_of<14>(d.q.w), r7 = row_of<15>(d.q.w);
      ELP_B3(8, r0, F_hi, FV_hi, CX_hi); ELP_B3(9, r1, F_hi, FV_hi, CX_hi); ELP_B3(10, r2, F_hi, FV_hi, CX_hi); ELP_B3(11, r3, F_hi, FV_hi, CX_hi);
      ELP_B3(12, r4, F_hi, FV_hi, CX_hi); ELP_B3(13, r5, F_hi, FV_hi, CX_hi); ELP_B3(14, r6, F_hi, FV_hi, CX_hi); ELP_B3(15, r7, F_hi, FV_hi, CX_hi);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef ELP_B3
    {
      if (E_lo) mismatches(E_lo, FV_lo, CX_lo, d.q.x, d.q.y, tb, st, rowb);
      if (E_hi) mismatches(E_hi, FV_hi, CX_hi, d.q.z, d.q.w, tb + 8u * ust, st, rowb);
    }
  }

  // adds the private table into the dense int64 tables (cycle: [cov][94][2*max_cycle+1][2], context: [cov][94][16][2]) and clears it
  __device__ __forceinline__ void flush() {
    __syncthreads();
    const int rpc = n_q + C3_XROWS, rows = (cov_fixed >= 0 ? 1 : n_cov) * rpc;
    const int ncyc_l = 2 * lmax + 1, ncyc_g = 2 * max_cycle + 1;
    constexpr int R = 1 << RLOG, cyc_w = 16 * R + 16;
    for (int k = threadIdx.x; k < rows * ncyc_l; k += C3_NT) {
      const int row = k / ncyc_l, x = k - row * ncyc_l;
      uint32_t *cell = &tbl[row * rsw + cyc_w + ((17 * x) >> 4)];
      const uint32_t v = *cell;
      if (v) {
        *cell = 0;
        const int cov = cov_fixed >= 0 ? cov_fixed : row / rpc, slot = row % rpc;
        const int cyc = x - lmax;
        if (slot >= n_q) {
          err |= slot == n_q ? 8u : (slot == n_q + 1 ? 128u : 0u);
        } else if (cyc >= -max_cycle && cyc <= max_cycle) {
          const int q = slot_q[slot];
          unsigned long long *g = cycle_tbl + (((size_t)cov * ELP_NQUAL + q) * ncyc_g + (size_t)(cyc + max_cycle)) * 2;
          atomicAdd(g, (unsigned long long)(v & 0xFFFFu));
          if (v >> 16) atomicAdd(g + 1, (unsigned long long)(v >> 16));
        }
      }
    }
    for (int k = threadIdx.x; k < rows * 16; k += C3_NT) {
      const int row = k >> 4, cx = k & 15;
      uint32_t *obs = &tbl[row * rsw + cx * R];
      unsigned long long o = 0;
      for (int r = 0; r < R; r++) { o += obs[r]; obs[r] = 0; }
      uint32_t *mc = &tbl[row * rsw + 16 * R + cx];
      const uint32_t m = *mc;
      *mc = 0;
      if (o | m) {
        const int cov = cov_fixed >= 0 ? cov_fixed : row / rpc, slot = row % rpc;
        if (slot < n_q) {
          const int q = slot_q[slot];
          // cx = prev | cur << 2 is exactly (key >> 4) & 15 of keyFromContext (bqsr.go:64-76)
          unsigned long long *g = ctx_tbl + (((size_t)cov * ELP_NQUAL + q) * ELP_NCTX + (size_t)cx) * 2;
          if (o) atomicAdd(g, o);
          if (m) atomicAdd(g + 1, (unsigned long long)m);
        }
      }
    }
    __syncthreads();
  }
};

template <int RLOG, bool OTHER>
__global__ __launch_bounds__(C3_NT) void k_bqsr_count3(Count3Args A, QMap qm) {
  __shared__ uint32_t qrow[256];
  __shared__ uint32_t spread8[256];
  __shared__ uint8_t slot_q[96];
  extern __shared__ __attribute__((aligned(16))) uint32_t tbl[];
  const int n_all = (A.ncs ? 1 : A.n_cov) * (A.n_q + C3_XROWS) * A.rsw + C3_PAD;
  const uint32_t tbl_at = lds_address(tbl);
  for (int k = threadIdx.x; k < n_all; k += C3_NT) tbl[k] = 0;
  for (int q = threadIdx.x; q < 256; q += C3_NT) {
    int row;
    if (q < 6) row = A.n_q + 2;                // not counted (bqsr.go:301-305)
    else if (q >= ELP_NQUAL) row = A.n_q;      // bad quality
    else {
      const uint8_t s = qm.slot[q];
      row = s == 255 ? A.n_q + 2 : (s == 254 ? A.n_q + 1 : (int)s);  // counted in another pass / not in the table
      if (s < 254) slot_q[s] = (uint8_t)q;
    }
    qrow[q] = tbl_at + (uint32_t)(row * A.rsw) * 4u;
    uint32_t sp = 0;
    for (int b = 0; b < 8; b++) sp |= ((uint32_t)(q >> b) & 1u) << (4 * b);
    spread8[q] = sp;
  }
  __syncthreads();
  Count3<RLOG, OTHER> B;
  B.qual = A.qual; B.seq_m1 = A.seq4 - 1; B.skipbits = A.skipbits; B.desc = A.desc; B.cigar = A.cigar; B.cig_scratch = A.cig_scratch;
  B.ref_seq = A.ref_seq; B.ref_seq_len = A.ref_seq_len; B.cycle_tbl = A.cycle_tbl; B.ctx_tbl = A.ctx_tbl;
  B.n_cov = A.n_cov; B.n_q = A.n_q; B.lmax = A.lmax; B.max_cycle = A.max_cycle; B.rsw = A.rsw;
  B.qrow_at = lds_address(qrow); B.spread_at = lds_address(spread8); B.slot_q = slot_q; B.tbl = tbl;
  B.rpc_bytes = (uint32_t)((A.n_q + C3_XROWS) * A.rsw) * 4u;
  B.t_origin = 16 * ((16 << RLOG) + 16) + 17 * A.lmax;
  B.rep4 = 4u * ((threadIdx.x & 63u) & ((1u << RLOG) - 1u));
  B.two = 2u;
  asm volatile("" : "+v"(B.two));  // keep it in a register: the SDWA form takes no inline constant
  B.err = 0;
  B.cov_fixed = A.ncs ? 0 : -1;

  // A trip of a workgroup covers RPI whole records of ONE segment; lane t works on block t % bpr of record slot t / bpr in every trip (the
  // last 1024 - RPI * bpr lanes idle).  Round 5: the launch's trips - segment after segment - are shared evenly among the workgroups: a
  // workgroup takes a contiguous range of them and meets the segments that range crosses one after the other (with the covariate split a
  // segment is one covariate: the private table is flushed and reused) - whatever the segments' sizes and however many there are.
  const uint32_t len = A.len, bpr = (len + 15u) >> 4, sbytes = (len + 1u) >> 1, RPI = C3_NT / bpr;
  const uint32_t slot = threadIdx.x / bpr, jb = threadIdx.x - slot * bpr;
  const bool lane_on = slot < RPI;
  B.k0 = 16u * jb;
  B.nb = len - B.k0 < 16u ? len - B.k0 : 16u;
  // trips per segment, their exclusive prefix sums (s_pre[k] for k >= nseg = the total)
  __shared__ uint32_t s_cnt[C3_MAXSEG], s_pre[C3_MAXSEG + 1], s_wsum[4];
  {
    const uint32_t my_cnt = threadIdx.x < A.nseg ? A.seg_cnt[(size_t)threadIdx.x * A.cnt_stride] : 0u;
    const uint32_t my_trips = (my_cnt + RPI - 1u) / RPI;
    uint32_t incl = my_trips;
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t v = __shfl_up(incl, d, 64);
      if ((int)(threadIdx.x & 63u) >= d) incl += v;
    }
    if (threadIdx.x < (uint32_t)C3_MAXSEG && (threadIdx.x & 63u) == 63u) s_wsum[threadIdx.x >> 6] = incl;
    __syncthreads();
    if (threadIdx.x < (uint32_t)C3_MAXSEG) {
      uint32_t off = 0;
      for (uint32_t w = 0; w < (threadIdx.x >> 6); w++) off += s_wsum[w];
      s_cnt[threadIdx.x] = my_cnt;
      s_pre[threadIdx.x] = off + incl - my_trips;
      if (threadIdx.x == (uint32_t)C3_MAXSEG - 1u) s_pre[C3_MAXSEG] = off + incl;
    }
    __syncthreads();
  }
  const uint64_t trips_all = to_sgpr(s_pre[C3_MAXSEG]);
  const uint32_t my_lo = to_sgpr((uint32_t)(trips_all * blockIdx.x / gridDim.x)), my_hi = to_sgpr((uint32_t)(trips_all * (blockIdx.x + 1ull) / gridDim.x));
  // a cycle cell (16 | 16 bits) takes at most one count per read
  const uint32_t flush_every = 30000u / (RPI + 1u) + 1u;
  uint64_t n = 0;                     // records of the current segment's part this workgroup works on ...
  const uint8_t *seg_recs = nullptr;  // ... and the first of them
  auto first_read = [&](uint64_t it) __attribute__((always_inline)) -> uint64_t { return it * RPI; };
  const uint32_t roff = slot * 32u;  // the lane's record inside a trip's span of the record array
  auto rec_load = [&](uint64_t it, u32x4 &za, u32x4 &zb) __attribute__((always_inline)) {
    const uint64_t r0 = first_read(it);
    if (lane_on && r0 + slot < n) {
      const uint8_t *rt = seg_recs + r0 * 32u;
      gload_x4(za, rt, roff);
      gload_x4(zb, rt, roff + 16u);
    }
  };
  auto data_load = [&](uint64_t it, const u32x4 &za, const u32x4 &zb, BlkData &d, BlkRec &f, uint32_t &qlow, uint32_t &idx) __attribute__((always_inline)) -> bool {
    const uint64_t r0 = first_read(it);
    const bool on = lane_on && r0 + slot < n;
    return B.load_data(za, zb, on, len, sbytes, d, f, qlow, idx);
  };
  // the ONE wait of a loop trip: everything issued so far has landed.  The data registers are operands of an empty statement behind the
  // wait (their uses stay behind it); the record moves out of its buffer by copies that stay behind the wait (gload.hpp)
  auto landed = [&](BlkData &d, const u32x4 &za, const u32x4 &zb, u32x4 &ra, u32x4 &rb) __attribute__((always_inline)) {
    gwait();
    asm volatile("" : "+v"(d.q), "+v"(d.skipw), "+v"(d.s), "+v"(d.w03), "+v"(d.w45));
    ra = amov(za);
    rb = amov(zb);
  };
  // Pipeline: records two trips ahead (buffer Z), data one trip ahead, two sets X / Y that swap roles (no copies).  In a trip: issue the
  // next trip's data loads and the record loads of the trip behind it, work on the current block, then wait for what was issued.
  BlkData dX, dY;
  dX.q = (u32x4){0, 0, 0, 0}; dX.s = dX.q; dX.w03 = dX.q; dX.w45 = (u32x2){0, 0}; dX.skipw = 0;
  dY = dX;
  BlkRec fX, fY;
  u32x4 za = (u32x4){0, 0, 0, 0}, zb = za, na = za, nb4 = za;  // record buffer in flight; record of the next trip's block
  uint32_t qlX = 0, qlY = 0, iX = 0, iY = 0;  // bit phase of the block's skip-column word; staging index of the block's read
  bool onX, onY = false;
  uint32_t since = 0;
  // the first segment of the range: the last one whose first trip is not behind my_lo
  uint32_t seg = 0;
  for (uint32_t a = 0, b = A.nseg; b - a > 1u;) {
    const uint32_t md = (a + b) >> 1;
    if (s_pre[md] <= my_lo) { a = md; seg = md; } else b = md;
  }
#pragma unroll 1
  for (; seg < A.nseg; seg++) {
    const uint32_t p0 = to_sgpr(s_pre[seg]);
    if (p0 >= my_hi) break;
    const uint32_t tr = to_sgpr(s_pre[seg + 1]) - p0, cnt_s = to_sgpr(s_cnt[seg]);
    const uint32_t t0 = my_lo > p0 ? my_lo - p0 : 0u, t1 = my_hi - p0 < tr ? my_hi - p0 : tr;
    if (t0 >= t1) continue;
    const uint64_t r_first = (uint64_t)t0 * RPI, r_end = (uint64_t)t1 * RPI < (uint64_t)cnt_s ? (uint64_t)t1 * RPI : (uint64_t)cnt_s;
    n = to_sgpr(r_end - r_first);
    seg_recs = reinterpret_cast<const uint8_t *>(to_sgpr(reinterpret_cast<uint64_t>(A.srecs) + ((uint64_t)A.seg_base[seg] + r_first) * 32u));
    const uint64_t n_trips = t1 - t0;  // every wave of the workgroup makes the same number of trips (flush barriers stay uniform)
    B.cov_fixed = A.ncs ? (int)(seg & ((uint32_t)A.ncs - 1u)) : -1;
    rec_load(0, za, zb);
    landed(dX, za, zb, na, nb4);
    onX = data_load(0, na, nb4, dX, fX, qlX, iX);
    rec_load(1, za, zb);
    landed(dX, za, zb, na, nb4);
#pragma unroll 1
    for (uint64_t it = 0; it < n_trips; it += 2) {
      {  // trip it: work on X; data of trip it + 1 -> Y; record of trip it + 2 -> Z
        onY = data_load(it + 1, na, nb4, dY, fY, qlY, iY);
        rec_load(it + 2, za, zb);
        if (onX) B.process(fX, iX, dX, qlX);
        landed(dY, za, zb, na, nb4);
        if (++since == flush_every) { B.flush(); since = 0; }
      }
      {  // trip it + 1: work on Y; data of trip it + 2 -> X; record of trip it + 3 -> Z
        onX = data_load(it + 2, na, nb4, dX, fX, qlX, iX);
        rec_load(it + 3, za, zb);
        if (onY) B.process(fY, iY, dY, qlY);
        landed(dX, za, zb, na, nb4);
        if (++since == flush_every) { B.flush(); since = 0; }
      }
    }
    if (A.ncs) {  // the next segment is another covariate: its counts go into a cleared table
      gwait();
      B.flush();
      since = 0;
    }
  }
  gwait();
  B.flush();
  uint32_t my_err = B.err;
  if (__any(my_err != 0)) {
    for (int d = 32; d >= 1; d >>= 1) my_err |= __shfl_xor(my_err, d, 64);
    if ((threadIdx.x & 63) == 0) atomicOr(&A.err[0], my_err);
  }
}

// Launch plan: one workgroup of 1024 threads per CU around one table; the context cells are replicated as often as the CU's LDS allows.
// Returns 1 if the tables of this pass do not fit (the caller uses k_bqsr_count).
int count3_plan(int n_cov, int n_q, int lmax, int *rsw_out, int *rlog_out, size_t *dyn_out, int force_rlog) {
  const size_t lds_cu = 160 * 1024, static_lds = 1024 + 1024 + 96 + 256 + (2 * C3_MAXSEG + 8) * 4;
  const int ncw = ((17 * 2 * lmax) >> 4) + 2;
  const size_t rows = (size_t)n_cov * (size_t)(n_q + C3_XROWS);
  for (int rlog = 5; rlog >= 1; rlog--) {
    if (force_rlog >= 0 && force_rlog != rlog) continue;  // elp_set_tuning "count3_rlog": measurements only
    const int rsw = ((16 << rlog) + 16 + ncw + 31) & ~31;
    const size_t dyn = (rows * (size_t)rsw + C3_PAD) * 4;
    if (dyn + static_lds <= lds_cu && rows * (size_t)rsw * 4 < (1u << 22)) {
      *rsw_out = rsw; *rlog_out = rlog; *dyn_out = dyn;
      return 0;
    }
  }
  return 1;
}

int count3_launch(elp_ctx *c, const Count3Args &A, const QMap &qm, size_t dyn) {
  // one workgroup per CU; they share the launch's trips evenly
  const int grid = c->n_cu;
#define ELP_C3K(RL, OT)                                                                                                                          \
  do {                                                                                                                                           \
    ELP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bqsr_count3<RL, OT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)); \
    ELP_LAUNCH(c, "bqsr_count", (k_bqsr_count3<RL, OT>), dim3(grid), dim3(C3_NT), dyn, A, qm);                                                  \
    return 0;                                                                                                                                    \
  } while (0)
#define ELP_C3(RL) do { if (A.other) ELP_C3K(RL, true); else ELP_C3K(RL, false); } while (0)
  switch (A.rlog) { case 5: ELP_C3(5); case 4: ELP_C3(4); case 3: ELP_C3(3); case 2: ELP_C3(2); default: ELP_C3(1); }
#undef ELP_C3K
#undef ELP_C3
}

}  // namespace elp
