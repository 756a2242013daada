// bqsr_common.hpp — what the BQSR translation units share: the per-record descriptor the prologue kernels leave for the count
// kernels, the 4-bit packed reference windows, LDS atomics and the one-instruction helpers.  (Moved out of bqsr.hip in round 3, when
// the second count kernel - count2.hip - arrived.)
#pragma once
#include "bqsr_dev.hpp"
#include "flat.hpp"

namespace elp {

constexpr int MAX_DESC_READ = 65535;  // u16 fields of the record descriptor


// per-record descriptor produced by the prologue (32 bytes, staged into LDS by k_bqsr_count)
//
// The clipped working copy of an eligible record is the base window [a, a+len) of the original read, and the mapping of its
// bases to the reference (computeSnpEvents, bqsr.go:254-285) is piecewise: clipped base c in piece k = [B_k, B_k+1) (B_0 = 0,
// B_1 = b1, B_2 = b2, B_3 = infinity) lies at 0-based reference index D_k + c, or has no reference base (insertion) if
// D_k == BQ_NOREF.  A clipped CIGAR of the form M, M I M, M D M, I M ... needs at most three pieces; records that need more
// are flagged BQ_COMPLEX and walk their CIGAR in the kernel (D0 = CIGAR index, b1 = op count, D2 = POS - 1).
constexpr int32_t BQ_NOREF = INT32_MIN;
enum : uint8_t { BQ_ELIGIBLE = 1, BQ_REVERSED = 2, BQ_LAST = 4, BQ_CIG_SCRATCH = 8, BQ_COMPLEX = 16 };
struct __attribute__((aligned(16))) BqDesc {
  int32_t D0, D1, D2;
  int32_t refid;
  uint16_t b1, b2;
  uint16_t a;      // first surviving base (original read coordinates)
  uint16_t len;    // surviving bases; 0 = record contributes nothing
  uint16_t left;   // low-quality-tail bounds inside the surviving window (left > right: everything masked)
  uint16_t right;  // 0xFFFF = -1
  uint8_t cov;     // read-group covariate id
  uint8_t fl;
  uint16_t pad;
};
static_assert(sizeof(BqDesc) == 32, "BqDesc is staged as two 16-byte words");


constexpr int REF_LDS = 256;      // contigs whose per-contig facts (pointers, lengths) are kept in LDS by the BQSR kernels

constexpr int64_t REF_PAD = 32;  // bytes of "other" (0x88) after the packed bases of a contig
constexpr uint64_t REF_OTHER = 0x8888888888888888ull;

// Reference window of a block: ref_load ISSUES the load of the 32 packed bases around reference index jb (clamped into the
// contig; its packed bases are followed by REF_PAD bytes of "other") and returns the nibble shift for ref_unpack, REF_NONE if
// nothing of [jb, jb+16) lies inside the contig.  ref_unpack: nibble b = reference base jb + b ("other" outside [0, rlen)).
constexpr int REF_NONE = 99;
__device__ __forceinline__ int ref_load(const uint8_t *__restrict__ rp, int64_t rlen, int64_t jb, uint64_t &v0, uint64_t &v1) {
  const bool valid = jb < rlen && jb > -16;
  int64_t jw = jb < 0 ? 0 : jb;
  jw = jw > rlen ? rlen : jw;
  jw &= ~(int64_t)1;
  __builtin_memcpy(&v0, rp + (jw >> 1), 8);
  __builtin_memcpy(&v1, rp + (jw >> 1) + 8, 8);
  return valid ? (int)(jb - jw) : REF_NONE;  // -15 .. 1
}
__device__ __forceinline__ uint64_t ref_unpack(uint64_t v0, uint64_t v1, int sn) {
  if (sn == 0 || sn == 1) {  // the common case (reference index >= 0): two 32-bit funnel shifts
    const uint32_t w0 = (uint32_t)v0, w1 = (uint32_t)(v0 >> 32), w2 = (uint32_t)v1, sh = 4u * (uint32_t)sn;
    return (uint64_t)__builtin_amdgcn_alignbit(w1, w0, sh) | ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32);
  }
  if (sn == REF_NONE) return REF_OTHER;
  // window starts at the contig's first base but the block starts before it (sn < 0): positions before the contig are "other"
  return nib_ext(v0, v1, sn) | (REF_OTHER & ~(NIBF << (4 * -sn)));
}
__device__ __forceinline__ uint64_t ref_nibbles(const uint8_t *__restrict__ rp, int64_t rlen, int64_t jb) {
  uint64_t v0, v1;
  const int sn = ref_load(rp, rlen, jb, v0, v1);
  return ref_unpack(v0, v1, sn);
}

// reference nibbles of a chunk for a record whose clipped CIGAR has more than three pieces: walks the CIGAR.
// Bits [blo, bhi) of the chunk are clipped bases cbase + b.  Insertions copy the read's own nibble (=> no mismatch).
__device__ __noinline__ uint64_t ref_nibbles_complex(const uint32_t *__restrict__ cg, int ncig, int64_t j0, int cbase, int blo, int bhi,
                                                     const uint8_t *__restrict__ rp, int64_t rlen, uint64_t S) {
  uint64_t R = 0;
  int ri = 0;
  int64_t j = j0;
  for (int i = 0; i < ncig; i++) {
    const uint32_t op = c_op(cg[i]);
    const int ln = c_len(cg[i]);
    if (op == OP_M || op == OP_EQ || op == OP_X || op == OP_I || op == OP_S) {
      int lo = ri - cbase, hi = ri + ln - cbase;
      lo = lo > blo ? lo : blo;
      hi = hi < bhi ? hi : bhi;
      if (lo < hi) {
        const uint64_t m = nib_fill(nib_range(lo, hi));
        if (op == OP_I || op == OP_S) R |= S & m;
        else R |= ref_nibbles(rp, rlen, j - ri + cbase) & m;
      }
      ri += ln;
      if (op != OP_I && op != OP_S) j += ln;
      if (ri - cbase >= bhi) break;
    } else if (op == OP_D || op == OP_N) {
      j += ln;
    }
  }
  return R;
}


// quality value -> LDS table row offset of this pass; the special values:
// qualities > 93 count into row n_q ("bad"), qualities 6..93 the host did not give a slot (sampling hint incomplete) into row
// n_q + 1 ("missing") of their covariate; the flush turns a non-zero cell of those rows into an error bit and the host reacts
struct QMap { uint8_t slot[96]; };         // 6..93 -> slot of this pass, 255 = other pass, 254 = unknown to the host


typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) unsigned long long lds_u64_t;
__device__ __forceinline__ uint32_t lds_address(const void *p) {
  return (uint32_t)reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void *)p);
}
__device__ __forceinline__ void lds_add_u32(uint32_t at, uint32_t v) {
  __hip_atomic_fetch_add(reinterpret_cast<lds_u32_t *>((uintptr_t)at), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_add_u64(uint32_t at, uint32_t lo, uint32_t hi) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  const u32x2 v = {lo, hi};
  __hip_atomic_fetch_add(reinterpret_cast<lds_u64_t *>((uintptr_t)at), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_WORKGROUP);
}
// the instruction, not whatever instcombine makes of the shift-and-mask around it
template <int OFF, int W>
__device__ __forceinline__ uint32_t bfe_u32(uint32_t x) {
  uint32_t r;
  asm("v_bfe_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(OFF), "n"(W));
  return r;
}
template <int SH>
__device__ __forceinline__ uint32_t lshl_add_u32(uint32_t a, uint32_t b) {  // (a << SH) + b
  uint32_t r;
  asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "n"(SH), "v"(b));
  return r;
}


// ---- the per-read record of the third count kernel (count3.hip), written by the prologue kernels next to (or instead of) BqDesc
//
// 32 bytes, everything in ORIGINAL read coordinates (the clipped copy the reference works on is bases [a, e) of the read), nothing
// that depends on the layout of the count tables (a quality-slot retry re-runs the count kernel on the same records):
//   ref      pointer to the packed-reference byte 8 bytes in front of the one that holds the reference base aligned with read base 0 along
//            the FIRST piece that has a reference (E0 >> 1, E0 = D_first - a); the reference base of read base k in piece j is E0 + delta_j + k
//   win      a | e << 16; e == a: the read contributes nothing
//   ctxw     lo | hi << 16: the bases whose context covariate is valid (bqsr.go:87-146 on the clipped copy)
//   t0       17 * (cf - a * ci): cycle-cell position of read base k, in sixteenths of a table word, is t0 + 17 ci k + (table origin)
//   fl       read-group covariate | RC_* << 8 | (bit j: piece j has no reference base - insertion) << 24
//   bpk      b1 | b2 << 10 | b3 << 20: first clipped base of pieces 1..3 (1023 = no such piece)
//   dpk      delta_1 | delta_2 << 8 | delta_3 << 16, signed bytes
// Reads the record cannot describe (more than four pieces, a piece further than -16 .. +14 bases from the first one, more than 1022
// bases, a window that is not inside the contig's allocation) carry RC_GENERAL: the kernel takes their data from BqDesc and walks.
struct __attribute__((aligned(16))) BqRec {
  uint32_t ref_lo, ref_hi, win, ctxw;
  int32_t t0;
  uint32_t fl, bpk, dpk;
};
static_assert(sizeof(BqRec) == 32, "BqRec is loaded as two 16-byte words");
enum : uint32_t { RC_REV = 1u << 8, RC_PAR = 1u << 9, RC_NEG = 1u << 10, RC_MULTI = 1u << 11, RC_GENERAL = 1u << 12,
                  RC_SKIPCOL = 1u << 13 };  // the read's known-site bits are in the skip column (else: the flags of its reference window)

// Where the records go (round 3): only reads that take part in the count get one, COMPACTED - the count kernel then spends no lane on a
// duplicate / unmapped / filtered read (one read in ten on the bench workload), and the reads whose blocks need the piece logic, the
// skip column or the descriptor ("other": class 2) sit apart from the reads that are one run of matches (class 1), so that no wave
// runs the long path for the sake of one lane.  Class 1 records fill `nseg` segments (a wave of the first prologue pass appends to
// segment wave % nseg with one atomic per tile: a single counter would serialise), class 2 records one region behind them.  The read's
// staging index travels in spare bits of the record.
// Round 5: the segments' first slots come from a device array (`seg_base`) instead of a fixed stride.  Without the covariate split they
// are a fixed stride apart as before; with it (ncs > 0) they are the prefix sums of an exact count of the reads each segment can receive
// (k_c3_seg_hist: the prologue's wave -> segment mapping applied to the RGID column), so the class-1 area is n records whatever the
// number of read groups - the fixed stride had to hold the worst case per segment, ncs * n records in all: 100 GB for 64 read groups and
// 50 M reads.  nseg = max(64, ncs): up to 256 segments, i.e. any number of read-group covariates a u8 holds.
constexpr int C3_NSEG = 64;       // segments without the covariate split; the least with it
constexpr int C3_MAXSEG = 256;
constexpr int C3_CSTRIDE = 64;  // words between two counters: every counter in a 256-byte line of its own (one line = one L2 channel would serialise them all)
struct RecOut {
  BqRec *recs;     // nullptr: descriptors are written instead (general count kernel)
  uint32_t *cnt;   // [s * C3_CSTRIDE], s < nseg: records in segment s; [nseg * C3_CSTRIDE]: records in the "other" region
  const uint32_t *seg_base;  // [nseg] first record slot of segment s
  uint64_t other_at;  // first record slot of the other region
  uint32_t nseg;   // 64, 128 or 256
  uint32_t ncs;    // 0: a segment holds records of every covariate.  > 0 (a power of two <= nseg): segment s holds records of covariate
                   // s % ncs only (workgroup b of the prologue appends to segment (b % (nseg / ncs)) * ncs + covariate): a workgroup of the count kernel
                   // then meets ONE covariate at a time and its private table needs that covariate's rows only
};
// k_bqsr_prologue_fast's workgroups take PF_TILES * 256 consecutive staged records, thread t of a workgroup the records t, t + 256, ...;
// k_c3_seg_hist walks the RGID column in the same workgroups to size the covariate-split segments exactly
constexpr int PF_TILES = 16;
__device__ __forceinline__ void rec_pack_idx(BqRec &r, uint32_t idx) {
  r.ref_hi = (r.ref_hi & 0xFFFFu) | (idx << 16);
  r.fl = (r.fl & ~(0x3FFu << 14)) | (((idx >> 16) & 0x3FFu) << 14);
  r.dpk = (r.dpk & 0x00FFFFFFu) | ((idx >> 26) << 24);
}
__device__ __forceinline__ uint32_t rec_idx(uint32_t ref_hi, uint32_t fl, uint32_t dpk) { return (ref_hi >> 16) | (((fl >> 14) & 0x3FFu) << 16) | ((dpk >> 24) << 26); }
__device__ __forceinline__ void rec_store(BqRec *recs, uint64_t at, const BqRec &rc) {
  reinterpret_cast<uint4 *>(recs)[2 * at] = make_uint4(rc.ref_lo, rc.ref_hi, rc.win, rc.ctxw);
  reinterpret_cast<uint4 *>(recs)[2 * at + 1] = make_uint4((uint32_t)rc.t0, rc.fl, rc.bpk, rc.dpk);
}
// class of a finished record: 0 = no part in the count, 1 = one run of matches with its known-site bits in the reference window, 2 = other
__device__ __forceinline__ int rec_class(const BqRec &rc) {
  if ((rc.win >> 16) == (rc.win & 0xFFFFu)) return 0;
  return (rc.fl & (RC_MULTI | RC_GENERAL | RC_SKIPCOL)) ? 2 : 1;
}
// the lanes of a wave with `flag` set take consecutive places behind *counter (one atomic per wave); returns the lane's place
__device__ __forceinline__ uint32_t wave_append(bool flag, uint32_t *counter) {
  const unsigned long long mask = __ballot(flag);
  if (!mask) return 0;
  const int lane = threadIdx.x & 63, leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = __shfl(base, leader, 64);
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// pieces of a clipped CIGAR (see BqDesc), up to four; np = -1: more
struct Pieces4 {
  int64_t v0, v1, v2, v3;  // reference index minus clipped read index along the piece
  int s1, s2, s3;          // first clipped base of pieces 1..3
  uint32_t noref;          // bit j: piece j is an insertion
  int np;
};
__device__ inline void pieces4(const uint32_t *cig, int ncig, int32_t pos, Pieces4 &P) {
  P.v0 = P.v1 = P.v2 = P.v3 = 0; P.s1 = P.s2 = P.s3 = 0; P.noref = 0; P.np = 0;
  int c = 0;
  int64_t delta = (int64_t)pos - 1, lastv = 0;
  bool lastn = false;
  for (int i = 0; i < ncig && P.np >= 0; i++) {
    const uint32_t op = c_op(cig[i]);
    const int ln = c_len(cig[i]);
    bool push = false, pn = false;
    if (op == OP_M || op == OP_EQ || op == OP_X) {
      push = ln > 0 && (P.np == 0 || lastn || lastv != delta);
      pn = false;
    } else if (op == OP_I || op == OP_S) {
      push = ln > 0 && (P.np == 0 || !lastn);
      pn = true;
    }
    if (push) {
      if (P.np == 4) { P.np = -1; break; }
      const int64_t v = pn ? 0 : delta;
      if (P.np == 0) P.v0 = v;
      else if (P.np == 1) { P.v1 = v; P.s1 = c; }
      else if (P.np == 2) { P.v2 = v; P.s2 = c; }
      else { P.v3 = v; P.s3 = c; }
      P.noref |= (pn ? 1u : 0u) << P.np;
      lastv = v; lastn = pn;
      P.np++;
    }
    if (op == OP_M || op == OP_EQ || op == OP_X) c += ln;
    else if (op == OP_I || op == OP_S) { c += ln; delta -= ln; }
    else if (op == OP_D || op == OP_N) delta += ln;
  }
}

// the record of an eligible read (a, len, left, right, cov, flags as in BqDesc); rp / rlen: packed contig and its length in bases
__device__ inline BqRec make_rec(int a, int len, int left, int right, uint32_t cov, bool rev, bool last, const Pieces4 &P, bool walk,
                                 const uint8_t *rp, int64_t rlen, int64_t lorig) {
  BqRec r;
  const int e = a + len;
  r.win = (uint32_t)a | ((uint32_t)e << 16);
  int cl = left + (rev ? 0 : 1), cr1 = right - (rev ? 1 : 0) + 1;
  cl = cl < 0 ? 0 : cl;
  cr1 = cr1 > len ? len : cr1;
  cr1 = cr1 < cl ? cl : cr1;
  r.ctxw = (uint32_t)(a + cl) | ((uint32_t)(a + cr1) << 16);
  const int rof = last ? -1 : 1;
  const int cf = rof + (rev ? (len - 1) * rof : 0), ci = rev ? -rof : rof;
  r.t0 = 17 * (cf - a * ci);
  uint32_t fl = (cov & 0xFFu) | (rev ? RC_REV : 0u) | (ci < 0 ? RC_NEG : 0u);
  bool general = walk || P.np < 1 || len > 1022;
  // first piece with a reference
  const uint32_t nr = P.noref;
  const int f = !(nr & 1u) ? 0 : (!(nr & 2u) ? 1 : (!(nr & 4u) ? 2 : 3));
  const int64_t vf = f == 0 ? P.v0 : (f == 1 ? P.v1 : (f == 2 ? P.v2 : P.v3));
  const bool any_ref = !general && f < P.np;
  int64_t E0 = any_ref ? vf - a : 16;
  int d1 = 0, d2 = 0, d3 = 0;
  if (!general && any_ref) {
    if (P.np > 1 && !(nr & 2u)) { const int64_t d = P.v1 - vf; general |= d < -16 || d > 14; d1 = (int)d; }
    if (P.np > 2 && !(nr & 4u)) { const int64_t d = P.v2 - vf; general |= d < -16 || d > 14; d2 = (int)d; }
    if (P.np > 3 && !(nr & 8u)) { const int64_t d = P.v3 - vf; general |= d < -16 || d > 14; d3 = (int)d; }
    general |= E0 < 16 || E0 + lorig > rlen + 32;  // every 24-byte window of the read lies inside the contig's allocation
  }
  general |= !any_ref;  // nothing but insertions: rare enough for the general path
  if (general) {
    fl |= RC_GENERAL;
    r.bpk = 0x3FFFFFFFu; r.dpk = 0;
  } else {
    fl |= (nr & 0xFu) << 24;
    if (P.np > 1 || (nr & 1u)) fl |= RC_MULTI;
    r.bpk = (uint32_t)(P.np > 1 ? P.s1 : 1023) | ((uint32_t)(P.np > 2 ? P.s2 : 1023) << 10) | ((uint32_t)(P.np > 3 ? P.s3 : 1023) << 20);
    r.dpk = ((uint32_t)d1 & 0xFFu) | (((uint32_t)d2 & 0xFFu) << 8) | (((uint32_t)d3 & 0xFFu) << 16);
    fl |= (E0 & 1) ? RC_PAR : 0u;
  }
  // the kernel loads 24 bytes from ref + (k0 >> 1): the window of block k0 starts 16 bases in front of its first reference base
  // (general reads: the contig's first bytes, never used)
  const uint64_t pq = reinterpret_cast<uint64_t>(general ? rp : rp + (E0 >> 1) - 8);
  r.ref_lo = (uint32_t)pq;
  r.ref_hi = (uint32_t)(pq >> 32);
  r.fl = fl;
  return r;
}

// ---- the count kernel for read sets of one length (count3.hip)
struct Count3Args {
  // The records of a launch lie in `nseg` segments of one array: the class-1 segments (RecOut), or the other region - one segment, or,
  // with the covariate split, one per covariate (the region sorted by covariate).  The workgroups share the launch's trips evenly,
  // whatever the segments' sizes (a workgroup takes a contiguous range of the concatenated segments' trips and meets one segment after
  // the other): read groups of very different sizes, or fewer covariates than segments, leave nobody idle.
  const uint32_t *seg_cnt;   // records in segment s: seg_cnt[s * cnt_stride]
  uint32_t cnt_stride;
  const uint32_t *seg_base;  // first record of segment s in srecs
  const uint4 *srecs;        // the record array of this launch (32-byte records)
  uint32_t nseg;
  int other;            // 0: the class-1 segments, 1: the other region
  int ncs;              // RecOut.ncs: > 0 = segment s holds covariate s % ncs only
  uint32_t len;  // every staged read has this many bases (SEQ and QUAL)
  const uint8_t *qual, *seq4;  // seq4: first SEQ byte of read 0
  const uint8_t *skipbits;
  const uint4 *desc;
  const uint32_t *cigar, *cig_scratch;
  uint8_t *const *ref_seq;
  const int64_t *ref_seq_len;
  int n_cov, n_q, lmax, max_cycle;
  int rsw, rlog;  // words per row; log2 of the context replication R
  unsigned long long *cycle_tbl, *ctx_tbl;
  uint32_t *err;
};
int count3_plan(int n_cov, int n_q, int lmax, int *rsw_out, int *rlog_out, size_t *dyn_out, int force_rlog = -1);
int count3_launch(elp_ctx *c, const Count3Args &A, const QMap &qm, size_t dyn);
// ---- ApplyBQSR for read sets of one length (apply3.hip)
int apply3_bytes(int n_cov, int n_qi, int lmax, size_t *dyn_out);
int apply3_launch(elp_ctx *c, int max_cycle, const uint8_t *d_lut, const uint8_t *d_cov_present, const uint16_t *t1, const uint8_t *t2, const uint32_t *n_dict_dev,
                  int n_qi, int lmax, size_t dyn, bool split /* records sorted by covariate, per-covariate row dictionaries */);

}  // namespace elp
