// bqsr.hip — base quality score recalibration on the HBM column store: covariate-table gather and LUT apply.
//
// Reference: BaseRecalibrator.Recalibrate (filters/bqsr.go:467-551), BaseRecalibratorTables.ApplyBQSR (:936-1005); device
// helpers (clipping, covariates) and their citations are in bqsr_dev.hpp.
//
// Gather = three kernels
//   bqsr_prologue  one thread per record: eligibility (recalibrateAln), adaptor + soft-clip hard clipping on a working copy,
//                  known-site skip mask (calculateSkipSlice) written into a 1-bit-per-base column, low-quality-tail bounds;
//                  leaves a 20-byte descriptor per record and the rewritten CIGAR in scratch.
//   bqsr_count     flat stream over QUAL/SEQ (flat.hpp): one lane per 16 bases; SNP event against the reference, cycle and
//                  context covariates, then ONE packed LDS atomic per base into a workgroup-private cycle table
//                  (observations in the low, mismatches in the high 16 bits) and one into the private context table.
//                  Private tables are flushed into per-workgroup partial tables in HBM (plain stores, no atomics).
//   bqsr_reduce    sums the partial tables into the dense int64 tables of the C ABI; the QualityScores table is the sum of the
//                  Cycles table over cycles (every counted base updates both with the same (read group, quality)).
// Apply = a small per-record prologue (low-quality bounds) + a flat per-base LUT kernel that rewrites QUAL in place.
#include "bqsr_dev.hpp"
#include "flat.hpp"

namespace elp {

constexpr int MAX_DESC_READ = 65535;  // u16 fields of the record descriptor

struct BqCols {
  uint64_t n;
  const int32_t *refid, *pos, *next_refid, *pnext, *tlen;
  const uint16_t *flag, *rgid;
  const uint8_t *mapq, *has_sr;
  const uint32_t *l_seq;
  const uint64_t *cigar_off, *seq_off, *qual_off;
  const uint32_t *cigar;
  const uint8_t *seq4;
  const uint8_t *qual;
  const int32_t *ref_len;
  const uint16_t *rg_cov;
  int32_t n_ref;
  uint8_t *const *ref_seq;
  const int64_t *ref_seq_len;
  int32_t *const *sites;
  const int64_t *n_sites;
};

// recalibrateAln, bqsr.go:225-244 (+ utils.go:121-139)
__device__ inline bool recalibrate_aln(const BqCols &m, uint64_t i) {
  if (m.has_sr[i]) return false;
  const uint8_t mq = m.mapq[i];
  if (!(mq > 0 && mq < 255)) return false;
  const uint16_t f = m.flag[i];
  if (f & (F_SECONDARY | F_DUPLICATE | F_QCFAILED)) return false;
  const int32_t r = m.refid[i], p = m.pos[i];
  if ((f & F_UNMAPPED) || r < 0 || p == 0) return false;
  if (!(p > 0)) return false;
  const uint32_t ls = m.l_seq[i];
  if (ls == 0) return false;
  if ((uint64_t)ls != m.qual_off[i + 1] - m.qual_off[i]) return false;
  if (m.rgid[i] == ELP_NIL16) return false;
  if (!(r < m.n_ref && p <= m.ref_len[r])) return false;
  int32_t rl = 0, refl = 0;
  for (uint64_t k = m.cigar_off[i]; k < m.cigar_off[i + 1]; k++) {
    const uint32_t c = m.cigar[k];
    if (c_op(c) == OP_N) return false;
    if (op_consumes_read(c_op(c))) rl += c_len(c);
    if (op_consumes_ref(c_op(c))) refl += c_len(c);
  }
  return refl >= 0 && (int32_t)ls == rl;
}

// per-record descriptor produced by the prologue
struct BqDesc {
  int32_t pos;     // POS of the clipped working copy
  uint32_t cig;    // index of its CIGAR: into the scratch CIGAR pool (fl & 8) or into the staged cigar column
  uint16_t ncig;
  uint16_t a;      // first surviving base (original read coordinates)
  uint16_t len;    // surviving bases; 0 = record contributes nothing
  uint16_t left;   // low-quality-tail bounds inside the surviving window (left > right: everything masked)
  uint16_t right;
  uint8_t cov;     // read-group covariate id
  uint8_t fl;      // 1 eligible, 2 reversed, 4 last segment, 8 CIGAR in scratch
};

__global__ __launch_bounds__(256) void k_bqsr_prologue(BqCols m, uint32_t *__restrict__ cig_scratch, BqDesc *__restrict__ desc,
                                                       uint32_t *skipbits, uint32_t *err) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.n) return;
  BqDesc d;
  d.pos = 0; d.cig = 0; d.ncig = 0; d.a = 0; d.len = 0; d.left = 0; d.right = 0; d.cov = 0; d.fl = 0;
  if (!recalibrate_aln(m, i)) { desc[i] = d; return; }
  if (m.l_seq[i] > (uint32_t)MAX_DESC_READ) { atomicOr(&err[0], 2u); desc[i] = d; return; }
  RAln a;
  a.pos = m.pos[i]; a.pnext = m.pnext[i]; a.tlen = m.tlen[i]; a.refid = m.refid[i]; a.next_refid = m.next_refid[i];
  a.flag = m.flag[i];
  a.cig = m.cigar + m.cigar_off[i];
  a.ncig = (int)(m.cigar_off[i + 1] - m.cigar_off[i]);
  a.off = 0; a.len = (int)m.l_seq[i];
  uint32_t *sc = cig_scratch + 2 * (m.cigar_off[i] + 4 * i);
  a.buf[0] = sc; a.buf[1] = sc + (a.ncig + 4);
  a.cur = -1;
  if (!hard_clip_adaptor(a)) { atomicOr(&err[0], 4u); desc[i] = d; return; }
  if (a.len == 0) { desc[i] = d; return; }
  hard_clip_soft_clipped(a);
  if (a.len == 0) { desc[i] = d; return; }

  // calculateSkipSlice, bqsr.go:389-414: bits live at (qual_off[i] + original base index)
  {
    const int ss = soft_start(a), se = soft_end(a);
    const int32_t *sv = m.sites[a.refid];
    const int64_t ns = m.n_sites[a.refid];
    int64_t lo = 0, hi = ns;
    while (lo < hi) { const int64_t md = lo + (hi - lo) / 2; if (!(sv[2 * md + 1] >= ss)) lo = md + 1; else hi = md; }
    const int64_t first = lo;
    lo = 0; hi = ns;
    while (lo < hi) { const int64_t md = lo + (hi - lo) / 2; if (!(sv[2 * md] > se)) lo = md + 1; else hi = md; }
    const int64_t last = lo;
    const uint64_t bit0 = m.qual_off[i] + (uint64_t)a.off;
    for (int64_t s = first; s < last; s++) {
      bool ok;
      int fs = get_read_coord(a.cig, a.ncig, ss, sv[2 * s], false, &ok);
      if (!ok || fs < 0) fs = 0;
      int fe = get_read_coord(a.cig, a.ncig, ss, sv[2 * s + 1], false, &ok);
      if (!ok || fe > a.len - 1) fe = a.len - 1;
      for (int k = fs; k <= fe;) {  // set bits word by word
        const uint64_t b = bit0 + (uint64_t)k;
        const int in_word = (int)(b & 31);
        int cnt = 32 - in_word;
        if (cnt > fe - k + 1) cnt = fe - k + 1;
        const uint32_t mask = (cnt == 32 ? 0xFFFFFFFFu : ((1u << cnt) - 1u)) << in_word;
        atomicOr(&skipbits[b >> 5], mask);
        k += cnt;
      }
    }
  }
  ReadView v{m.seq4 + m.seq_off[i], m.qual + m.qual_off[i], a.off, a.len, (bool)(a.flag & F_REVERSED), 0, -1};
  low_quality_bounds(v);
  d.pos = a.pos;
  if (a.cur < 0) { d.cig = (uint32_t)m.cigar_off[i]; }
  else { d.cig = (uint32_t)(a.buf[a.cur] - cig_scratch); d.fl |= 8; }
  d.ncig = (uint16_t)a.ncig;
  d.a = (uint16_t)a.off; d.len = (uint16_t)a.len;
  d.left = (uint16_t)v.left; d.right = (uint16_t)(v.right < 0 ? 0xFFFF : v.right);
  d.cov = (uint8_t)m.rg_cov[m.rgid[i]];
  d.fl |= 1 | ((a.flag & F_REVERSED) ? 2 : 0) | ((a.flag & F_LAST) ? 4 : 0);
  desc[i] = d;
}

struct QMap { uint8_t slot[96]; };  // quality value -> table slot of this pass, 255 = not in this pass

struct CountArgs {
  uint64_t n, qual_bytes;
  const uint64_t *qual_off, *seq_off;
  const uint8_t *qual, *seq4;
  const int32_t *refid;
  const BqDesc *desc;
  const uint32_t *cigar, *cig_scratch;
  const uint16_t *skip16;   // the skip-bit column viewed as 16-bit words (one per 16-byte QUAL chunk)
  uint8_t *const *ref_seq;
  const int64_t *ref_seq_len;
  int n_cov, n_q, lmax, cs, s16, max_cycle;  // cs = 16 * s16 padded cycle row, s16 = ceil((2*lmax+1)/16)
  uint32_t *partial;        // [grid][cyc cells * 2 + ctx cells * 2] u32
  uint32_t *err;
  const uint32_t *tile_first;
};

// 128-bit window over the packed bases of one record: nibble(k) for k in [kb, kb+32)
struct SeqWin {
  uint64_t v0, v1;
  int kb;
  __device__ __forceinline__ uint32_t nib(int k) const {
    const int t = k - kb;           // 0..31
    const int byte = t >> 1;
    const uint64_t w = byte < 8 ? v0 : v1;
    const uint32_t b = (uint32_t)(w >> (8 * (byte & 7))) & 0xFF;
    return (t & 1) ? (b & 0xF) : (b >> 4);
  }
};
__device__ __forceinline__ SeqWin load_seq_window(const uint8_t *__restrict__ seq, int k_first) {
  SeqWin w;
  w.kb = (k_first > 0 ? k_first : 0) & ~1;
  const uint8_t *p = seq + (w.kb >> 1);
  __builtin_memcpy(&w.v0, p, 8);
  __builtin_memcpy(&w.v1, p + 8, 8);
  return w;
}

__global__ __launch_bounds__(FL_THREADS) void k_bqsr_count(CountArgs A, QMap qm) {
  __shared__ FlatLds L;
  __shared__ BqDesc s_desc[FL_RMAX];
  __shared__ uint64_t s_seq[FL_RMAX];
  __shared__ int32_t s_ref[FL_RMAX];
  extern __shared__ __attribute__((aligned(16))) uint32_t tbl[];  // cyc[n_cov*n_q*cs] | ctx_obs[n_cov*n_q*16] | ctx_mism[n_cov*n_q*16]
  const int ncq = A.n_cov * A.n_q;
  const int n_cyc = ncq * A.cs, n_ctx = ncq * 16;
  uint32_t *t_cyc = tbl, *t_cobs = tbl + n_cyc, *t_cmis = t_cobs + n_ctx;
  const int n_all = n_cyc + 2 * n_ctx;
  for (int k = threadIdx.x; k < n_all; k += FL_THREADS) tbl[k] = 0;
  uint32_t *part = A.partial + (size_t)blockIdx.x * (size_t)(2 * n_cyc + 2 * n_ctx);
  __syncthreads();
  uint32_t reads_since_flush = 0;
  uint32_t my_err = 0;
  auto flush = [&]() {
    __syncthreads();
    for (int k = threadIdx.x; k < n_cyc; k += FL_THREADS) {
      const uint32_t v = t_cyc[k];
      if (v) { part[2 * k] += v & 0xFFFF; part[2 * k + 1] += v >> 16; t_cyc[k] = 0; }
    }
    for (int k = threadIdx.x; k < n_ctx; k += FL_THREADS) {
      const uint32_t o = t_cobs[k], e = t_cmis[k];
      if (o) { part[2 * n_cyc + 2 * k] += o; t_cobs[k] = 0; }
      if (e) { part[2 * n_cyc + 2 * k + 1] += e; t_cmis[k] = 0; }
    }
    __syncthreads();
  };
  const uint64_t ntiles = (A.qual_bytes + FL_TILE - 1) / FL_TILE;
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t tb = t * FL_TILE, te = (tb + FL_TILE < A.qual_bytes) ? tb + FL_TILE : A.qual_bytes;
    reads_since_flush += flat_tile(A.qual_off, A.n, A.qual, tb, te, A.tile_first[t], A.tile_first[t + 1], L,
              [&](uint32_t g0, uint32_t ng) __attribute__((always_inline)) {  // stage the group's descriptors in LDS (coalesced)
                for (uint32_t k = threadIdx.x; k < ng; k += FL_THREADS) {
                  s_desc[k] = A.desc[g0 + k];
                  s_seq[k] = A.seq_off[g0 + k];
                  s_ref[k] = A.refid[g0 + k];
                }
              },
              [&](uint32_t rl, int k0, int k1, Chunk &ch, int o, uint64_t p) __attribute__((always_inline)) {
                const BqDesc d = s_desc[rl];
                if (!(d.fl & 1)) return;
                int c0 = k0 - (int)d.a, c1 = k1 - (int)d.a;  // clipped coordinates
                if (c0 < 0) c0 = 0;
                if (c1 > (int)d.len) c1 = (int)d.len;
                if (c0 >= c1) return;
                const uint8_t *seq = A.seq4 + s_seq[rl];
                const SeqWin sw = load_seq_window(seq, (int)d.a + c0 - 1);
                const uint32_t skipw = A.skip16[p >> 4];
                const int rf = s_ref[rl];
                const uint8_t *ref = A.ref_seq[rf];
                const int64_t rlen = A.ref_seq_len[rf];
                const uint32_t *cg = ((d.fl & 8) ? A.cig_scratch : A.cigar) + d.cig;
                const bool rev = d.fl & 2;
                const int rof = (d.fl & 4) ? -1 : 1;
                const int cf = rof + (rev ? ((int)d.len - 1) * rof : 0), ci = (rev ? -1 : 1) * rof;  // bqsr.go:376-383
                const int left = d.left, right = d.right == 0xFFFF ? -1 : (int)d.right;
                // walk the CIGAR to clipped base c0 (computeSnpEvents, bqsr.go:254-285)
                int opi = 0, rem = 0;
                bool ism = false;
                int64_t j = (int64_t)d.pos - 1;
                {
                  int ri = 0;
                  for (; opi < (int)d.ncig; opi++) {
                    const uint32_t el = cg[opi];
                    const uint32_t op = c_op(el);
                    const int ln = c_len(el);
                    if (op == OP_M || op == OP_EQ || op == OP_X) {
                      if (c0 < ri + ln) { ism = true; rem = ri + ln - c0; j += c0 - ri; break; }
                      ri += ln; j += ln;
                    } else if (op == OP_D || op == OP_N) {
                      j += ln;
                    } else if (op == OP_I || op == OP_S) {
                      if (c0 < ri + ln) { ism = false; rem = ri + ln - c0; break; }
                      ri += ln;
                    }
                  }
                }
                const int row0 = (int)d.cov * A.n_q;
                for (int c = c0; c < c1; c++) {
                  // advance to the op that holds base c
                  while (rem == 0 && opi < (int)d.ncig) {
                    opi++;
                    if (opi >= (int)d.ncig) break;
                    const uint32_t el = cg[opi];
                    const uint32_t op = c_op(el);
                    if (op == OP_M || op == OP_EQ || op == OP_X) { ism = true; rem = c_len(el); }
                    else if (op == OP_I || op == OP_S) { ism = false; rem = c_len(el); }
                    else if (op == OP_D || op == OP_N) { j += c_len(el); }
                  }
                  const bool in_cigar = rem > 0;
                  const int64_t jj = j;
                  if (in_cigar) { rem--; if (ism) j++; }
                  const int kk = (int)d.a + c;                 // original base index
                  const int bi = o + (kk - k0);                // byte inside the lane's chunk
                  if ((skipw >> bi) & 1u) continue;
                  const uint32_t nb = sw.nib(kk);
                  const int bidx = base_index_of_nibble(nb);
                  if (bidx < 0) continue;
                  const uint32_t q = ch.get(bi);
                  if (q < 6) continue;
                  if (q >= ELP_NQUAL) { my_err |= 8u; continue; }
                  const uint32_t slot = qm.slot[q];
                  if (slot == 255) continue;
                  uint32_t e = 0;
                  if (in_cigar && ism) {
                    const int rb = (jj >= 0 && jj < rlen) ? base_code_of_ref(ref[jj]) : 0;
                    e = (bidx + 1) != rb;
                  }
                  const int cyc = cf + c * ci;
                  if (cyc > A.max_cycle || cyc < -A.max_cycle) { my_err |= 16u; continue; }  // checkCycleCovariate :364-369
                  const int cidx = cyc + A.lmax;
                  const int row = row0 + (int)slot;
                  atomicAdd(&t_cyc[row * A.cs + (cidx & 15) * A.s16 + (cidx >> 4)], 1u | (e << 16));
                  // context covariate (bqsr.go:87-146) of clipped base c
                  int cx = -1;
                  if (c >= left && c <= right) {
                    const int cn = rev ? c + 1 : c - 1;
                    if (cn >= left && cn <= right && cn >= 0 && cn < (int)d.len) {
                      const int pn = base_index_of_nibble(sw.nib((int)d.a + cn));
                      if (pn >= 0) cx = rev ? (((3 - pn)) | ((3 - bidx) << 2)) : (pn | (bidx << 2));
                    }
                  }
                  if (cx >= 0) {
                    atomicAdd(&t_cobs[row * 16 + cx], 1u);
                    if (e) atomicAdd(&t_cmis[row * 16 + cx], 1u);
                  }
                }
              },
              [&](uint64_t, int, int, Chunk &) __attribute__((always_inline)) {}, [&](uint32_t, uint32_t) __attribute__((always_inline)) {});
    if (reads_since_flush > 50000u) { flush(); reads_since_flush = 0; }
  }
  flush();
  if (__any(my_err != 0)) {
    for (int d = 32; d >= 1; d >>= 1) my_err |= __shfl_xor(my_err, d, 64);
    if ((threadIdx.x & 63) == 0) atomicOr(&A.err[0], my_err);
  }
}

// sums the per-workgroup partials into the dense tables; one thread per logical (cov, slot, cycle) / (cov, slot, ctx) cell
__global__ __launch_bounds__(256) void k_bqsr_reduce(const uint32_t *__restrict__ partial, int nblk, int n_cov, int n_q, int lmax, int cs, int s16,
                                                     int max_cycle, QMap slot_to_q, unsigned long long *cycle_tbl, unsigned long long *ctx_tbl) {
  const int ncq = n_cov * n_q;
  const int ncyc_l = 2 * lmax + 1;
  const int n_cyc = ncq * cs, n_ctx = ncq * 16;
  const size_t stride = (size_t)(2 * n_cyc + 2 * n_ctx);
  const int total = ncq * ncyc_l + ncq * 16;
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= total) return;
  const int ncyc_g = 2 * max_cycle + 1;
  if (id < ncq * ncyc_l) {
    const int row = id / ncyc_l, cidx = id % ncyc_l;
    const size_t k = (size_t)row * cs + (size_t)(cidx & 15) * s16 + (cidx >> 4);
    unsigned long long o = 0, e = 0;
    for (int b = 0; b < nblk; b++) { o += partial[b * stride + 2 * k]; e += partial[b * stride + 2 * k + 1]; }
    if (o | e) {
      const int cov = row / n_q, q = slot_to_q.slot[row % n_q];
      const int cyc = cidx - lmax;
      const size_t g = (((size_t)cov * ELP_NQUAL + q) * ncyc_g + (size_t)(cyc + max_cycle)) * 2;
      cycle_tbl[g] += o; cycle_tbl[g + 1] += e;
    }
  } else {
    const int x = id - ncq * ncyc_l;
    const int row = x / 16, cx = x % 16;
    unsigned long long o = 0, e = 0;
    for (int b = 0; b < nblk; b++) { o += partial[b * stride + 2 * n_cyc + 2 * x]; e += partial[b * stride + 2 * n_cyc + 2 * x + 1]; }
    if (o | e) {
      const int cov = row / n_q, q = slot_to_q.slot[row % n_q];
      // cx = prev | cur << 2 is exactly (key >> 4) & 15 of keyFromContext (bqsr.go:64-76)
      const size_t g = (((size_t)cov * ELP_NQUAL + q) * ELP_NCTX + (size_t)cx) * 2;
      ctx_tbl[g] += o; ctx_tbl[g + 1] += e;
    }
  }
}

// QualityScores[cov][q] = sum over cycles of Cycles[cov][q][*]
__global__ __launch_bounds__(256) void k_bqsr_qual_from_cycle(int n_rows, int ncyc_g, const unsigned long long *__restrict__ cycle_tbl,
                                                              unsigned long long *__restrict__ qual_tbl) {
  const int row = blockIdx.x;  // one workgroup per (cov, q)
  if (row >= n_rows) return;
  __shared__ unsigned long long so[256], se[256];
  unsigned long long o = 0, e = 0;
  for (int c = threadIdx.x; c < ncyc_g; c += 256) { o += cycle_tbl[((size_t)row * ncyc_g + c) * 2]; e += cycle_tbl[((size_t)row * ncyc_g + c) * 2 + 1]; }
  so[threadIdx.x] = o; se[threadIdx.x] = e;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if ((int)threadIdx.x < d) { so[threadIdx.x] += so[threadIdx.x + d]; se[threadIdx.x] += se[threadIdx.x + d]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { qual_tbl[2 * row] = so[0]; qual_tbl[2 * row + 1] = se[0]; }
}

// ------------------------------------------------------------------ apply
struct ApDesc { uint16_t left, right; uint8_t cov; uint8_t fl; };  // fl: 1 recalibrate, 2 reversed, 4 last

__global__ __launch_bounds__(256) void k_apply_prologue(uint64_t n, const uint16_t *__restrict__ flag, const uint16_t *__restrict__ rgid,
                                                        const uint16_t *__restrict__ rg_cov, const uint32_t *__restrict__ l_seq,
                                                        const uint64_t *__restrict__ qual_off, const uint8_t *__restrict__ qual,
                                                        const uint8_t *__restrict__ cov_present, ApDesc *__restrict__ desc, uint32_t *err) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ApDesc d{0, 0, 0, 0};
  const uint16_t rg = rgid[i];
  if (rg == ELP_NIL16) { atomicOr(&err[0], 32u); desc[i] = d; return; }  // readGroupCovariate panics, bqsr.go:38
  const uint32_t cov = rg_cov[rg];
  if (!cov_present[cov]) { desc[i] = d; return; }  // :953-955
  const int len = (int)l_seq[i];
  if ((uint64_t)len != qual_off[i + 1] - qual_off[i]) { atomicOr(&err[0], 64u); desc[i] = d; return; }
  if (len > MAX_DESC_READ) { atomicOr(&err[0], 2u); desc[i] = d; return; }
  const uint16_t f = flag[i];
  ReadView v{nullptr, qual + qual_off[i], 0, len, (bool)(f & F_REVERSED), 0, -1};
  low_quality_bounds(v);
  d.left = (uint16_t)v.left; d.right = (uint16_t)(v.right < 0 ? 0xFFFF : v.right);
  d.cov = (uint8_t)cov;
  d.fl = 1 | ((f & F_REVERSED) ? 2 : 0) | ((f & F_LAST) ? 4 : 0);
  desc[i] = d;
}

struct ApplyArgs {
  uint64_t n, qual_bytes;
  const uint64_t *qual_off, *seq_off;
  uint8_t *qual;
  const uint8_t *seq4;
  const uint32_t *l_seq;
  const ApDesc *desc;
  const uint32_t *tile_first;
  const uint8_t *lut;  // [n_cov][94][2*max_cycle+1][17]
  int max_cycle;
  uint32_t *err;
};

__global__ __launch_bounds__(FL_THREADS) void k_bqsr_apply_flat(ApplyArgs A) {
  __shared__ FlatLds L;
  __shared__ ApDesc s_desc[FL_RMAX];
  __shared__ uint64_t s_seq[FL_RMAX];
  __shared__ uint32_t s_len[FL_RMAX];
  uint32_t my_err = 0;
  const int ncyc = 2 * A.max_cycle + 1;
  const uint64_t ntiles = (A.qual_bytes + FL_TILE - 1) / FL_TILE;
  for (uint64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const uint64_t tb = t * FL_TILE, te = (tb + FL_TILE < A.qual_bytes) ? tb + FL_TILE : A.qual_bytes;
    flat_tile(A.qual_off, A.n, A.qual, tb, te, A.tile_first[t], A.tile_first[t + 1], L,
              [&](uint32_t g0, uint32_t ng) __attribute__((always_inline)) {
                for (uint32_t k = threadIdx.x; k < ng; k += FL_THREADS) {
                  s_desc[k] = A.desc[g0 + k];
                  s_seq[k] = A.seq_off[g0 + k];
                  s_len[k] = A.l_seq[g0 + k];
                }
              },
              [&](uint32_t rl, int k0, int k1, Chunk &ch, int o, uint64_t) __attribute__((always_inline)) {
                const ApDesc d = s_desc[rl];
                if (!(d.fl & 1)) return;
                const int len = (int)s_len[rl];
                const SeqWin sw = load_seq_window(A.seq4 + s_seq[rl], k0 - 1);
                const bool rev = d.fl & 2;
                const int rof = (d.fl & 4) ? -1 : 1;
                const int cf = rof + (rev ? (len - 1) * rof : 0), ci = (rev ? -1 : 1) * rof;
                const int left = d.left, right = d.right == 0xFFFF ? -1 : (int)d.right;
                const uint8_t *lc = A.lut + (size_t)d.cov * ELP_NQUAL * ncyc * 17;
                for (int k = k0; k < k1; k++) {
                  const int bi = o + (k - k0);
                  const uint32_t q = ch.get(bi);
                  if (q < 6) continue;
                  if (q >= ELP_NQUAL) { my_err |= 8u; continue; }
                  const int cyc = cf + k * ci;
                  if (cyc > A.max_cycle || cyc < -A.max_cycle) { my_err |= 16u; continue; }
                  int cx = 16;
                  if (k >= left && k <= right) {
                    const int kn = rev ? k + 1 : k - 1;
                    if (kn >= left && kn <= right && kn >= 0 && kn < len) {
                      const int pn = base_index_of_nibble(sw.nib(kn)), cu = base_index_of_nibble(sw.nib(k));
                      if (pn >= 0 && cu >= 0) cx = rev ? ((3 - pn) | ((3 - cu) << 2)) : (pn | (cu << 2));
                    }
                  }
                  ch.set(bi, lc[((size_t)q * ncyc + (size_t)(cyc + A.max_cycle)) * 17 + cx]);
                }
              },
              [&](uint64_t p, int lo, int hi, Chunk &ch) __attribute__((always_inline)) {
                if (lo == 0 && hi == FL_CHUNK) {
                  uint4 v;
                  v.x = (uint32_t)ch.w0; v.y = (uint32_t)(ch.w0 >> 32); v.z = (uint32_t)ch.w1; v.w = (uint32_t)(ch.w1 >> 32);
                  *reinterpret_cast<uint4 *>(A.qual + p) = v;
                } else {
                  for (int b = lo; b < hi; b++) A.qual[p + b] = (uint8_t)ch.get(b);  // partial chunk: another lane may own the rest
                }
              },
              [&](uint32_t, uint32_t) __attribute__((always_inline)) {});
  }
  if (__any(my_err != 0)) {
    for (int d = 32; d >= 1; d >>= 1) my_err |= __shfl_xor(my_err, d, 64);
    if ((threadIdx.x & 63) == 0) atomicOr(&A.err[0], my_err);
  }
}

static int bqsr_error(elp_ctx *c, uint32_t e) {
  ELP_HIP(c, hipMemsetAsync(c->err_flag.p, 0, 4, c->stream));
  if (e & 2u) return set_error(c, ELP_ERR_UNSUPPORTED, "BQSR: read longer than %d bases", MAX_DESC_READ);
  if (e & 4u) return set_error(c, ELP_ERR_DATA, "reference coordinate matches a non-existing base in read (reference: log.Panicf, filters/utils.go:253,262)");
  if (e & 8u) return set_error(c, ELP_ERR_DATA, "BQSR: base quality above 93");
  if (e & 16u) return set_error(c, ELP_ERR_DATA, "cycle value exceeds maximum cycle value (reference: log.Panic, filters/bqsr.go:364-369)");
  if (e & 32u) return set_error(c, ELP_ERR_DATA, "BQSR requires input with read groups (reference: log.Panic, filters/bqsr.go:38)");
  if (e & 64u) return set_error(c, ELP_ERR_DATA, "ApplyBQSR: len(QUAL) != len(SEQ) (reference: index out of range panic)");
  return set_error(c, ELP_ERR_DATA, "BQSR: device error word %u", e);
}

static int sync_bqsr_ptrs(elp_ctx *c) {
  if (!c->bqsr_ptrs_dirty) return 0;
  const size_t nr = (size_t)c->n_ref;
  ELP_TRY(ensure(c, c->d_ref_seq, nr + 1));
  ELP_TRY(ensure(c, c->d_ref_seq_len, nr + 1));
  ELP_TRY(ensure(c, c->d_sites, nr + 1));
  ELP_TRY(ensure(c, c->d_n_sites, nr + 1));
  if (nr) {
    ELP_HIP(c, hipMemcpyAsync(c->d_ref_seq.p, c->h_ref_seq.data(), nr * sizeof(uint8_t *), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipMemcpyAsync(c->d_ref_seq_len.p, c->h_ref_seq_len.data(), nr * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipMemcpyAsync(c->d_sites.p, c->h_sites.data(), nr * sizeof(int32_t *), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipMemcpyAsync(c->d_n_sites.p, c->h_n_sites.data(), nr * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipStreamSynchronize(c->stream));
  }
  c->bqsr_ptrs_dirty = false;
  return 0;
}

static int gather_impl(elp_ctx *c, int max_cycle, int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl) {
  for (int r = 0; r < c->n_ref; r++)
    if (!c->h_ref_seq[r]) return set_error(c, ELP_ERR_ARG, "elp_bqsr_gather: no reference sequence set for refid %d", r);
  for (int r = 0; r < c->n_ref; r++)
    if (!c->h_sites[r]) ELP_TRY(elp_bqsr_set_known_sites(c, r, nullptr, 0));
  ELP_TRY(sync_bqsr_ptrs(c));
  ELP_TRY(ensure_adapted(c));  // provides the set of quality values present (qual_present)
  if (c->n_cov > 255) return set_error(c, ELP_ERR_UNSUPPORTED, "more than 255 read-group covariates");
  if (c->cigar_ops + 4 * c->n >= 0x7FFFFFFFull) return set_error(c, ELP_ERR_UNSUPPORTED, "CIGAR pool exceeds 2^31 operations per context");
  const int ncyc_g = 2 * max_cycle + 1;
  const size_t nq = (size_t)c->n_cov * ELP_NQUAL * 2, nc = nq * ncyc_g, nx = nq * ELP_NCTX;
  unsigned long long *tb;
  ELP_TRY(scratch(c, 0, nq + nc + nx + 8, &tb));
  hipStream_t st = c->stream;
  ELP_HIP(c, hipMemsetAsync(tb, 0, (nq + nc + nx) * sizeof(unsigned long long), st));
  const uint64_t n = c->n;
  if (n && c->qual_bytes) {
    uint32_t *cs_pool;
    ELP_TRY(scratch(c, 1, 2 * (c->cigar_ops + 4 * n) + 64, &cs_pool));
    BqDesc *desc;
    ELP_TRY(scratch(c, 2, n + 4, &desc));
    uint32_t *skipbits;
    const size_t skip_words = (size_t)((c->qual_bytes + 31) / 32 + 8);
    ELP_TRY(scratch(c, 3, skip_words, &skipbits));
    ELP_HIP(c, hipMemsetAsync(skipbits, 0, skip_words * 4, st));
    BqCols m{n, c->refid.p, c->pos.p, c->next_refid.p, c->pnext.p, c->tlen.p, c->flag.p, c->rgid.p, c->mapq.p, c->has_sr.p, c->l_seq.p,
             c->cigar_off.p, c->seq_off.p, c->qual_off.p, c->cigar.p, c->seq4.p, c->qual.p, c->ref_len.p, c->rg_cov.p, c->n_ref,
             c->d_ref_seq.p, c->d_ref_seq_len.p, c->d_sites.p, c->d_n_sites.p};
    ELP_LAUNCH(c, "bqsr_prologue", k_bqsr_prologue, dim3(blocks_for(n, 256)), dim3(256), 0, m, cs_pool, desc, skipbits, c->err_flag.p);
    // quality values present (>= 6, <= 93) -> passes of at most `qcap` slots so the private tables fit in LDS
    std::vector<int> quals;
    for (int q = 6; q < ELP_NQUAL; q++)
      if ((q < 64 ? (c->qual_present[0] >> q) : (c->qual_present[1] >> (q - 64))) & 1ull) quals.push_back(q);
    const int lmax = (int)std::max<uint32_t>(c->max_l_seq, 1);
    const int s16 = (2 * lmax + 1 + 15) / 16, cs = 16 * s16;
    // dynamic LDS for the private tables: 160 KiB per CU / 3 workgroups, minus the kernel's static LDS (offsets + descriptors)
    const size_t static_lds = sizeof(FlatLds) + (size_t)FL_RMAX * (sizeof(BqDesc) + 8 + 4);
    const size_t lds_budget = 53 * 1024 - static_lds - 256;
    const size_t per_slot = (size_t)c->n_cov * ((size_t)cs + 32) * 4;
    int qcap = (int)(lds_budget / std::max<size_t>(per_slot, 1));
    if (qcap < 1) return set_error(c, ELP_ERR_UNSUPPORTED, "BQSR private tables do not fit in LDS (n_cov=%d, max read length=%d)", c->n_cov, lmax);
    const uint64_t ntiles = (c->qual_bytes + FL_TILE - 1) / FL_TILE;
    const int grid = (int)std::min<uint64_t>(ntiles, 768);  // 3 workgroups per CU
    for (size_t q0 = 0; q0 < quals.size(); q0 += (size_t)qcap) {
      const int nqs = (int)std::min<size_t>((size_t)qcap, quals.size() - q0);
      QMap qm, s2q;
      memset(qm.slot, 255, sizeof qm.slot);
      memset(s2q.slot, 0, sizeof s2q.slot);
      for (int s = 0; s < nqs; s++) { qm.slot[quals[q0 + s]] = (uint8_t)s; s2q.slot[s] = (uint8_t)quals[q0 + s]; }
      const int ncq = c->n_cov * nqs;
      const size_t cells = (size_t)ncq * cs + 2 * (size_t)ncq * 16;
      const size_t part_words = (size_t)grid * (2 * (size_t)ncq * cs + 2 * (size_t)ncq * 16);
      uint32_t *partial;
      ELP_TRY(scratch(c, 4, part_words + 16, &partial));
      ELP_HIP(c, hipMemsetAsync(partial, 0, part_words * 4, st));
      CountArgs A{n, c->qual_bytes, c->qual_off.p, c->seq_off.p, c->qual.p, c->seq4.p, c->refid.p, desc, c->cigar.p, cs_pool,
                  reinterpret_cast<const uint16_t *>(skipbits), c->d_ref_seq.p, c->d_ref_seq_len.p, c->n_cov, nqs, lmax, cs, s16, max_cycle,
                  partial, c->err_flag.p, c->tile_first.p};
      ELP_LAUNCH(c, "bqsr_count", k_bqsr_count, dim3(grid), dim3(FL_THREADS), cells * 4, A, qm);
      const int total = ncq * (2 * lmax + 1) + ncq * 16;
      ELP_LAUNCH(c, "bqsr_reduce", k_bqsr_reduce, dim3(blocks_for(total, 256)), dim3(256), 0, (const uint32_t *)partial, grid, c->n_cov, nqs, lmax, cs,
                 s16, max_cycle, s2q, tb + nq, tb + nq + nc);
    }
    ELP_LAUNCH(c, "bqsr_qual_from_cycle", k_bqsr_qual_from_cycle, dim3(c->n_cov * ELP_NQUAL), dim3(256), 0, c->n_cov * ELP_NQUAL, ncyc_g,
               (const unsigned long long *)(tb + nq), tb);
  }
  ELP_HIP(c, hipMemcpyAsync(qual_tbl, tb, nq * 8, hipMemcpyDeviceToHost, st));
  ELP_HIP(c, hipMemcpyAsync(cycle_tbl, tb + nq, nc * 8, hipMemcpyDeviceToHost, st));
  ELP_HIP(c, hipMemcpyAsync(ctx_tbl, tb + nq + nc, nx * 8, hipMemcpyDeviceToHost, st));
  uint32_t e[4];
  ELP_TRY(fetch_err(c, e));
  if (e[0]) return bqsr_error(c, e[0]);
  return 0;
}

}  // namespace elp

using namespace elp;

extern "C" {

int elp_bqsr_set_reference(elp_ctx *c, int32_t refid, const uint8_t *bases, int64_t len) {
  if (!c || !c->have_header || refid < 0 || refid >= c->n_ref || len < 0 || (len && !bases)) return set_error(c, ELP_ERR_ARG, "elp_bqsr_set_reference: bad arguments");
  ELP_HIP(c, hipSetDevice(c->device));
  if (c->h_ref_seq[refid]) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->h_ref_seq[refid]); c->h_ref_seq[refid] = nullptr; }
  uint8_t *d = nullptr;
  ELP_HIP(c, hipMalloc((void **)&d, (size_t)len + 16));
  if (len) ELP_HIP(c, hipMemcpyAsync(d, bases, (size_t)len, hipMemcpyHostToDevice, c->stream));
  ELP_HIP(c, hipStreamSynchronize(c->stream));
  c->h_ref_seq[refid] = d;
  c->h_ref_seq_len[refid] = len;
  c->bqsr_ptrs_dirty = true;
  return 0;
}

int elp_bqsr_set_known_sites(elp_ctx *c, int32_t refid, const int32_t *start_end, int64_t n) {
  if (!c || !c->have_header || refid < 0 || refid >= c->n_ref || n < 0 || (n && !start_end)) return set_error(c, ELP_ERR_ARG, "elp_bqsr_set_known_sites: bad arguments");
  for (int64_t k = 1; k < n; k++)
    if (!(start_end[2 * k] > start_end[2 * k - 1])) return set_error(c, ELP_ERR_ARG, "known sites of refid %d are not sorted and flattened at index %lld", refid, (long long)k);
  ELP_HIP(c, hipSetDevice(c->device));
  if (c->h_sites[refid]) { (void)hipStreamSynchronize(c->stream); (void)hipFree(c->h_sites[refid]); c->h_sites[refid] = nullptr; }
  int32_t *d = nullptr;
  ELP_HIP(c, hipMalloc((void **)&d, (size_t)(2 * n + 4) * sizeof(int32_t)));
  if (n) ELP_HIP(c, hipMemcpyAsync(d, start_end, (size_t)(2 * n) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  ELP_HIP(c, hipStreamSynchronize(c->stream));
  c->h_sites[refid] = d;
  c->h_n_sites[refid] = n;
  c->bqsr_ptrs_dirty = true;
  return 0;
}

int elp_bqsr_gather(elp_ctx *c, int max_cycle, int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl) {
  if (!c || !qual_tbl || !cycle_tbl || !ctx_tbl || max_cycle < 1) return set_error(c, ELP_ERR_ARG, "elp_bqsr_gather: bad arguments");
  ELP_HIP(c, hipSetDevice(c->device));
  return gather_impl(c, max_cycle, qual_tbl, cycle_tbl, ctx_tbl);
}

int elp_bqsr_apply(elp_ctx *c, int max_cycle, const uint8_t *lut, const uint8_t *cov_present) {
  if (!c || !lut || !cov_present || max_cycle < 1) return set_error(c, ELP_ERR_ARG, "elp_bqsr_apply: bad arguments");
  ELP_HIP(c, hipSetDevice(c->device));
  if (c->n_cov > 255) return set_error(c, ELP_ERR_UNSUPPORTED, "more than 255 read-group covariates");
  const size_t ncyc = 2 * (size_t)max_cycle + 1;
  const size_t lut_bytes = (size_t)c->n_cov * ELP_NQUAL * ncyc * 17;
  uint8_t *dl;
  ELP_TRY(scratch(c, 0, lut_bytes + (size_t)c->n_cov + 64, &dl));
  ELP_HIP(c, hipMemcpyAsync(dl, lut, lut_bytes, hipMemcpyHostToDevice, c->stream));
  ELP_HIP(c, hipMemcpyAsync(dl + lut_bytes, cov_present, (size_t)c->n_cov, hipMemcpyHostToDevice, c->stream));
  const uint64_t n = c->n;
  if (n) {
    ApDesc *desc;
    ELP_TRY(scratch(c, 2, n + 4, &desc));
    ELP_LAUNCH(c, "bqsr_apply_prologue", k_apply_prologue, dim3(blocks_for(n, 256)), dim3(256), 0, n, (const uint16_t *)c->flag.p,
               (const uint16_t *)c->rgid.p, (const uint16_t *)c->rg_cov.p, (const uint32_t *)c->l_seq.p, (const uint64_t *)c->qual_off.p,
               (const uint8_t *)c->qual.p, (const uint8_t *)(dl + lut_bytes), desc, c->err_flag.p);
    if (c->qual_bytes) {
      const uint64_t ntiles = (c->qual_bytes + FL_TILE - 1) / FL_TILE;
      const unsigned grid = (unsigned)std::min<uint64_t>(ntiles, 256 * 8);
      ELP_TRY(ensure_flat_index(c));
      ApplyArgs A{n, c->qual_bytes, c->qual_off.p, c->seq_off.p, c->qual.p, c->seq4.p, c->l_seq.p, desc, c->tile_first.p, dl, max_cycle, c->err_flag.p};
      ELP_LAUNCH(c, "bqsr_apply", k_bqsr_apply_flat, dim3(grid), dim3(FL_THREADS), 0, A);
    }
  }
  uint32_t e[4];
  ELP_TRY(fetch_err(c, e));
  if (e[0]) return bqsr_error(c, e[0]);
  c->adapted = false;  // scores depend on QUAL
  return 0;
}

}  // extern "C"
