// bqsr.hip — base quality score recalibration on the HBM column store: covariate-table gather and LUT apply.
//
// Reference: BaseRecalibrator.Recalibrate (filters/bqsr.go:467-551), BaseRecalibratorTables.ApplyBQSR (:936-1005); device
// helpers (clipping, covariates) and their citations are in bqsr_dev.hpp.
//
// Gather = two kernels
//   bqsr_prologue  one thread per record: eligibility (recalibrateAln), adaptor + soft-clip hard clipping on a working copy,
//                  known-site skip mask (calculateSkipSlice) written into a 1-bit-per-base column, low-quality-tail bounds, and
//                  the clipped CIGAR folded into <= 3 (clipped base -> reference index) pieces; leaves a 32-byte descriptor.
//   bqsr_count     flat stream over QUAL/SEQ (flat.hpp): one lane per 16 bases.  Per chunk, SWAR in nibble space gives the
//                  eligible-base flags, the SNP flags (one XOR of 16 read nibbles with 16 nibbles of the 4-bit packed
//                  reference), the cycle parameters and the 16 context keys; then per counted base ONE packed 32-bit LDS
//                  atomic into the workgroup-private cycle table and ONE 64-bit LDS atomic into the private context table.
//                  Private tables are added into the dense int64 tables in HBM with global atomics when a workgroup is done
//                  (and every 50000 reads).  QualityScores is the sum of the Cycles table over cycles.
// Apply = a small per-record prologue (low-quality bounds) + a flat per-base LUT kernel that rewrites QUAL in place (16
// independent byte gathers from the L2-resident LUT per lane and chunk).
#include <utility>

#include "bqsr_common.hpp"

namespace elp {

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
  uint32_t *const *site_idx;  // per contig and 64-bp bucket: first site whose end is >= 64 * bucket (k_site_index)
  const uint64_t *qbounds;    // per record: low-quality-tail bounds of the full read (adapt_score)
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

// pieces of the clipped CIGAR; false if it needs more than three
__device__ inline bool build_pieces(const uint32_t *cig, int ncig, int32_t pos, BqDesc &d) {
  int64_t val[3];
  bool noref[3];
  int start[3];
  int np = 0;
  int c = 0;
  int64_t delta = (int64_t)pos - 1;  // reference index minus clipped read index for the current match run
  for (int i = 0; i < ncig; i++) {
    const uint32_t op = c_op(cig[i]);
    const int ln = c_len(cig[i]);
    if (op == OP_M || op == OP_EQ || op == OP_X) {
      if (ln > 0 && (np == 0 || noref[np - 1] || val[np - 1] != delta)) {
        if (np == 3) return false;
        val[np] = delta; noref[np] = false; start[np] = c; np++;
      }
      c += ln;
    } else if (op == OP_I || op == OP_S) {
      if (ln > 0 && (np == 0 || !noref[np - 1])) {
        if (np == 3) return false;
        val[np] = 0; noref[np] = true; start[np] = c; np++;
      }
      c += ln;
      delta -= ln;
    } else if (op == OP_D || op == OP_N) {
      delta += ln;
    }
  }
  int32_t D[3] = {BQ_NOREF, BQ_NOREF, BQ_NOREF};
  for (int k = 0; k < np; k++) {
    if (noref[k]) continue;
    if (val[k] <= (int64_t)INT32_MIN + 70000 || val[k] >= (int64_t)INT32_MAX - 70000) return false;
    D[k] = (int32_t)val[k];
  }
  d.D0 = D[0]; d.D1 = D[1]; d.D2 = D[2];
  d.b1 = np > 1 ? (uint16_t)start[1] : (uint16_t)0xFFFF;
  d.b2 = np > 2 ? (uint16_t)start[2] : (uint16_t)0xFFFF;
  return true;
}

// marks bases [fs, fe] of the record whose first QUAL byte is at bit0 in the skip column
__device__ __forceinline__ void set_skip_bits(uint32_t *skipbits, uint64_t bit0, int fs, int fe) {
  for (int k = fs; k <= fe;) {  // word by word
    const uint64_t b = bit0 + (uint64_t)k;
    const int in_word = (int)(b & 31);
    int cnt = 32 - in_word;
    if (cnt > fe - k + 1) cnt = fe - k + 1;
    const uint32_t mask = (cnt == 32 ? 0xFFFFFFFFu : ((1u << cnt) - 1u)) << in_word;
    atomicOr(&skipbits[b >> 5], mask);
    k += cnt;
  }
}

// clears the bits of bases [0, nbits) of the record whose first QUAL byte is at bit0.  Round 6: when RECORDS are written (count3.hip) only the
// reads that put known-site bits into the column ever read it (RC_SKIPCOL), and only their own bits - such a read clears its range before
// it sets bits, and the fill of the whole column (a bit per staged base: 0.14 ms per 50 M reads) is made for the descriptor form only.
// Neighbouring reads share words, never bits: atomics on the words, in program order per thread.
__device__ __forceinline__ void clear_skip_bits(uint32_t *skipbits, uint64_t bit0, uint32_t nbits) {
  for (uint32_t k = 0; k < nbits;) {
    const uint64_t b = bit0 + (uint64_t)k;
    const uint32_t in_word = (uint32_t)(b & 31);
    uint32_t cnt = 32 - in_word;
    if (cnt > nbits - k) cnt = nbits - k;
    const uint32_t mask = (cnt == 32 ? 0xFFFFFFFFu : ((1u << cnt) - 1u)) << in_word;
    atomicAnd(&skipbits[b >> 5], ~mask);
    k += cnt;
  }
}

// Fast prologue, one thread per record.  Decides eligibility (recalibrateAln) for every record and finishes the records whose
// CIGAR is a single M/=/X operation and that need no adaptor clipping (≈ 5 of 6 reads): for those the clipped copy is the read
// itself, getReadCoordinateForReferenceCoordinate(ref) is ref - POS inside the read and fails outside (utils.go:267-349 with one
// match operation), and there is one reference piece.  Everything else is appended to `queue` for the general kernel, so that
// kernel's long divergent code runs with all lanes busy.  All column loads are issued before the first test (one latency, not 15).
// (PF_TILES: bqsr_common.hpp)
// a record's columns (k_bqsr_prologue_fast loads them a tile ahead)
struct PfCols {
  uint8_t has_sr, mq;
  uint16_t f, rg;
  int32_t r, p, pnext, tlen, nrefid;
  uint32_t ls;
  uint64_t q0, q1, c0, c1, qb;
};
__device__ __forceinline__ PfCols pf_load_cols(const BqCols &m, uint64_t i) {
  PfCols c;
  c.has_sr = m.has_sr[i]; c.mq = m.mapq[i]; c.f = m.flag[i]; c.rg = m.rgid[i];
  c.r = m.refid[i]; c.p = m.pos[i]; c.pnext = m.pnext[i]; c.tlen = m.tlen[i]; c.nrefid = m.next_refid[i];
  c.ls = m.l_seq[i];
  c.q0 = m.qual_off[i]; c.q1 = m.qual_off[i + 1]; c.c0 = m.cigar_off[i]; c.c1 = m.cigar_off[i + 1];
  c.qb = m.qbounds[i];
  return c;
}
// per-contig facts in LDS (contig length, known-site array, its length, its bucket index): the walk over the known sites then depends on
// ONE global round trip (the bucket entry) instead of three (pointer tables first)
struct PfLds {
  int32_t ref_len[REF_LDS];
  const int32_t *sites[REF_LDS];
  int64_t nsites[REF_LDS];
  const uint32_t *sidx[REF_LDS];
};
__device__ __forceinline__ void pf_lds_fill(PfLds &L, const BqCols &m, int nt) {
  if (m.n_ref <= REF_LDS)
    for (int r = threadIdx.x; r < m.n_ref; r += nt) { L.ref_len[r] = m.ref_len[r]; L.sites[r] = m.sites[r]; L.nsites[r] = m.n_sites[r]; L.sidx[r] = m.site_idx[r]; }
}

// The prologue of ONE record.  PLAIN_PASS false: the first, streaming pass - finishes the reads whose CIGAR is [H] [S] <match> [S] [H], sends
// reads of match / insertion / deletion operations to the second pass (to_plain) and everything else to the general kernel (defer).
// PLAIN_PASS true: the second pass over the reads the first one listed; my_cig_w = five LDS words of the thread.
template <bool PLAIN_PASS>
__device__ __forceinline__ void pf_record(const BqCols &m, const uint64_t i, const PfCols &cur, const PfLds &L, const bool ref_lds, BqDesc *__restrict__ desc,
                                          uint32_t *skipbits, uint32_t *err, const bool recs /* records, not descriptors */, uint32_t *my_cig_w, bool &defer, bool &to_plain,
                                          BqRec &rc_out, int &rc_class, uint4 *__restrict__ plain_rec /* first pass: where a read of the second pass leaves its columns */,
                                          const uint32_t *pre_ops /* second pass: the read's five CIGAR operation slots, handed over */) {
    const uint8_t has_sr = cur.has_sr, mq = cur.mq;
    const uint16_t f = cur.f, rg = cur.rg;
    const int32_t r = cur.r, p = cur.p, pnext = cur.pnext, tlen = cur.tlen, nrefid = cur.nrefid;
    const uint32_t ls = cur.ls;
    const uint64_t q0 = cur.q0, q1 = cur.q1, c0 = cur.c0, c1 = cur.c1;
    const uint64_t qb = cur.qb;
    BqDesc d;
    d.D0 = d.D1 = d.D2 = BQ_NOREF; d.refid = 0; d.b1 = d.b2 = 0xFFFF; d.a = 0; d.len = 0; d.left = 0; d.right = 0; d.cov = 0; d.fl = 0; d.pad = 0;
    const uint8_t *rec_rp = nullptr;
    int64_t rec_rlen = 0;
    bool used_col = false, rec_skipped_walk = false;  // known-site bits of the read went into the skip column; no walk: they come with the reference window
    // recalibrateAln, bqsr.go:225-244 (+ utils.go:121-139), the part that needs no dependent load
    bool ok = !has_sr && mq > 0 && mq < 255 && !(f & (F_SECONDARY | F_DUPLICATE | F_QCFAILED)) && !(f & F_UNMAPPED) && r >= 0 && p > 0 && ls != 0 &&
              (uint64_t)ls == q1 - q0 && rg != ELP_NIL16 && r < m.n_ref;
    if (ok) {
      // CIGARs of the form [H] [S] <match> [S] [H] (one M/=/X operation, clips only at the ends: plain reads and soft-clipped ones):
      // hardClipSoftClippedBases (utils.go:519-548) leaves the match operation between hard clips, i.e. the clipped copy is bases
      // [aoff, aoff + len) of the read with ONE reference piece starting at POS, softStart = POS, softEnd = End, and
      // getReadCoordinateForReferenceCoordinate is ref - POS inside it.  All five operation slots are read at once.
      const uint64_t nop = c1 - c0;
      if (recs) { rec_rp = m.ref_seq[r]; rec_rlen = m.ref_seq_len[r]; }  // issued with the CIGAR loads: one round trip for both
      uint32_t opv[5];
#pragma unroll
      for (int k = 0; k < 5; k++) opv[k] = PLAIN_PASS ? pre_ops[k] : ((uint64_t)k < nop ? m.cigar[c0 + k] : 0u);
      const int32_t rl = ref_lds ? L.ref_len[r] : m.ref_len[r];
      // the same round trip: the read group's covariate index and - when descriptors are written - the known-site bucket entry (read
      // whether or not the tests below pass).  When RECORDS are written (count3.hip) a read that is one run of matches needs no walk over
      // the site list: its known-site bits come with the reference window (k_ref_mark_sites); the bucket entry is then fetched only by the
      // reads that do walk (indels; a window the record cannot describe)
      const uint16_t cov_rg = m.rg_cov[rg];
      const int32_t *sv = ref_lds ? L.sites[r] : m.sites[r];
      const int64_t ns = ref_lds ? L.nsites[r] : m.n_sites[r];
      auto bucket_entry = [&]() __attribute__((always_inline)) -> int64_t {
        const int64_t nbuck = ((int64_t)rl >> 6) + 1;
        int64_t bk = (int64_t)(p < rl ? p : rl) >> 6;
        bk = bk >= nbuck ? nbuck - 1 : bk;
        return (int64_t)(ref_lds ? L.sidx[r] : m.site_idx[r])[bk];
      };
      int64_t s_first = 0;
      if (!recs && ns > 0) s_first = bucket_entry();
      ok = p <= rl;
      bool simple = nop >= 1 && nop <= 5;
      uint32_t aoff = 0, mlen = 0, trail = 0;
      {
        uint32_t k = 0;
        if (simple && c_op(opv[0]) == OP_H) k = 1;
        // (select chains on purpose: opv[] indexed by a variable would move the array to scratch memory)
        auto at = [&](uint32_t j) { return j == 0 ? opv[0] : (j == 1 ? opv[1] : (j == 2 ? opv[2] : (j == 3 ? opv[3] : opv[4]))); };
        if (simple && k < nop && c_op(at(k)) == OP_S) { aoff = (uint32_t)c_len(at(k)); k++; }
        if (simple && k < nop && (c_op(at(k)) == OP_M || c_op(at(k)) == OP_EQ || c_op(at(k)) == OP_X)) { mlen = (uint32_t)c_len(at(k)); k++; }
        else simple = false;
        if (simple && k < nop && c_op(at(k)) == OP_S) { trail = (uint32_t)c_len(at(k)); k++; }
        if (simple && k < nop && c_op(at(k)) == OP_H) k++;
        simple = simple && k == nop && mlen != 0;
      }
      // CIGARs of match / insertion / deletion operations only (two to five of them: reads with an indel or two): nothing is clipped
      // unless the adaptor test says so, the window is the whole read; reference pieces and read coordinates of known sites come
      // from the same device functions the general kernel uses, on the CIGAR as staged
      bool plain = false;
      uint32_t plain_read = 0, plain_ref = 0;
      if (!simple && nop >= 2 && nop <= 5) {
        plain = true;
#pragma unroll
        for (int k = 0; k < 5; k++) {
          if ((uint64_t)k < nop) {
            const uint32_t o = c_op(opv[k]), ln = (uint32_t)c_len(opv[k]);
            if (o == OP_M || o == OP_EQ || o == OP_X) { plain_read += ln; plain_ref += ln; }
            else if (o == OP_I) plain_read += ln;
            else if (o == OP_D) plain_ref += ln;
            else plain = false;
          }
        }
      }
      if (!PLAIN_PASS) {
        // first pass: a read with indels goes to the second, dense pass (k_bqsr_prologue_plain) - a wave that holds one would otherwise
        // run the piece / read-coordinate code for all of its lanes (the kernel is bound by vector issue: 1250 instructions per wave and
        // tile with both paths in one kernel, profiles/round3 PMC)
        if (ok && plain && ls <= (uint32_t)MAX_DESC_READ) {
          // the columns this thread holds go along in ONE 64-byte line at the read's own place (round 6): the second pass gathered them
          // again from fifteen columns - 850 bytes of sectors per listed read, 0.52 ms for the 7.5 % of the reads that have an indel
          uint4 *pr = plain_rec + 4 * i;
          pr[0] = make_uint4(opv[0], opv[1], opv[2], opv[3]);
          pr[1] = make_uint4(opv[4], (uint32_t)f | ((uint32_t)rg << 16), (uint32_t)r, (uint32_t)p);
          pr[2] = make_uint4(ls | ((uint32_t)nop << 16) | (nrefid < 0 ? 1u << 19 : 0u), (uint32_t)c0, (uint32_t)q0, (uint32_t)(q0 >> 32));
          pr[3] = make_uint4((uint32_t)qb, (uint32_t)(qb >> 32), (uint32_t)pnext, (uint32_t)tlen);
          to_plain = true;
          return;
        }
        plain = false;
      } else if (plain) {
#pragma unroll
        for (int k = 0; k < 5; k++) my_cig_w[k] = opv[k];
      }
      const uint32_t *my_cig = my_cig_w;
      if (ok && !((simple || plain) && ls <= (uint32_t)MAX_DESC_READ)) { defer = true; ok = false; }  // the general kernel redoes the tests
      if (ok) ok = (plain ? plain_read : aoff + mlen + trail) == ls;  // SEQ length == CIGAR read length (utils.go:121-128)
      if (ok) {
        const int len = plain ? (int)ls : (int)mlen;
        const int32_t end = p + (plain ? (int)plain_ref : len) - 1;  // aln.End() (sam/sam-types.go:769-775)
        // hardClipAdaptorSequence (utils.go:149-180, 214-222) would clip?
        const bool rev = f & F_REVERSED;
        bool clip = false;
        if (tlen != 0 && (f & F_MULTIPLE) && !((f & F_NEXT_UNMAPPED) || nrefid < 0 || pnext == 0) && rev != (bool)(f & F_NEXT_REVERSED)) {
          const bool well = rev ? end > pnext : p <= pnext + tlen;
          if (well) {
            const int boundary = rev ? (int)pnext - 1 : (int)p + (tlen < 0 ? -(int)tlen : (int)tlen);
            clip = boundary >= (int)p && boundary <= (int)end;
          }
        }
        // computeStrandedClippedSeq mask bounds (bqsr.go:316-332) inside the window: adapt_score recorded the first / last quality > 2
        // of the whole read; where that lies outside the window the window's own end decides (if it does not: general kernel)
        const uint32_t hi1 = (uint32_t)qb;
        int left = len, right = len - 1;
        if (!clip && hi1) {
          const int f0 = (int)(qb >> 32) - (int)aoff, l0 = (int)hi1 - 1 - (int)aoff;  // relative to the window; f0 <= l0
          if (f0 < len && l0 >= 0) {
            if (f0 >= 0) left = f0;
            else if (m.qual[q0 + aoff] > 2) left = 0;
            else clip = true;
            if (l0 < len) right = l0;
            else if (m.qual[q0 + aoff + (uint32_t)len - 1] > 2) right = len - 1;
            else clip = true;
          }
        }
        if (clip) {
          defer = true;
        } else {
          // calculateSkipSlice (bqsr.go:389-414): softStart = POS, softEnd = End
          // (records: a plain run of matches whose window the record describes takes its known-site bits from the reference window)
          const bool rec_simple = recs && !plain && len <= 1022 && (int64_t)p - 1 - (int64_t)aoff >= 16 && (int64_t)p - 1 - (int64_t)aoff + (int64_t)ls <= rec_rlen + 32;
          rec_skipped_walk = ns > 0 && rec_simple;
          if (ns > 0 && !rec_simple) {
            if (recs) s_first = bucket_entry();
            // the first two candidate sites in one round trip (most reads touch none or one); any further ones from memory
            const int2 *sv2 = reinterpret_cast<const int2 *>(sv);
            int64_t s = s_first;
            const int2 cand0 = s < ns ? sv2[s] : make_int2(0, 0), cand1 = s + 1 < ns ? sv2[s + 1] : make_int2(0, 0);
            int sk_x = cand0.x, sk_y = cand0.y;  // site s
            auto fetch = [&]() __attribute__((always_inline)) {
              if (s == s_first + 1) { sk_x = cand1.x; sk_y = cand1.y; }
              else if (s < ns) { const int2 t = sv2[s]; sk_x = t.x; sk_y = t.y; }
            };
            while (s < ns && sk_y < p) { s++; fetch(); }
            for (; s < ns && sk_x <= end; s++, fetch()) {
              struct { int x, y; } sk = {sk_x, sk_y};
              int fs, fe;
              if (plain) {
                bool okc;
                fs = get_read_coord(my_cig, (int)nop, (int)p, sk.x, false, &okc);
                if (!okc || fs < 0) fs = 0;
                fe = get_read_coord(my_cig, (int)nop, (int)p, sk.y, false, &okc);
                if (!okc || fe > len - 1) fe = len - 1;
              } else {
                const int a0 = sk.x - p, a1 = sk.y - p;
                fs = (a0 < 0 || a0 >= len) ? 0 : a0;          // !ok || < 0 -> 0
                fe = (a1 < 0 || a1 >= len) ? len - 1 : a1;    // !ok || > len-1 -> len-1 (a1 < 0 cannot happen: End >= POS)
              }
              if (recs && !used_col) clear_skip_bits(skipbits, q0, ls);
              set_skip_bits(skipbits, q0 + aoff, fs, fe);
              used_col = true;
            }
          }
          d.D0 = p - 1;
          d.refid = r;
          d.a = (uint16_t)aoff;
          d.len = (uint16_t)len;
          uint8_t complex_fl = 0;
          if (plain && !build_pieces(my_cig, (int)nop, p, d)) {  // more than three pieces: the count kernel walks the CIGAR
            complex_fl = BQ_COMPLEX;
            d.D0 = (int32_t)c0;
            d.b1 = (uint16_t)nop;
            d.D2 = p - 1;
          }
          d.left = (uint16_t)left; d.right = (uint16_t)(right < 0 ? 0xFFFF : right);
          d.cov = (uint8_t)cov_rg;
          d.fl = BQ_ELIGIBLE | (rev ? BQ_REVERSED : 0) | ((f & F_LAST) ? BQ_LAST : 0) | complex_fl;
        }
      }
    }
    if (!defer) {
      if (recs) {  // the record count3.hip works from; BqDesc only for the reads the record cannot describe
        BqRec rc;
        rc.ref_lo = rc.ref_hi = rc.win = rc.ctxw = 0; rc.t0 = 0; rc.fl = rc.bpk = rc.dpk = 0;
        if (d.fl & BQ_ELIGIBLE) {
          Pieces4 P;
          if (d.fl & BQ_COMPLEX) pieces4(my_cig_w, (int)(c1 - c0), p, P);
          else if (d.b1 != 0xFFFFu) pieces4(my_cig_w, (int)(c1 - c0), p, P);
          else { P.v0 = (int64_t)d.D0; P.v1 = P.v2 = P.v3 = 0; P.s1 = P.s2 = P.s3 = 0; P.noref = d.D0 == BQ_NOREF ? 1u : 0u; P.np = 1; }
          rc = make_rec((int)d.a, (int)d.len, (int)d.left, d.right == 0xFFFFu ? -1 : (int)d.right, d.cov, (d.fl & BQ_REVERSED) != 0, (d.fl & BQ_LAST) != 0, P,
                        P.np < 0, rec_rp, rec_rlen, (int64_t)ls);
          if (used_col) rc.fl |= RC_SKIPCOL;
          else if ((rc.fl & RC_GENERAL) && rec_skipped_walk) atomicOr(&err[0], 1024u);  // (cannot happen: rec_simple restates make_rec's tests)
        }
        if (rc.fl & RC_GENERAL) desc[i] = d;
        rc_class = rec_class(rc);
        rec_pack_idx(rc, (uint32_t)i);
        rc_out = rc;
      } else {
        desc[i] = d;
      }
    }
}

// First pass, one thread per record; all column loads are issued before the first test (one latency, not 15), a tile ahead.
__global__ __launch_bounds__(256) void k_bqsr_prologue_fast(BqCols m, BqDesc *__restrict__ desc, uint32_t *skipbits, uint32_t *__restrict__ queue,
                                                            uint32_t *queue_n, uint32_t *err, RecOut ro, uint32_t *__restrict__ plist, uint4 *__restrict__ plain_rec) {
  // a workgroup handles PF_TILES * 256 consecutive records and collects the deferred ones in LDS - the general kernel's from the front of
  // the list, the second pass's from its end: one global atomic per workgroup and list (a global atomic per wave on a single counter
  // serialises at ~12 ns each: 9 ms for 50 M reads)
  __shared__ uint32_t lq[PF_TILES * 256];
  __shared__ uint32_t lcount, gbase, pcount, pbase;
  __shared__ uint32_t seg_n[C3_MAXSEG], seg_at[C3_MAXSEG];  // covariate-split segments: the tile's class-1 records per covariate, their first place
  __shared__ PfLds L;
  const bool ref_lds = m.n_ref <= REF_LDS;
  pf_lds_fill(L, m, 256);
  if (threadIdx.x == 0) { lcount = 0; pcount = 0; }
  seg_n[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t i_first = (uint64_t)blockIdx.x * PF_TILES * 256 + threadIdx.x;
  PfCols nxt = {};
  if (i_first < m.n) nxt = pf_load_cols(m, i_first);
#pragma unroll 1
  for (int tile = 0; tile < PF_TILES; tile++) {
    const uint64_t i = i_first + (uint64_t)tile * 256;
    bool defer = false, to_plain = false;
    const PfCols cur = nxt;
    if (tile + 1 < PF_TILES && i + 256 < m.n) nxt = pf_load_cols(m, i + 256);
    BqRec rc;
    int rcl = 0;
    if (i < m.n) pf_record<false>(m, i, cur, L, ref_lds, desc, skipbits, err, ro.recs != nullptr, nullptr, defer, to_plain, rc, rcl, plain_rec, nullptr);
    if (ro.recs) {  // the tile's records, compacted: class 1 into this wave's segment, the rare class 2 ones (windows the record cannot describe) behind
      if (!ro.ncs) {
        const uint32_t seg = (blockIdx.x * 4u + (threadIdx.x >> 6)) % ro.nseg;  // the wave's segment
        const uint32_t at1 = wave_append(rcl == 1, &ro.cnt[seg * C3_CSTRIDE]);
        if (rcl == 1) rec_store(ro.recs, (uint64_t)ro.seg_base[seg] + at1, rc);
      } else {
        // segments by covariate: the workgroup's class-1 records of this tile take their rank among those of their covariate from a
        // returning LDS atomic, then ONE global atomic per covariate that occurs reserves the places in segment (workgroup % groups, covariate)
        // (measured with the reservation taken out: the global atomics WERE the cost of many read groups - one per wave and covariate,
        // 4 M / 7 M of them at 16 / 32 read groups and 16 M reads, 0.74 / 0.98 ms against 0.41 / 0.43 without)
        const uint32_t cov = rc.fl & 0xFFu, seg = (blockIdx.x % (ro.nseg / ro.ncs)) * ro.ncs + cov;
        uint32_t rank = 0;
        if (rcl == 1) rank = atomicAdd(&seg_n[cov], 1u);
        __syncthreads();
        if (threadIdx.x < ro.ncs) {
          const uint32_t t = seg_n[threadIdx.x];
          if (t) {
            seg_at[threadIdx.x] = atomicAdd(&ro.cnt[((blockIdx.x % (ro.nseg / ro.ncs)) * ro.ncs + threadIdx.x) * C3_CSTRIDE], t);
            seg_n[threadIdx.x] = 0;
          }
        }
        __syncthreads();
        if (rcl == 1) rec_store(ro.recs, (uint64_t)ro.seg_base[seg] + seg_at[cov] + rank, rc);
      }
      const uint32_t at2 = wave_append(rcl == 2, &ro.cnt[ro.nseg * C3_CSTRIDE]);
      if (rcl == 2) rec_store(ro.recs, ro.other_at + at2, rc);
    }
    const int lane = threadIdx.x & 63;
    const unsigned long long mask = __ballot(defer);
    if (mask) {  // one LDS atomic per wave
      const int leader = __ffsll((long long)mask) - 1;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&lcount, (uint32_t)__popcll(mask));
      base = __shfl(base, leader, 64);
      if (defer) lq[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)i;
    }
    const unsigned long long pmask = __ballot(to_plain);
    if (pmask) {
      const int leader = __ffsll((long long)pmask) - 1;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&pcount, (uint32_t)__popcll(pmask));
      base = __shfl(base, leader, 64);
      if (to_plain) lq[PF_TILES * 256 - 1 - (base + (uint32_t)__popcll(pmask & ((1ull << lane) - 1ull)))] = (uint32_t)i;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    gbase = lcount ? atomicAdd(queue_n, lcount) : 0u;
    pbase = pcount ? atomicAdd(queue_n + 1, pcount) : 0u;
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < lcount; k += 256) queue[gbase + k] = lq[k];
  for (uint32_t k = threadIdx.x; k < pcount; k += 256) plist[pbase + k] = lq[PF_TILES * 256 - 1 - k];
}

// Second pass: the reads of match / insertion / deletion operations, one thread per listed read (every lane of a wave does the same
// kind of work); the list's length stays on the device.  A read this pass cannot finish either (adaptor geometry) joins the general
// kernel's queue: one global atomic per workgroup and trip.
__global__ __launch_bounds__(256) void k_bqsr_prologue_plain(BqCols m, BqDesc *__restrict__ desc, uint32_t *skipbits, const uint32_t *__restrict__ plist,
                                                             uint32_t *__restrict__ queue, uint32_t *queue_n, uint32_t *err, RecOut ro, const uint4 *__restrict__ plain_rec) {
  __shared__ uint32_t s_cig[256][5];  // the thread's CIGAR: build_pieces / get_read_coord walk it several times
  __shared__ uint32_t wg_n, wg_base, wr_n, wr_base;
  __shared__ PfLds L;
  const bool ref_lds = m.n_ref <= REF_LDS;
  pf_lds_fill(L, m, 256);
  __syncthreads();
  const uint64_t np = (uint64_t)queue_n[1];
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (uint64_t t0 = (uint64_t)blockIdx.x * 256; t0 < np; t0 += stride) {  // (uniform trip count per workgroup: barriers inside)
    const uint64_t t = t0 + threadIdx.x;
    bool defer = false, to_plain = false;
    uint32_t i = 0;
    BqRec rc;
    int rcl = 0;
    if (t < np) {
      i = plist[t];
      // the read's columns as the first pass left them (pf_record<false>): one 64-byte line
      const uint4 *pr = plain_rec + 4 * (size_t)i;
      const uint4 w0 = pr[0], w1 = pr[1], w2 = pr[2], w3 = pr[3];
      const uint32_t ops[5] = {w0.x, w0.y, w0.z, w0.w, w1.x};
      PfCols cur;
      cur.has_sr = 0; cur.mq = 1;  // (the first pass made recalibrateAln's tests)
      cur.f = (uint16_t)w1.y; cur.rg = (uint16_t)(w1.y >> 16);
      cur.r = (int32_t)w1.z; cur.p = (int32_t)w1.w;
      cur.ls = w2.x & 0xFFFFu;
      cur.nrefid = (w2.x >> 19) & 1u ? -1 : 0;  // (only its sign is looked at)
      cur.c0 = (uint64_t)w2.y; cur.c1 = cur.c0 + ((w2.x >> 16) & 7u);
      cur.q0 = (uint64_t)w2.z | ((uint64_t)w2.w << 32); cur.q1 = cur.q0 + cur.ls;
      cur.qb = (uint64_t)w3.x | ((uint64_t)w3.y << 32);
      cur.pnext = (int32_t)w3.z; cur.tlen = (int32_t)w3.w;
      pf_record<true>(m, i, cur, L, ref_lds, desc, skipbits, err, ro.recs != nullptr, s_cig[threadIdx.x], defer, to_plain, rc, rcl, nullptr, ops);
    }
    // deferred reads -> the general kernel's queue, finished records (all class 2 here... or 1 if the CIGAR folded to one run) -> the other
    // region: one global atomic each per workgroup and trip
    if (threadIdx.x == 0) { wg_n = 0; wr_n = 0; }
    __syncthreads();
    uint32_t my = 0, myr = 0;
    if (defer) my = atomicAdd(&wg_n, 1u);
    if (rcl) myr = atomicAdd(&wr_n, 1u);
    __syncthreads();
    if (threadIdx.x == 0) {
      wg_base = wg_n ? atomicAdd(queue_n, wg_n) : 0u;
      wr_base = wr_n ? atomicAdd(&ro.cnt[ro.nseg * C3_CSTRIDE], wr_n) : 0u;
    }
    __syncthreads();
    if (defer) queue[wg_base + my] = i;
    if (rcl) rec_store(ro.recs, ro.other_at + wr_base + myr, rc);
  }
}

// General prologue: one thread per record of `queue` (the records k_bqsr_prologue_fast left: anything but a plain "<len>M" CIGAR
// without adaptor read-through); literal transliteration of the reference's clipping code.
__device__ inline void prologue_general(const BqCols &m, const uint64_t i, uint32_t *__restrict__ cig_scratch, BqDesc *__restrict__ desc, uint32_t *skipbits,
                                        uint32_t *err, const bool recs, BqRec &rc_out, int &rc_class) {
  int32_t rec_pos = 0;  // POS of the clipped copy (set before the final put)
  // stores the descriptor, or - recs != nullptr - the record count3.hip works from (and the descriptor only if the record cannot
  // describe the read)
  auto put = [&](const BqDesc &dd, const uint32_t *cg, int ncg) {
    if (!recs) { desc[i] = dd; return; }
    BqRec rc;
    rc.ref_lo = rc.ref_hi = rc.win = rc.ctxw = 0; rc.t0 = 0; rc.fl = rc.bpk = rc.dpk = 0;
    if (dd.fl & BQ_ELIGIBLE) {
      Pieces4 P;
      pieces4(cg, ncg, rec_pos, P);
      rc = make_rec((int)dd.a, (int)dd.len, (int)dd.left, dd.right == 0xFFFFu ? -1 : (int)dd.right, dd.cov, (dd.fl & BQ_REVERSED) != 0, (dd.fl & BQ_LAST) != 0, P,
                    P.np < 0, m.ref_seq[dd.refid], m.ref_seq_len[dd.refid], (int64_t)m.l_seq[i]);
      rc.fl |= RC_SKIPCOL;  // this kernel's reads have their known-site bits in the skip column
    }
    if (rc.fl & RC_GENERAL) desc[i] = dd;
    rc_class = rec_class(rc) ? 2 : 0;  // (this kernel's reads read the skip column)
    rec_pack_idx(rc, (uint32_t)i);
    rc_out = rc;
  };
  BqDesc d;
  d.D0 = d.D1 = d.D2 = BQ_NOREF; d.refid = 0; d.b1 = d.b2 = 0xFFFF; d.a = 0; d.len = 0; d.left = 0; d.right = 0; d.cov = 0; d.fl = 0; d.pad = 0;
  if (!recalibrate_aln(m, i)) { { put(d, nullptr, 0); return; } }
  if (m.l_seq[i] > (uint32_t)MAX_DESC_READ) { atomicOr(&err[0], 2u); { put(d, nullptr, 0); return; } }
  RAln a;
  a.pos = m.pos[i]; a.pnext = m.pnext[i]; a.tlen = m.tlen[i]; a.refid = m.refid[i]; a.next_refid = m.next_refid[i];
  a.flag = m.flag[i];
  a.cig = m.cigar + m.cigar_off[i];
  a.ncig = (int)(m.cigar_off[i + 1] - m.cigar_off[i]);
  a.off = 0; a.len = (int)m.l_seq[i];
  uint32_t *sc = cig_scratch + 2 * (m.cigar_off[i] + 4 * i);
  a.buf[0] = sc; a.buf[1] = sc + (a.ncig + 4);
  a.cur = -1;
  if (!hard_clip_adaptor(a)) { atomicOr(&err[0], 4u); { put(d, nullptr, 0); return; } }
  if (a.len == 0) { { put(d, nullptr, 0); return; } }
  hard_clip_soft_clipped(a);
  if (a.len == 0) { { put(d, nullptr, 0); return; } }

  // calculateSkipSlice, bqsr.go:389-414: bits live at (qual_off[i] + original base index)
  {
    const int ss = soft_start(a), se = soft_end(a);
    const int32_t *sv = m.sites[a.refid];
    const int64_t ns = m.n_sites[a.refid];
    // intervals.Intersect (intervals/intervals.go:166-173): sites with End >= softStart and Start <= softEnd.  The bucket index
    // replaces the two binary searches (28 dependent loads) by one look-up and a short walk.
    int64_t first = ns, last = ns;
    if (ns > 0) {
      const int64_t nbuck = ((int64_t)m.ref_len[a.refid] >> 6) + 1;
      int64_t bk = (int64_t)(ss < 0 ? 0 : ss) >> 6;
      bk = bk >= nbuck ? nbuck - 1 : bk;
      first = m.site_idx[a.refid][bk];
      while (first < ns && sv[2 * first + 1] < ss) first++;
      last = first;
      while (last < ns && sv[2 * last] <= se) last++;
    }
    const uint64_t bit0 = m.qual_off[i] + (uint64_t)a.off;
    if (recs) clear_skip_bits(skipbits, m.qual_off[i], m.l_seq[i]);  // (this kernel's reads all read the column, RC_SKIPCOL)
    for (int64_t s = first; s < last; s++) {
      bool ok;
      int fs = get_read_coord(a.cig, a.ncig, ss, sv[2 * s], false, &ok);
      if (!ok || fs < 0) fs = 0;
      int fe = get_read_coord(a.cig, a.ncig, ss, sv[2 * s + 1], false, &ok);
      if (!ok || fe > a.len - 1) fe = a.len - 1;
      set_skip_bits(skipbits, bit0, fs, fe);
    }
  }
  ReadView v{m.seq4 + m.seq_off[i], m.qual + m.qual_off[i], a.off, a.len, (bool)(a.flag & F_REVERSED), 0, -1};
  low_quality_bounds(v);
  d.refid = a.refid;
  d.a = (uint16_t)a.off; d.len = (uint16_t)a.len;
  d.left = (uint16_t)v.left; d.right = (uint16_t)(v.right < 0 ? 0xFFFF : v.right);
  d.cov = (uint8_t)m.rg_cov[m.rgid[i]];
  d.fl = BQ_ELIGIBLE | ((a.flag & F_REVERSED) ? BQ_REVERSED : 0) | ((a.flag & F_LAST) ? BQ_LAST : 0);
  rec_pos = a.pos;
  if (!build_pieces(a.cig, a.ncig, a.pos, d)) {
    d.fl |= BQ_COMPLEX;
    if (a.cur < 0) { d.D0 = (int32_t)m.cigar_off[i]; }
    else { d.D0 = (int32_t)(a.buf[a.cur] - cig_scratch); d.fl |= BQ_CIG_SCRATCH; }
    d.b1 = (uint16_t)a.ncig;
    d.D2 = a.pos - 1;
    if (a.ncig > 0xFFFF) atomicOr(&err[0], 2u);
  }
  put(d, a.cig, a.ncig);
}

// the queue's length stays on the device (no read-back between the two prologue kernels): a fixed grid strides over it
__global__ __launch_bounds__(256) void k_bqsr_prologue(BqCols m, const uint32_t *__restrict__ queue, const uint32_t *__restrict__ queue_n,
                                                       uint32_t *__restrict__ cig_scratch, BqDesc *__restrict__ desc, uint32_t *skipbits,
                                                       uint32_t *err, RecOut ro) {
  __shared__ uint32_t wr_n, wr_base;
  const uint64_t nq = (uint64_t)*queue_n;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t t0 = (uint64_t)blockIdx.x * blockDim.x; t0 < nq; t0 += stride) {  // (uniform trip count per workgroup: barriers inside)
    const uint64_t t = t0 + threadIdx.x;
    BqRec rc;
    int rcl = 0;
    if (t < nq) prologue_general(m, queue[t], cig_scratch, desc, skipbits, err, ro.recs != nullptr, rc, rcl);
    if (!ro.recs) continue;
    if (threadIdx.x == 0) wr_n = 0;
    __syncthreads();
    uint32_t myr = 0;
    if (rcl) myr = atomicAdd(&wr_n, 1u);
    __syncthreads();
    if (threadIdx.x == 0) wr_base = wr_n ? atomicAdd(&ro.cnt[ro.nseg * C3_CSTRIDE], wr_n) : 0u;
    __syncthreads();
    if (rcl) rec_store(ro.recs, ro.other_at + wr_base + myr, rc);
  }
}

// reference contigs are kept as 4-bit code nibbles like the restaged SEQ column (ctx.hip k_recode_seq), first base in the LOW
// nibble: A 0, C 1, G 2, T 3, anything else 8 (baseToIntMap, bqsr.go:247-252: a/A/'*' -> A ...; a read base is only ever compared
// when it is A, C, G or T, so "other" needs no finer code).  Comparing 16 read bases with 16 reference bases is one 64-bit XOR.
__global__ __launch_bounds__(256) void k_pack_reference(const uint8_t *__restrict__ ascii, int64_t len, uint8_t *__restrict__ packed, int64_t packed_bytes) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= packed_bytes) return;
  uint32_t out = 0;
  for (int h = 0; h < 2; h++) {
    const int64_t j = 2 * i + h;
    uint32_t code = 8;  // past the contig: "other"
    if (j < len) {
      switch (ascii[j]) {
        case 'a': case 'A': case '*': code = 0; break;
        case 'c': case 'C': code = 1; break;
        case 'g': case 'G': code = 2; break;
        case 't': case 'T': code = 3; break;
        default: code = 8;
      }
    }
    out |= code << (4 * h);
  }
  packed[i] = (uint8_t)out;
}
// idx[b] = first site whose End is >= 64 b (sites are sorted and flattened, so Ends increase with the index)
// Known sites inside the packed reference: bit 2 of the nibble of every base that lies in a site interval (the base codes use bits 0, 1
// and 3; nib_code_differs ignores bit 2).  A read whose clipped copy is ONE run of matches then has its known-site bits in the
// reference window the count kernel loads anyway (count3.hip): no walk over the site list and no skip-column bits for it.
__global__ __launch_bounds__(256) void k_ref_clear_site_flags(uint32_t *__restrict__ packed_words, int64_t n_words) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_words) packed_words[i] &= 0xBBBBBBBBu;
}
__global__ __launch_bounds__(256) void k_ref_mark_sites(const int32_t *__restrict__ sv, int64_t ns, int64_t cap_bases, uint32_t *__restrict__ packed_words) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= ns) return;
  int64_t lo = (int64_t)sv[2 * s] - 1, hi = (int64_t)sv[2 * s + 1] - 1;  // 0-based, inclusive
  lo = lo < 0 ? 0 : lo;
  hi = hi >= cap_bases ? cap_bases - 1 : hi;  // (bases behind the contig's end are "other" codes inside the allocation: flags there are harmless and exact)
  for (int64_t j = lo; j <= hi;) {  // word by word (8 bases)
    const int64_t w = j >> 3;
    int64_t last = (w << 3) + 7;
    last = last > hi ? hi : last;
    const int a = (int)(j & 7), b = (int)(last & 7);
    const uint32_t m = (0x44444444u >> (4 * (7 - b))) & (0x44444444u << (4 * a));
    atomicOr(&packed_words[w], m);
    j = last + 1;
  }
}

__global__ __launch_bounds__(256) void k_site_index(const int32_t *__restrict__ sv, int64_t ns, int64_t nbuck, uint32_t *__restrict__ idx) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbuck) return;
  const int64_t x = b << 6;
  int64_t lo = 0, hi = ns;
  while (lo < hi) { const int64_t md = lo + (hi - lo) / 2; if (sv[2 * md + 1] < x) lo = md + 1; else hi = md; }
  idx[b] = (uint32_t)lo;
}

struct CountArgs {
  uint64_t n, qual_bytes;
  const uint64_t *qual_off, *seq_off;
  const uint8_t *qual, *seq4;
  const BqDesc *desc;
  const uint32_t *cigar, *cig_scratch;
  const uint8_t *skipbits;  // the skip-bit column: bit (QUAL offset of the base)
  uint8_t *const *ref_seq;  // packed (k_pack_reference)
  const int64_t *ref_seq_len;
  int n_ref, n_cov, n_q, lmax, rs, max_cycle;  // rs = row stride of the private table (u32 words)
  unsigned long long *cycle_tbl, *ctx_tbl;  // dense int64 tables of the C ABI (device copies)
  uint32_t *err;
  const uint32_t *tile_first;
  // this pass counts the reads of covariates [cov0, cov0 + n_cov) only (n_cov above = the covariates of the PASS): with many read groups
  // the rows of all covariates do not fit one workgroup's LDS, the host then runs one pass per covariate subset - the reference's maps
  // just grow (filters/bqsr.go:467-551)
  int cov0;
};

// private table of one workgroup: per covariate n_q + CT_XROWS rows of rs words - [0, CT_CYC) sixteen context cells of 32 | 32
// bits (observations | mismatches), then the cycle cells of 16 | 16 bits at word CT_CYC + ((17 * (cycle + lmax)) >> 4) (the 17/16
// stretch keeps the blocks of one read, sixteen cycles apart, out of each other's LDS banks); CT_PAD words behind the last row
// take the zero-adds of bases outside the read.  16-bit cycle counters are safe because a read touches a cycle cell at most once and
// the table is flushed (atomic adds into the dense int64 tables in HBM) before 2^16 reads have passed (CountBody::tile_end).
constexpr int CT_CYC = 32, CT_XROWS = 3, CT_PAD = 64;

// MG ("mismatches global"): the cycle cells hold observations only, 16 bits each, two per word, and the (rare) mismatches of the
// cycle table go straight to the dense table in HBM with one global atomic each - the private table shrinks from 4 to 2 bytes per
// (quality, cycle), so that ~40 qualities x 4 read groups x 150-base reads fit ONE workgroup's LDS in ONE pass.
template <bool CHECK_CYCLE, bool REFLDS, int NTV = FL_THREADS, bool MG = false>
struct CountBody {
  // groups of 256 reads: the per-read LDS (44 B) competes with the private tables for the 80 KB that let two workgroups share a CU
  // (with four read groups and six qualities the tables take 50 KB); 256 KiB steps save the per-step restart of the pipeline.
  // NTV = 1024: one workgroup per CU shares one big table (many qualities x read groups): same waves per SIMD as two of 512
  static constexpr int NT = NTV, TILES = 8, RMAX = NTV == 1024 ? 512 : 256;
  // kernel arguments (scalar copies: a reference to the argument struct would keep this object in scratch memory)
  const uint64_t *__restrict__ seq_off;
  const uint8_t *__restrict__ qual;
  const uint8_t *__restrict__ seq4;
  const uint4 *__restrict__ desc;
  const uint32_t *__restrict__ cigar;
  const uint32_t *__restrict__ cig_scratch;
  const uint8_t *__restrict__ skipbits;
  uint8_t *const *__restrict__ ref_seq;
  const int64_t *__restrict__ ref_seq_len;
  unsigned long long *cycle_tbl, *ctx_tbl;
  int n_cov, n_q, lmax, rs, max_cycle;
  uint32_t cov0;           // first covariate of this pass (n_cov = covariates of the pass)
  // LDS
  const uint64_t *s_refp;  // [REF_LDS] packed-contig pointers and lengths (REFLDS: n_ref <= REF_LDS; else they are read from HBM)
  const int64_t *s_refl;
  uint64_t *s_rp;          // [RMAX] per read of the group: its contig's packed bases and length (resolved once per read at
  int32_t *s_rl;           //           stage time, so that a block's loads depend on ONE LDS round trip after the read is known)
  uint4 *s_desc;
  uint32_t *s_seq;
  const uint32_t *qrow;    // [256] quality -> LDS byte address of its row in covariate 0
  const uint8_t *slot_q;
  uint32_t *tbl;
  uint32_t real_end;       // LDS byte address behind the last real row of covariate 0
  uint32_t rpc_bytes;      // bytes of one covariate's rows
  uint64_t seq_base;
  uint32_t err;
  uint32_t reads_since_flush;
  uint64_t bases_since_flush;

  __device__ __forceinline__ void ref_of(int32_t refid, const uint8_t *__restrict__ &rp, int64_t &rlen) const {
    if (REFLDS) { rp = (const uint8_t *)(const __attribute__((address_space(1))) uint8_t *)s_refp[refid]; rlen = s_refl[refid]; }
    else { rp = ref_seq[refid]; rlen = ref_seq_len[refid]; }
  }
  __device__ __forceinline__ void stage(uint32_t g0, uint32_t ng) {
    const uint4 *src = desc + 2 * (size_t)g0;
    for (uint32_t k = threadIdx.x; k < 2 * ng; k += NT) s_desc[k] = src[k];
    seq_base = seq_off[g0];
    for (uint32_t k = threadIdx.x; k < ng; k += NT) {
      s_seq[k] = (uint32_t)(seq_off[g0 + k] - seq_base);
      const uint4 dx = src[2 * k], dy = src[2 * k + 1];
      const uint8_t *rp = nullptr;
      int64_t rlen = 0;
      if ((dy.w >> 8) & BQ_ELIGIBLE) ref_of((int32_t)dx.w, rp, rlen);
      s_rp[k] = reinterpret_cast<uint64_t>(rp);
      s_rl[k] = (int32_t)rlen;
    }
  }
  __device__ __forceinline__ void ref_of_read(uint32_t rl, const uint8_t *__restrict__ &rp, int64_t &rlen) const {
    rp = (const uint8_t *)(const __attribute__((address_space(1))) uint8_t *)s_rp[rl];
    rlen = s_rl[rl];
  }

  // One base, branch-free and without a select: a base that is not counted adds ZERO to whatever cell its quality and cycle
  // point at (a cell of the table, of the row pad behind it, or of the static arrays in front of it - harmless everywhere), so
  // the sixteen bases of a block are straight-line code of ~11 VALU instructions each.  Qualities that are not counted at all
  // (< 6, or not in this pass) have a row of their own that the flush throws away; qualities > 93 and qualities without a table
  // slot count into two more rows behind the real ones (they become error bits at flush time).
  // z: (counted | mismatch << 16) of four bases, 4 bits apart; fv / ev: counted-with-context / its mismatch, 4 bits apart
  template <int I>
  __device__ __forceinline__ void base(uint32_t z, uint32_t fv, uint32_t ev, uint32_t cw, uint32_t ro, int t, uint32_t rowb, int cyc) {
    constexpr int sh = 4 * (I & 7), zs = 4 * (I & 3);
    uint32_t v1 = (z >> zs) & 0x10001u;
    uint32_t lo = bfe_u32<sh, 1>(fv), hi = bfe_u32<sh, 1>(ev);
    if (CHECK_CYCLE) {  // checkCycleCovariate, bqsr.go:364-369
      const bool out = v1 != 0 && ro < real_end && (cyc > max_cycle || cyc < -max_cycle);
      err |= out ? 16u : 0u;
      v1 = out ? 0u : v1; lo = out ? 0u : lo; hi = out ? 0u : hi;
    }
    const uint32_t row = ro + rowb;
    if (MG) lds_add_u32(lshl_add_u32<2>((uint32_t)(t >> 5), row), (v1 & 1u) << (t & 16));  // cell (t >> 4): word cell / 2, half cell & 1
    else lds_add_u32(lshl_add_u32<2>((uint32_t)(t >> 4), row), v1);
    lds_add_u64(lshl_add_u32<3>(bfe_u32<sh, 4>(cw), row), lo, hi);
  }
  // MG: the mismatches of the block's counted bases -> cycle table in HBM.  E: flag nibbles (bit 4b = base b counted and mismatching)
  __device__ __forceinline__ void mismatches_global(uint64_t E, const Chunk &ch, uint32_t cov, int cyc0, int ci) {
    const uint64_t qlo = (uint64_t)ch.w0 | ((uint64_t)ch.w1 << 32), qhi = (uint64_t)ch.w2 | ((uint64_t)ch.w3 << 32);
    const int ncyc_g = 2 * max_cycle + 1;
    while (E) {
      const int b = __builtin_ctzll(E) >> 2;
      E &= E - 1;
      const uint32_t q = (uint32_t)(((b & 8) ? qhi : qlo) >> (8 * (b & 7))) & 0xFFu;
      const int cyc = cyc0 + b * ci;
      if (qrow[q] < real_end && cyc >= -max_cycle && cyc <= max_cycle)  // a real row of this pass (not "not counted" / bad / missing)
        atomicAdd(cycle_tbl + (((size_t)cov * ELP_NQUAL + q) * ncyc_g + (size_t)(cyc + max_cycle)) * 2 + 1, 1ull);
    }
  }

  struct Pre {
    Chunk ch;            // QUAL bytes
    uint32_t skipw;      // 32 skip bits starting at bit (qpos & ~7)
    uint64_t v0, v1;     // SEQ window
    uint64_t r0, r1;     // reference window of the first piece
    int rsn;
    uint32_t rl, qlow;
    int k0, nb;
  };
  // every global load of the block is issued here: QUAL, skip bits, SEQ window, reference window
  __device__ __forceinline__ bool prefetch(uint32_t rl, int k0, int nb, uint64_t qpos, uint32_t, Pre &p) {
    const uint4 dy = s_desc[2 * rl + 1];
    const uint32_t fl = (dy.w >> 8) & 0xFFu;
    if (!(fl & BQ_ELIGIBLE)) return false;
    if ((dy.w & 0xFFu) - cov0 >= (uint32_t)n_cov) return false;  // a covariate of another pass
    const int a = (int)(dy.y & 0xFFFFu), len = (int)(dy.y >> 16);
    const int cbase = k0 - a;   // clipped base index of block bit 0
    if ((cbase < 0 ? -cbase : 0) >= (len - cbase < nb ? len - cbase : nb)) return false;  // no clipped base in the block
    const uint4 dx = s_desc[2 * rl];
    p.rl = rl; p.k0 = k0; p.nb = nb; p.qlow = (uint32_t)(qpos & 7);
    p.ch.load(qual + qpos);
    __builtin_memcpy(&p.skipw, skipbits + (qpos >> 3), 4);
    seq_load(seq4 + seq_base + s_seq[rl], k0, p.v0, p.v1);
    const uint8_t *__restrict__ rp;
    int64_t rlen;
    ref_of_read(rl, rp, rlen);
    const int32_t D0 = (int32_t)dx.x;
    p.rsn = ref_load(rp, rlen, ((fl & BQ_COMPLEX) || D0 == BQ_NOREF) ? (int64_t)0 : (int64_t)D0 + cbase, p.r0, p.r1);
    return true;
  }

  __device__ __forceinline__ void process(Pre &p) {
    const uint32_t rl = p.rl;
    const int k0 = p.k0, nb = p.nb;
    const uint4 dy = s_desc[2 * rl + 1];
    const uint4 dx = s_desc[2 * rl];
    const uint32_t fl = (dy.w >> 8) & 0xFFu;
    const int a = (int)(dy.y & 0xFFFFu), len = (int)(dy.y >> 16);
    const int cbase = k0 - a;
    int blo = -cbase, bhi = len - cbase;
    blo = blo > 0 ? blo : 0;
    bhi = bhi < nb ? bhi : nb;
    const int32_t D0 = (int32_t)dx.x, D1 = (int32_t)dx.y, D2 = (int32_t)dx.z;
    const int b1 = (int)(dy.x & 0xFFFFu), b2 = (int)(dy.x >> 16);
    const int left = (int)(dy.z & 0xFFFFu), right = (dy.z >> 16) == 0xFFFFu ? -1 : (int)(dy.z >> 16);
    const uint32_t cov = dy.w & 0xFFu;
    const bool rev = fl & BQ_REVERSED;
    const bool complex_read = fl & BQ_COMPLEX;
    const Chunk ch = p.ch;
    const uint32_t skipw = p.skipw >> p.qlow;  // known-site skip bits of the block's bases: bit (qpos + b) of the skip column
    uint64_t S, N;
    seq_unpack(p.v0, p.v1, k0, rev, S, N);
    const uint64_t R0 = ref_unpack(p.r0, p.r1, p.rsn);
    const uint64_t inw = nib_range(blo, bhi);
    uint64_t ohS, cS, ohN, cN;
    nib_classify(S, ohS, cS);
    nib_classify(N, ohN, cN);
    const uint64_t F = inw & ohS & ~nib_spread16(skipw);
    if (F == 0) return;
    // context covariate (bqsr.go:87-146): base and its predecessor in sequencing direction inside [left, right]
    const int cl = left + (rev ? 0 : 1), cr = right - (rev ? 1 : 0);
    const uint64_t CV = ohS & ohN & inw & nib_range_clamped(cl - cbase, cr - cbase + 1);
    const uint64_t CX = (cN | (cS << 2)) ^ (rev ? NIBF : 0ull);  // only read where CV is set
    // SNP events (computeSnpEvents, bqsr.go:254-285): read nibble vs reference nibble
    uint64_t X;
    {
      uint64_t R = 0;
      if (!complex_read) {
        const int B1 = b1 - cbase, B2 = b2 - cbase;  // piece boundaries in block bits (0xFFFF - cbase >= 16 when unused)
        {
          const int hi = bhi < B1 ? bhi : B1;
          if (blo < hi) {
            const uint64_t m = nib_fill(nib_range(blo, hi));
            R |= (D0 == BQ_NOREF ? S : R0) & m;
          }
        }
        if (B1 < bhi) {
          const uint8_t *__restrict__ rp;
          int64_t rlen;
          ref_of_read(rl, rp, rlen);
          const int lo = blo > B1 ? blo : B1, hi = bhi < B2 ? bhi : B2;
          if (lo < hi) {
            const uint64_t m = nib_fill(nib_range(lo, hi));
            R |= (D1 == BQ_NOREF ? S : ref_nibbles(rp, rlen, (int64_t)D1 + cbase)) & m;
          }
          if (B2 < bhi) {
            const int lo2 = blo > B2 ? blo : B2;
            if (lo2 < bhi) {
              const uint64_t m = nib_fill(nib_range(lo2, bhi));
              R |= (D2 == BQ_NOREF ? S : ref_nibbles(rp, rlen, (int64_t)D2 + cbase)) & m;
            }
          }
        }
      } else {
        const uint8_t *__restrict__ rp;
        int64_t rlen;
        ref_of_read(rl, rp, rlen);
        const uint32_t *cg = ((fl & BQ_CIG_SCRATCH) ? cig_scratch : cigar) + (uint32_t)D0;
        R = ref_nibbles_complex(cg, b1, (int64_t)D2, cbase, blo, bhi, rp, rlen, S);
      }
      X = nib_code_differs(S ^ R);  // only read where F is set
    }
    // cycle covariate (bqsr.go:376-387) of block bit b: cf + (cbase + b) * ci
    const int rof = (fl & BQ_LAST) ? -1 : 1;
    const int cf = rof + (rev ? (len - 1) * rof : 0), ci = rev ? -rof : rof;
    const int cyc0 = cf + cbase * ci;
    const uint32_t rowb = (cov - cov0) * rpc_bytes;                      // the covariate's rows
    // cycle cell (in words from the row start): (P + b * st) >> 4;  MG: two cells per word, word (P + b * st) >> 5 (CT_CYC doubled in P)
    const int P = (CT_CYC << (MG ? 5 : 4)) + 17 * (cyc0 + lmax), st = 17 * ci;

    const uint64_t E = X & F, FV = F & CV, EV = E & CV;
    const uint32_t f0 = (uint32_t)F, e0 = (uint32_t)E, f1 = (uint32_t)(F >> 32), e1 = (uint32_t)(E >> 32);
    const uint32_t za = (f0 & 0x1111u) | (e0 << 16), zb = (f0 >> 16) | (e0 & 0x11110000u);
    const uint32_t zc = (f1 & 0x1111u) | (e1 << 16), zd = (f1 >> 16) | (e1 & 0x11110000u);
    const uint32_t fv0 = (uint32_t)FV, ev0 = (uint32_t)EV, fv1 = (uint32_t)(FV >> 32), ev1 = (uint32_t)(EV >> 32);
    const uint32_t c0 = (uint32_t)CX, c1 = (uint32_t)(CX >> 32);
    // groups of eight bases between scheduling barriers: enough independent work to cover the LDS latency without letting the
    // scheduler hoist all sixteen address computations at once (register pressure => occupancy)
#define ELP_B(I, Z, FVW, EVW, CW, R) base<I>(Z, FVW, EVW, CW, R, P + (I) * st, rowb, cyc0 + (I) * ci)
    {
      const uint32_t r0 = qrow[ch.get<0>()], r1 = qrow[ch.get<1>()], r2 = qrow[ch.get<2>()], r3 = qrow[ch.get<3>()];
      const uint32_t r4 = qrow[ch.get<4>()], r5 = qrow[ch.get<5>()], r6 = qrow[ch.get<6>()], r7 = qrow[ch.get<7>()];
      ELP_B(0, za, fv0, ev0, c0, r0); ELP_B(1, za, fv0, ev0, c0, r1); ELP_B(2, za, fv0, ev0, c0, r2); ELP_B(3, za, fv0, ev0, c0, r3);
      ELP_B(4, zb, fv0, ev0, c0, r4); ELP_B(5, zb, fv0, ev0, c0, r5); ELP_B(6, zb, fv0, ev0, c0, r6); ELP_B(7, zb, fv0, ev0, c0, r7);
      __builtin_amdgcn_sched_barrier(0);
    }
    {
      const uint32_t r8 = qrow[ch.get<8>()], r9 = qrow[ch.get<9>()], r10 = qrow[ch.get<10>()], r11 = qrow[ch.get<11>()];
      const uint32_t r12 = qrow[ch.get<12>()], r13 = qrow[ch.get<13>()], r14 = qrow[ch.get<14>()], r15 = qrow[ch.get<15>()];
      ELP_B(8, zc, fv1, ev1, c1, r8); ELP_B(9, zc, fv1, ev1, c1, r9); ELP_B(10, zc, fv1, ev1, c1, r10); ELP_B(11, zc, fv1, ev1, c1, r11);
      ELP_B(12, zd, fv1, ev1, c1, r12); ELP_B(13, zd, fv1, ev1, c1, r13); ELP_B(14, zd, fv1, ev1, c1, r14); ELP_B(15, zd, fv1, ev1, c1, r15);
      __builtin_amdgcn_sched_barrier(0);
    }
#undef ELP_B
    if (MG && E) mismatches_global(E, ch, cov, cyc0, ci);
  }
  __device__ __forceinline__ void slots(uint32_t) {}
  __device__ __forceinline__ void retire() {}
  __device__ __forceinline__ void group_end(uint32_t, uint32_t) {}

  // adds the private table into the dense int64 tables (cycle: [cov][94][2*max_cycle+1][2], context: [cov][94][16][2]) and clears it;
  // the extra rows per covariate: bad and missing qualities become error bits, the row of the qualities that are not counted is dropped
  __device__ __forceinline__ void flush() {
    __syncthreads();
    const int rpc = n_q + CT_XROWS, rows = n_cov * rpc;
    const int ncyc_l = 2 * lmax + 1, ncyc_g = 2 * max_cycle + 1;
    if (MG) {
      // words of two observation cells: cell c = 2 w + half holds cycle index x with (17 x) >> 4 == c, i.e. x = c - c / 17
      const int nw = (((17 * (ncyc_l - 1)) >> 4) >> 1) + 1;
      for (int k = threadIdx.x; k < rows * nw; k += NT) {
        const int row = k / nw, w = k % nw;
        uint32_t *cell = &tbl[row * rs + CT_CYC + w];
        const uint32_t v = *cell;
        if (v) {
          *cell = 0;
          const int cov = (int)cov0 + row / rpc, slot = row % rpc;
          if (slot >= n_q) {
            err |= slot == n_q ? 8u : (slot == n_q + 1 ? 128u : 0u);
          } else {
            const int q = slot_q[slot];
#pragma unroll
            for (int half = 0; half < 2; half++) {
              const uint32_t obs = (v >> (16 * half)) & 0xFFFFu;
              const int c = 2 * w + half, cyc = c - c / 17 - lmax;
              if (obs && cyc >= -max_cycle && cyc <= max_cycle)
                atomicAdd(cycle_tbl + (((size_t)cov * ELP_NQUAL + q) * ncyc_g + (size_t)(cyc + max_cycle)) * 2, (unsigned long long)obs);
            }
          }
        }
      }
    } else
    for (int k = threadIdx.x; k < rows * ncyc_l; k += NT) {
      const int row = k / ncyc_l, x = k % ncyc_l;
      uint32_t *cell = &tbl[row * rs + CT_CYC + ((17 * x) >> 4)];
      const uint32_t v = *cell;
      if (v) {
        *cell = 0;
        const int cov = (int)cov0 + row / rpc, slot = row % rpc;
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
    for (int k = threadIdx.x; k < rows * 16; k += NT) {
      const int row = k >> 4, cx = k & 15;
      unsigned long long *cell = reinterpret_cast<unsigned long long *>(&tbl[row * rs + 2 * cx]);
      const unsigned long long v = *cell;
      if (v) {
        *cell = 0;
        const int cov = (int)cov0 + row / rpc, slot = row % rpc;
        if (slot < n_q) {
          const int q = slot_q[slot];
          // cx = prev | cur << 2 is exactly (key >> 4) & 15 of keyFromContext (bqsr.go:64-76)
          unsigned long long *g = ctx_tbl + (((size_t)cov * ELP_NQUAL + q) * ELP_NCTX + (size_t)cx) * 2;
          atomicAdd(g, v & 0xFFFFFFFFull);
          if (v >> 32) atomicAdd(g + 1, v >> 32);
        }
      }
    }
    __syncthreads();
  }
  // a cycle cell (16 | 16 bits) takes at most one count per read, a context cell (32 | 32 bits) at most one per base; a tile
  // starts at most FL_TILE reads and holds at most FL_TILE + FL_MAX_READ bases
  __device__ __forceinline__ void tile_end(uint32_t nreads, uint64_t nbases) {
    reads_since_flush += nreads;
    bases_since_flush += nbases;
    if (reads_since_flush > 30000u || bases_since_flush > (1ull << 31)) { flush(); reads_since_flush = 0; bases_since_flush = 0; }
  }
};

template <bool CHECK_CYCLE, bool REFLDS, int NTV, bool MG = false>
__global__ __launch_bounds__(NTV, 4) void k_bqsr_count(CountArgs A, QMap qm) {
  constexpr int RMAX = CountBody<CHECK_CYCLE, REFLDS, NTV, MG>::RMAX;
  __shared__ FlatLds<RMAX> L;
  __shared__ uint4 s_desc[2 * RMAX];
  __shared__ uint32_t s_seq[RMAX];
  __shared__ uint32_t qrow[256];
  __shared__ uint8_t slot_q[96];
  __shared__ uint64_t s_refp[REF_LDS];
  __shared__ int64_t s_refl[REF_LDS];
  __shared__ uint64_t s_rp[RMAX];
  __shared__ int32_t s_rl[RMAX];
  extern __shared__ __attribute__((aligned(16))) uint32_t tbl[];
  const int n_all = A.n_cov * (A.n_q + CT_XROWS) * A.rs + CT_PAD;
  const uint32_t tbl_at = lds_address(tbl);
  if (REFLDS)
    for (int r = threadIdx.x; r < A.n_ref; r += NTV) { s_refp[r] = reinterpret_cast<uint64_t>(A.ref_seq[r]); s_refl[r] = A.ref_seq_len[r]; }
  for (int k = threadIdx.x; k < n_all; k += NTV) tbl[k] = 0;
  for (int q = threadIdx.x; q < 256; q += NTV) {
    int row;
    if (q < 6) row = A.n_q + 2;                // not counted (bqsr.go:301-305)
    else if (q >= ELP_NQUAL) row = A.n_q;      // bad quality
    else {
      const uint8_t s = qm.slot[q];
      row = s == 255 ? A.n_q + 2 : (s == 254 ? A.n_q + 1 : (int)s);  // counted in another pass / not in the table
      if (s < 254) slot_q[s] = (uint8_t)q;
    }
    qrow[q] = tbl_at + (uint32_t)(row * A.rs) * 4u;
  }
  __syncthreads();
  CountBody<CHECK_CYCLE, REFLDS, NTV, MG> B;
  B.seq_off = A.seq_off; B.qual = A.qual; B.seq4 = A.seq4; B.desc = reinterpret_cast<const uint4 *>(A.desc);
  B.cigar = A.cigar; B.cig_scratch = A.cig_scratch; B.skipbits = A.skipbits; B.ref_seq = A.ref_seq; B.ref_seq_len = A.ref_seq_len;
  B.cycle_tbl = A.cycle_tbl; B.ctx_tbl = A.ctx_tbl;
  B.n_cov = A.n_cov; B.n_q = A.n_q; B.lmax = A.lmax; B.rs = A.rs; B.max_cycle = A.max_cycle; B.cov0 = (uint32_t)A.cov0;
  B.s_desc = s_desc; B.s_seq = s_seq; B.qrow = qrow; B.slot_q = slot_q; B.tbl = tbl;
  B.s_refp = s_refp; B.s_refl = s_refl; B.s_rp = s_rp; B.s_rl = s_rl;
  B.real_end = tbl_at + (uint32_t)(A.n_q * A.rs) * 4u;
  B.rpc_bytes = (uint32_t)((A.n_q + CT_XROWS) * A.rs) * 4u;
  B.err = 0;
  B.reads_since_flush = 0;
  B.bases_since_flush = 0;
  flat_run(A.qual_off, A.n, A.qual_bytes, A.tile_first, L, B);
  B.flush();
  uint32_t my_err = B.err;
  if (__any(my_err != 0)) {
    for (int d = 32; d >= 1; d >>= 1) my_err |= __shfl_xor(my_err, d, 64);
    if ((threadIdx.x & 63) == 0) atomicOr(&A.err[0], my_err);
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
struct ApDesc { uint16_t left, right, len; uint8_t cov; uint8_t fl; };  // fl: BQ_ELIGIBLE recalibrate, BQ_REVERSED, BQ_LAST
static_assert(sizeof(ApDesc) == 8, "ApDesc is staged as one 8-byte word");

struct ApplyArgs {
  uint64_t n, qual_bytes;
  const uint64_t *qual_off, *seq_off;
  uint8_t *qual;
  const uint8_t *seq4;
  // per-read facts the stage step turns into the 8-byte descriptor (there is no prologue kernel and no descriptor column any more)
  const uint16_t *flag, *rgid, *rg_cov;
  const uint32_t *l_seq;
  const uint64_t *qbounds;
  const uint8_t *cov_present;
  const uint32_t *tile_first;
  const uint8_t *lut;  // [n_cov][94][2*max_cycle+1][17]
  int max_cycle;
  uint32_t *err;
  // LDS-resident two-level LUT (MODE 1 / 2): t1 [n_cov][n_qi + 1][2*lmax+1] ids of distinct 17-byte LUT rows (qi = quality - qlo;
  // row n_qi = "not resident"), t2 [n_dict + 1][17] the rows themselves (row n_dict = 0x80 everywhere)
  const uint16_t *t1;
  const uint8_t *t2;
  int n_cov, n_qi, qlo, lmax, n_dict;
};

// ApplyBQSR (bqsr.go:936-1005): every base with quality >= 6 of a record with a known read group is replaced by the LUT value
// of (read group, quality, cycle, context); cycle and context are taken on the full, unclipped read.
//
// MODE 0: one byte gather per base from the dense LUT in HBM / L2.
// MODE 1, 2: two-level LUT in LDS.  The dense LUT is a table of 17-byte rows (one per (read group, quality, cycle); 16 contexts +
// "no context"), and few of them are distinct: estimateHierarchicalBayesianQuality (bqsr.go:901-919) adds the cycle entry's and
// the context entry's integer empirical qualities to a prior that depends on (read group, quality) only, so a row is determined by
// (read group, quality, empirical quality of the cycle entry).  Level 1 maps (read group, quality in [qlo, qhi], cycle) to a row id
// (MODE 1: one byte, at most 255 rows, level 2 rows 32 bytes apart; MODE 2: the row's byte offset in 16 bits), level 2 holds the
// distinct rows.  A few tens of KB instead of the 143 KB of the rows spelled out, so three workgroups share a CU (one before), and
// ~40 distinct qualities x 4 read groups still fit (the spelled-out table did not: HBM gathers).  Qualities below qlo read row qlo
// (they are < 6 and put back by a byte mask), qualities above qhi read the "not resident" row: bit 7 of the result sends them to
// the rolled fix-up loop (dense LUT, or the error for qualities > 93).
template <bool CHECK_CYCLE, int MODE>
struct ApplyBody {
  // 128 KiB steps in groups of up to 512 reads (12 B of LDS per read)
  static constexpr int NT = FL_THREADS, TILES = 4, RMAX = 512;
  static constexpr int ES = MODE == 1 ? 1 : 2;  // bytes per level-1 entry
  const uint64_t *__restrict__ seq_off;
  const uint64_t *__restrict__ qual_off;
  uint8_t *__restrict__ qual;
  const uint8_t *__restrict__ seq4;
  const uint16_t *__restrict__ flag;
  const uint16_t *__restrict__ rgid;
  const uint16_t *__restrict__ rg_cov;
  const uint32_t *__restrict__ l_seq;
  const uint64_t *__restrict__ qbounds;
  const uint8_t *__restrict__ cov_present;
  const uint8_t *__restrict__ lut;
  int max_cycle;
  uint64_t *s_desc;
  uint32_t *s_seq;
  uint32_t t1_at, t2_at;  // LDS byte addresses of the two levels (MODE != 0); t1_at already has qlo's rows subtracted
  int lmax, rows_w;       // rows_w = (n_qi + 1) * (2 * lmax + 1): level-1 entries per read group
  uint32_t w_es;          // (2 * lmax + 1) * ES
  uint32_t qlo, qhi1;     // resident quality range [qlo, qhi1 - 1]; qhi1 reads the "not resident" row
  uint64_t seq_base;
  uint32_t err;
  Chunk out;              // the block processed last: stored by retire()
  uint64_t out_at;
  int out_nb;

  // the read's descriptor {left, right, len, cov, flags} (ApDesc) straight from the columns: which reads ApplyBQSR touches
  // (bqsr.go:947-958) and the low-quality-tail bounds of computeStrandedClippedSeq (:316-332) that adapt_score left per read
  __device__ __forceinline__ void stage(uint32_t g0, uint32_t ng) {
    seq_base = seq_off[g0];
    for (uint32_t k = threadIdx.x; k < ng; k += NT) {
      const uint64_t i = (uint64_t)g0 + k;
      // all column loads first (one memory latency instead of one per test)
      const uint16_t rg = rgid[i], f = flag[i];
      const int len = (int)l_seq[i];
      const uint64_t q0 = qual_off[i], q1 = qual_off[i + 1], qb = qbounds[i], so = seq_off[i];
      uint64_t d = 0;
      if (rg == ELP_NIL16) err |= 32u;                               // readGroupCovariate panics, bqsr.go:38
      else {
        const uint32_t cov = rg_cov[rg];
        if (cov_present[cov]) {                                      // else: read group absent from the tables, read untouched (:953-955)
          if ((uint64_t)len != q1 - q0) err |= 64u;
          else if (len > MAX_DESC_READ) err |= 2u;
          else {
            const uint32_t hi1 = (uint32_t)qb;
            const int left = hi1 ? (int)(qb >> 32) : len, right = hi1 ? (int)hi1 - 1 : len - 1;
            const uint32_t fl = BQ_ELIGIBLE | ((f & F_REVERSED) ? BQ_REVERSED : 0) | ((f & F_LAST) ? BQ_LAST : 0);
            d = (uint64_t)(uint16_t)left | ((uint64_t)(uint16_t)(right < 0 ? 0xFFFF : right) << 16) | ((uint64_t)(uint16_t)len << 32) | ((uint64_t)(cov & 0xFFu) << 48) |
                ((uint64_t)fl << 56);
          }
        }
      }
      s_desc[k] = d;
      s_seq[k] = (uint32_t)(so - seq_base);
    }
  }
  // dense LUT in HBM/L2: one byte gather per base
  template <int I>
  __device__ __forceinline__ uint32_t base(const Chunk &ch, int nb, uint32_t vw, uint32_t cw, uint32_t Q, int st, int cyc0, int ci, uint32_t qstride) {
    constexpr int sh = 4 * (I & 7);
    const uint32_t q = ch.get<I>();
    bool act = I < nb && q >= 6u;
    err |= (act && q >= (uint32_t)ELP_NQUAL) ? 8u : 0u;
    act = act && q < (uint32_t)ELP_NQUAL;
    if (CHECK_CYCLE) {
      const int cyc = cyc0 + I * ci;
      const bool out = act && (cyc > max_cycle || cyc < -max_cycle);
      err |= out ? 16u : 0u;
      act = act && !out;
    }
    const uint32_t cx = ((cw >> sh) & 15u) | ((((~vw) >> sh) & 1u) << 4);  // 16 = no context
    const uint32_t idx = Q + (uint32_t)(I * st) + q * qstride + cx;
    const uint32_t v = lut[act ? idx : 0u];
    return act ? v : q;
  }
  // two-level LUT in LDS: two dependent LDS reads per base, no select.  bpi = level-1 address of (read group, quality 0 + qlo
  // folded in, cycle of base I); cxw: context index (0..15, 16 = none) of four bases, one byte each.  Bases past the read's end
  // are looked up too (their bytes are never stored), qualities < 6 are put back by a byte-mask select over the result words.
  template <int I>
  __device__ __forceinline__ uint32_t base_lds(const Chunk &ch, uint32_t cxw, uint32_t bp, int ci_es) {
    constexpr int bs = 8 * ((I & 7) >> 1);
    uint32_t qc;
    const uint32_t q = ch.get<I>();
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(qc) : "v"(q), "v"(qlo), "v"(qhi1));
    const uint32_t a1 = __umul24(qc, w_es) + (bp + (uint32_t)(I * ci_es));
    uint32_t id;
    if (MODE == 1) id = *reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>((uintptr_t)a1);
    else id = *reinterpret_cast<const __attribute__((address_space(3))) uint16_t *>((uintptr_t)a1);
    const uint32_t cx = t2_at + ((cxw >> bs) & 0xFFu);
    const uint32_t a2 = MODE == 1 ? lshl_add_u32<5>(id, cx) : id + cx;
    return *reinterpret_cast<const __attribute__((address_space(3))) uint8_t *>((uintptr_t)a2);
  }
  // word of result bytes where the original quality is >= 6, original bytes elsewhere (ApplyBQSR leaves qualities < 6 alone)
  __device__ __forceinline__ static uint32_t keep_low(uint32_t orig, uint32_t res) {
    const uint32_t t = ((orig | 0x80808080u) - 0x06060606u) & 0x80808080u;  // bit 7 of a byte: (quality & 127) >= 6
    const uint32_t m = (t - (t >> 7)) | t;
    return (res & m) | (orig & ~m);
  }
  // bases of a block whose lookup hit the 0x80 row (bases past nb may have raised the flag falsely): quality > 93 -> error;
  // quality above the resident range -> dense LUT.  Rolled loop over the original bytes.
  __device__ __forceinline__ void fixup(Chunk &ch, const Chunk &orig, int nb, uint64_t CV, uint64_t CX, uint32_t Q, int st, int cyc0, int ci) {
    uint64_t lo = (uint64_t)ch.w0 | ((uint64_t)ch.w1 << 32), hi = (uint64_t)ch.w2 | ((uint64_t)ch.w3 << 32);
    const uint64_t olo = (uint64_t)orig.w0 | ((uint64_t)orig.w1 << 32), ohi = (uint64_t)orig.w2 | ((uint64_t)orig.w3 << 32);
    const uint32_t qstride = (uint32_t)(2 * max_cycle + 1) * 17u;
#pragma unroll 1
    for (int i = 0; i < nb; i++) {
      const int bs = 8 * (i & 7);
      const uint32_t q = (uint32_t)(((i & 8) ? ohi : olo) >> bs) & 0xFFu;
      if (q < 6u) continue;                                      // put back by keep_low
      uint64_t v = q;
      if (q >= (uint32_t)ELP_NQUAL) err |= 8u;
      else if (q < qhi1) continue;                               // resident: done by the straight-line code
      else {
        if (CHECK_CYCLE) {
          const int cyc = cyc0 + i * ci;
          if (cyc > max_cycle || cyc < -max_cycle) { err |= 16u; continue; }
        }
        const uint32_t cx = ((uint32_t)(CX >> (4 * i)) & 15u) | ((((uint32_t)(~CV >> (4 * i))) & 1u) << 4);
        v = (uint64_t)lut[Q + (uint32_t)(i * st) + q * qstride + cx];
      }
      const uint64_t m = ~(0xFFull << bs);
      lo = (i & 8) ? lo : ((lo & m) | (v << bs));
      hi = (i & 8) ? ((hi & m) | (v << bs)) : hi;
    }
    ch.w0 = (uint32_t)lo; ch.w1 = (uint32_t)(lo >> 32); ch.w2 = (uint32_t)hi; ch.w3 = (uint32_t)(hi >> 32);
  }

  struct Pre {
    Chunk ch;
    uint64_t v0, v1;  // SEQ window
    uint64_t qpos;
    uint32_t rl;
    int k0, nb;
  };
  __device__ __forceinline__ bool prefetch(uint32_t rl, int k0, int nb, uint64_t qpos, uint32_t, Pre &p) {
    const uint32_t fl = (uint32_t)(s_desc[rl] >> 56);
    if (!(fl & BQ_ELIGIBLE)) return false;
    p.rl = rl; p.k0 = k0; p.nb = nb; p.qpos = qpos;
    p.ch.load(qual + qpos);
    seq_load(seq4 + seq_base + s_seq[rl], k0, p.v0, p.v1);
    return true;
  }
  __device__ __forceinline__ void process(Pre &p) {
    const uint32_t rl = p.rl;
    const int k0 = p.k0, nb = p.nb;
    const uint64_t qpos = p.qpos;
    const uint64_t dw = s_desc[rl];
    const uint32_t fl = (uint32_t)(dw >> 56);
    const int left = (int)(dw & 0xFFFFu), right = ((dw >> 16) & 0xFFFFu) == 0xFFFFu ? -1 : (int)((dw >> 16) & 0xFFFFu);
    const int len = (int)((dw >> 32) & 0xFFFFu);
    const uint32_t cov = (uint32_t)(dw >> 48) & 0xFFu;
    const bool rev = fl & BQ_REVERSED;
    Chunk ch = p.ch;
    uint64_t S, N;
    seq_unpack(p.v0, p.v1, k0, rev, S, N);
    uint64_t ohS, cS, ohN, cN;
    nib_classify(S, ohS, cS);
    nib_classify(N, ohN, cN);
    const int cl = left + (rev ? 0 : 1), cr = right - (rev ? 1 : 0);
    int rhi = cr - k0 + 1;
    rhi = rhi < nb ? rhi : nb;
    const uint64_t CV = ohS & ohN & nib_range_clamped(cl - k0, rhi);
    const uint64_t CX = ((cN | (cS << 2)) ^ (rev ? NIBF : 0ull)) & nib_fill(CV);
    const int rof = (fl & BQ_LAST) ? -1 : 1;
    const int cf = rof + (rev ? (len - 1) * rof : 0), ci = rev ? -rof : rof;
    const int cyc0 = cf + k0 * ci, st = 17 * ci;
    const int ncyc = 2 * max_cycle + 1;
    const uint32_t Q = (uint32_t)((int)cov * ELP_NQUAL * ncyc * 17 + (cyc0 + max_cycle) * 17);
    const uint32_t v0 = (uint32_t)CV, v1 = (uint32_t)(CV >> 32), c0 = (uint32_t)CX, c1 = (uint32_t)(CX >> 32);
    uint32_t b0, b1, b2, b3, b4, b5, b6, b7, b8, b9, b10, b11, b12, b13, b14, b15;
    const Chunk orig = ch;
    if (MODE) {
      const uint32_t bp = t1_at + (uint32_t)((int)cov * rows_w + (cyc0 + lmax)) * (uint32_t)ES;
      const int ci_es = ci * ES;
      // context index per base, one byte each: even bases in ce, odd bases in co
      constexpr uint64_t EVN = 0x0F0F0F0F0F0F0F0Full;
      const uint64_t NV = ~CV & NIB1;
      const uint64_t ce = (CX & EVN) | ((NV & (NIB1 & EVN)) << 4), co = ((CX >> 4) & EVN) | (NV & (NIB1 & ~EVN));
      const uint32_t e0 = (uint32_t)ce, e1 = (uint32_t)(ce >> 32), d0 = (uint32_t)co, d1 = (uint32_t)(co >> 32);
      b0 = base_lds<0>(ch, e0, bp, ci_es); b1 = base_lds<1>(ch, d0, bp, ci_es); b2 = base_lds<2>(ch, e0, bp, ci_es); b3 = base_lds<3>(ch, d0, bp, ci_es);
      b4 = base_lds<4>(ch, e0, bp, ci_es); b5 = base_lds<5>(ch, d0, bp, ci_es); b6 = base_lds<6>(ch, e0, bp, ci_es); b7 = base_lds<7>(ch, d0, bp, ci_es);
      b8 = base_lds<8>(ch, e1, bp, ci_es); b9 = base_lds<9>(ch, d1, bp, ci_es); b10 = base_lds<10>(ch, e1, bp, ci_es); b11 = base_lds<11>(ch, d1, bp, ci_es);
      b12 = base_lds<12>(ch, e1, bp, ci_es); b13 = base_lds<13>(ch, d1, bp, ci_es); b14 = base_lds<14>(ch, e1, bp, ci_es); b15 = base_lds<15>(ch, d1, bp, ci_es);
    } else {
      const uint32_t qstride = (uint32_t)ncyc * 17u;
      b0 = base<0>(ch, nb, v0, c0, Q, st, cyc0, ci, qstride); b1 = base<1>(ch, nb, v0, c0, Q, st, cyc0, ci, qstride);
      b2 = base<2>(ch, nb, v0, c0, Q, st, cyc0, ci, qstride); b3 = base<3>(ch, nb, v0, c0, Q, st, cyc0, ci, qstride);
      b4 = base<4>(ch, nb, v0, c0, Q, st, cyc0, ci, qstride); b5 = base<5>(ch, nb, v0, c0, Q, st, cyc0, ci, qstride);
      b6 = base<6>(ch, nb, v0, c0, Q, st, cyc0, ci, qstride); b7 = base<7>(ch, nb, v0, c0, Q, st, cyc0, ci, qstride);
      b8 = base<8>(ch, nb, v1, c1, Q, st, cyc0, ci, qstride); b9 = base<9>(ch, nb, v1, c1, Q, st, cyc0, ci, qstride);
      b10 = base<10>(ch, nb, v1, c1, Q, st, cyc0, ci, qstride); b11 = base<11>(ch, nb, v1, c1, Q, st, cyc0, ci, qstride);
      b12 = base<12>(ch, nb, v1, c1, Q, st, cyc0, ci, qstride); b13 = base<13>(ch, nb, v1, c1, Q, st, cyc0, ci, qstride);
      b14 = base<14>(ch, nb, v1, c1, Q, st, cyc0, ci, qstride); b15 = base<15>(ch, nb, v1, c1, Q, st, cyc0, ci, qstride);
    }
    ch.w0 = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    ch.w1 = b4 | (b5 << 8) | (b6 << 16) | (b7 << 24);
    ch.w2 = b8 | (b9 << 8) | (b10 << 16) | (b11 << 24);
    ch.w3 = b12 | (b13 << 8) | (b14 << 16) | (b15 << 24);
    if (MODE) {
      const uint32_t any = (ch.w0 | ch.w1) | (ch.w2 | ch.w3);
      ch.w0 = keep_low(orig.w0, ch.w0); ch.w1 = keep_low(orig.w1, ch.w1); ch.w2 = keep_low(orig.w2, ch.w2); ch.w3 = keep_low(orig.w3, ch.w3);
      if (any & 0x80808080u) fixup(ch, orig, nb, CV, CX, Q, st, cyc0, ci);
    }
    out = ch; out_at = qpos; out_nb = nb;
  }
  __device__ __forceinline__ void slots(uint32_t) {}
  __device__ __forceinline__ void retire() {
    if (out_nb) out.store(qual + out_at, out_nb);
    out_nb = 0;
  }
  __device__ __forceinline__ void group_end(uint32_t, uint32_t) {}
  __device__ __forceinline__ void tile_end(uint32_t, uint64_t) {}
};

template <bool CHECK_CYCLE, int MODE>
__global__ __launch_bounds__(FL_THREADS, MODE ? 6 : 4) void k_bqsr_apply_flat(ApplyArgs A) {
  typedef ApplyBody<CHECK_CYCLE, MODE> AB;
  constexpr int RMAX = AB::RMAX;
  __shared__ FlatLds<RMAX> L;
  __shared__ uint64_t s_desc[RMAX];
  __shared__ uint32_t s_seq[RMAX];
  extern __shared__ __attribute__((aligned(16))) uint8_t llut[];
  const int w = 2 * A.lmax + 1, n1 = A.n_cov * (A.n_qi + 1) * w;
  const int t1_bytes = (n1 * AB::ES + 15) & ~15;
  if (MODE) {
    // level 1: ids -> one byte (MODE 1) or the row's byte offset (MODE 2)
    for (int k = threadIdx.x; k < n1; k += AB::NT) {
      const uint32_t id = A.t1[k];
      if (MODE == 1) llut[k] = (uint8_t)id;
      else reinterpret_cast<uint16_t *>(llut)[k] = (uint16_t)(id * 17u);
    }
    // level 2: rows 32 (MODE 1) or 17 (MODE 2) bytes apart
    const int n2 = (A.n_dict + 1) * 17;
    for (int k = threadIdx.x; k < n2; k += AB::NT) {
      const int row = k / 17, cx = k - 17 * row;
      llut[t1_bytes + (MODE == 1 ? 32 * row + cx : k)] = A.t2[k];
    }
    __syncthreads();
  }
  AB B;
  B.seq_off = A.seq_off; B.qual_off = A.qual_off; B.qual = A.qual; B.seq4 = A.seq4; B.lut = A.lut;
  B.flag = A.flag; B.rgid = A.rgid; B.rg_cov = A.rg_cov; B.l_seq = A.l_seq; B.qbounds = A.qbounds; B.cov_present = A.cov_present;
  B.max_cycle = A.max_cycle; B.s_desc = s_desc; B.s_seq = s_seq;
  B.lmax = A.lmax; B.rows_w = (A.n_qi + 1) * w; B.w_es = (uint32_t)(w * AB::ES);
  B.qlo = (uint32_t)A.qlo; B.qhi1 = (uint32_t)(A.qlo + A.n_qi);
  B.t1_at = lds_address(llut) - (uint32_t)A.qlo * B.w_es;
  B.t2_at = lds_address(llut) + (uint32_t)t1_bytes;
  B.err = 0;
  B.out_nb = 0; B.out_at = 0; B.out.w0 = B.out.w1 = B.out.w2 = B.out.w3 = 0;
  flat_run(A.qual_off, A.n, A.qual_bytes, A.tile_first, L, B);
  uint32_t my_err = B.err;
  if (__any(my_err != 0)) {
    for (int d = 32; d >= 1; d >>= 1) my_err |= __shfl_xor(my_err, d, 64);
    if ((threadIdx.x & 63) == 0) atomicOr(&A.err[0], my_err);
  }
}

// ---- distinct rows of the dense LUT over (read group, quality in [qlo, qlo + n_qi), cycle in [-lmax, lmax]) ----
// per_cov (round 5, apply3's covariate split): one dictionary PER covariate - rows of different covariates never share an id, the ids
// count from 0 in every covariate (counter[cov]), covariate c's rows lie at t2 + c * LUT_PC_ROWS * 17
constexpr uint32_t LUT_PC_ROWS = 256;
struct LutRows { const uint8_t *lut; int n_cov, qlo, n_qi, lmax, max_cycle, per_cov; };
__device__ __forceinline__ const uint8_t *lut_row(const LutRows &R, int r) {  // r = (cov * n_qi + qi) * w + x
  const int w = 2 * R.lmax + 1, ncyc = 2 * R.max_cycle + 1;
  const int x = r % w, qi = (r / w) % R.n_qi, cov = r / (w * R.n_qi);
  return R.lut + (((size_t)cov * ELP_NQUAL + (size_t)(R.qlo + qi)) * ncyc + (size_t)(x - R.lmax + R.max_cycle)) * 17;
}
__device__ __forceinline__ int lut_row_cov(const LutRows &R, int r) { return r / ((2 * R.lmax + 1) * R.n_qi); }
__device__ __forceinline__ bool row_eq(const uint8_t *a, const uint8_t *b) {
  bool eq = true;
#pragma unroll
  for (int k = 0; k < 17; k++) eq &= a[k] == b[k];
  return eq;
}
// every row finds or becomes the representative of its content in an open-addressing table of row indices
__global__ __launch_bounds__(256) void k_lut_rows_insert(LutRows R, int n_rows, uint32_t *slots, uint32_t mask, uint32_t *__restrict__ row_slot) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_rows) return;
  const uint8_t *mine = lut_row(R, r);
  const int my_cov = R.per_cov ? lut_row_cov(R, r) : 0;
  uint64_t h = 0x9e3779b97f4a7c15ull + (uint64_t)my_cov;
#pragma unroll
  for (int k = 0; k < 17; k++) h = (h ^ mine[k]) * 0x100000001b3ull;
  uint32_t s = (uint32_t)mix64(h) & mask;
  for (;;) {
    uint32_t cur = __hip_atomic_load(&slots[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0xFFFFFFFFu) {
      cur = atomicCAS(&slots[s], 0xFFFFFFFFu, (uint32_t)r);
      if (cur == 0xFFFFFFFFu) break;
    }
    if ((!R.per_cov || lut_row_cov(R, (int)cur) == my_cov) && row_eq(lut_row(R, (int)cur), mine)) break;
    s = (s + 1) & mask;
  }
  row_slot[r] = s;
}
// occupied slots get dense ids; the representative's row becomes row `id` of level 2
__global__ __launch_bounds__(256) void k_lut_rows_number(LutRows R, const uint32_t *__restrict__ slots, uint32_t n_slots, uint32_t *__restrict__ slot_id,
                                                         uint32_t *counter, uint8_t *__restrict__ t2, uint32_t t2_cap) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_slots) return;
  const uint32_t rep = slots[s];
  if (rep == 0xFFFFFFFFu) return;
  const int cov = R.per_cov ? lut_row_cov(R, (int)rep) : 0;
  const uint32_t id = atomicAdd(counter + cov, 1u);
  slot_id[s] = id;
  if (id < t2_cap) {
    const uint8_t *src = lut_row(R, (int)rep);
    uint8_t *dst = t2 + ((size_t)cov * LUT_PC_ROWS + id) * 17;  // (cov = 0 without per_cov)
    for (int k = 0; k < 17; k++) dst[k] = src[k];
  }
}
// level 1 [cov][n_qi + 1][w]: ids; the extra row per read group and (below) the extra level-2 row stand for "not resident"
__global__ __launch_bounds__(256) void k_lut_rows_index(LutRows R, const uint32_t *__restrict__ row_slot, const uint32_t *__restrict__ slot_id,
                                                        const uint32_t *__restrict__ counter, uint16_t *__restrict__ t1, uint8_t *__restrict__ t2,
                                                        uint32_t t2_cap) {
  const int w = 2 * R.lmax + 1, n1 = R.n_cov * (R.n_qi + 1) * w;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < 17 * (R.per_cov ? R.n_cov : 1)) {  // the 0x80 row behind (every covariate's) distinct rows
    const int cv = k / 17;
    const uint32_t nd = counter[cv];
    if (nd < t2_cap) t2[((size_t)cv * LUT_PC_ROWS + nd) * 17 + (k - 17 * cv)] = 0x80;
  }
  if (k >= n1) return;
  const int x = k % w, qi = (k / w) % (R.n_qi + 1), cov = k / (w * (R.n_qi + 1));
  const uint32_t n_dict = counter[R.per_cov ? cov : 0];
  t1[k] = qi == R.n_qi ? (uint16_t)n_dict : (uint16_t)slot_id[row_slot[(cov * R.n_qi + qi) * w + x]];
}

static int bqsr_error(elp_ctx *c, uint32_t e) {
  ELP_HIP(c, hipMemsetAsync(c->err_flag.p, 0, 4, c->stream));
  if (e & 2u) return set_error(c, ELP_ERR_UNSUPPORTED, "BQSR: read longer than %d bases", MAX_DESC_READ);
  if (e & 4u) return set_error(c, ELP_ERR_DATA, "reference coordinate matches a non-existing base in read (reference: log.Panicf, filters/utils.go:253,262)");
  if (e & 8u) return set_error(c, ELP_ERR_DATA, "BQSR: base quality above 93");
  if (e & 16u) return set_error(c, ELP_ERR_DATA, "cycle value exceeds maximum cycle value (reference: log.Panic, filters/bqsr.go:364-369)");
  if (e & 32u) return set_error(c, ELP_ERR_DATA, "BQSR requires input with read groups (reference: log.Panic, filters/bqsr.go:38)");
  if (e & 64u) return set_error(c, ELP_ERR_DATA, "ApplyBQSR: len(QUAL) != len(SEQ) (reference: index out of range panic)");
  if (e & 256u) return set_error(c, ELP_ERR_HIP, "radix sort: tile look-back timed out");
  if (e & 1024u) return set_error(c, ELP_ERR_HIP, "BQSR prologue: a read without known-site bits took a record form that needs them (internal)");
  return set_error(c, ELP_ERR_DATA, "BQSR: device error word %u", e);
}

static int sync_bqsr_ptrs(elp_ctx *c) {
  if (!c->bqsr_ptrs_dirty) return 0;
  const size_t nr = (size_t)c->n_ref;
  // known-site flags of the packed contigs whose reference or site list changed (either order of the two setters)
  for (size_t r = 0; r < nr; r++) {
    if (!c->ref_flags_dirty[r] || !c->h_ref_seq[r]) continue;
    const int64_t len = c->h_ref_seq_len[r], n_words = ((len + 1) / 2 + REF_PAD) / 4;
    uint32_t *pw = reinterpret_cast<uint32_t *>(c->h_ref_seq[r]);
    hipLaunchKernelGGL(k_ref_clear_site_flags, dim3(blocks_for((uint64_t)std::max<int64_t>(n_words, 1), 256)), dim3(256), 0, c->stream, pw, n_words);
    if (c->h_sites[r] && c->h_n_sites[r] > 0)
      hipLaunchKernelGGL(k_ref_mark_sites, dim3(blocks_for((uint64_t)c->h_n_sites[r], 256)), dim3(256), 0, c->stream, (const int32_t *)c->h_sites[r], c->h_n_sites[r], n_words * 8, pw);
    ELP_HIP(c, hipGetLastError());
    c->ref_flags_dirty[r] = 0;
  }
  ELP_TRY(ensure(c, c->d_ref_seq, nr + 1));
  ELP_TRY(ensure(c, c->d_ref_seq_len, nr + 1));
  ELP_TRY(ensure(c, c->d_sites, nr + 1));
  ELP_TRY(ensure(c, c->d_n_sites, nr + 1));
  ELP_TRY(ensure(c, c->d_site_idx, nr + 1));
  if (nr) {
    ELP_HIP(c, hipMemcpyAsync(c->d_ref_seq.p, c->h_ref_seq.data(), nr * sizeof(uint8_t *), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipMemcpyAsync(c->d_ref_seq_len.p, c->h_ref_seq_len.data(), nr * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipMemcpyAsync(c->d_sites.p, c->h_sites.data(), nr * sizeof(int32_t *), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipMemcpyAsync(c->d_n_sites.p, c->h_n_sites.data(), nr * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipMemcpyAsync(c->d_site_idx.p, c->h_site_idx.data(), nr * sizeof(uint32_t *), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, elp::stream_wait(c->stream));
  }
  c->bqsr_ptrs_dirty = false;
  return 0;
}

// builds the three tables in c->dev_tables; qual_tbl != nullptr: also copies them to the host
int tables_written(elp_ctx *c) {
  if (!c->tables_ev) ELP_HIP(c, hipEventCreateWithFlags(&c->tables_ev, hipEventDisableTiming));
  ELP_HIP(c, hipEventRecord(c->tables_ev, c->stream));
  return 0;
}

// The "other" region of the count kernel's records (reads with indels, clipped windows, descriptors; appended by three kernels in
// arrival order) sorted by covariate for the covariate-split count: counts per covariate, offsets, a scatter into a second region.
constexpr int CO_TILE = 1024, CO_MAXCOV = 256;  // (a covariate id is a byte: any number of read groups the context accepts)
__global__ __launch_bounds__(256) void k_c3_other_hist(const uint4 *__restrict__ recs, const uint32_t *__restrict__ n_dev, uint32_t *__restrict__ cnt /* [CO_MAXCOV] */) {
  __shared__ uint32_t h[CO_MAXCOV];
  const uint32_t n = *n_dev;
  if ((uint64_t)blockIdx.x * CO_TILE >= n) return;
  h[threadIdx.x] = 0;
  __syncthreads();
  for (uint32_t k = blockIdx.x * CO_TILE + threadIdx.x; k < n && k < (blockIdx.x + 1u) * CO_TILE; k += 256) atomicAdd(&h[recs[2 * (size_t)k + 1].y & (CO_MAXCOV - 1)], 1u);
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&cnt[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_c3_other_offsets(const uint32_t *__restrict__ cnt, uint32_t *__restrict__ off /* [CO_MAXCOV + 1] */, uint32_t *__restrict__ cursor) {
  uint32_t at = 0;
  for (int c = 0; c < CO_MAXCOV; c++) { off[c] = at; cursor[c] = at; at += cnt[c]; }
  off[CO_MAXCOV] = at;
}
__global__ __launch_bounds__(256) void k_c3_other_scatter(const uint4 *__restrict__ recs, const uint32_t *__restrict__ n_dev, uint32_t *cursor, uint4 *__restrict__ out) {
  __shared__ uint32_t h[CO_MAXCOV], base[CO_MAXCOV];
  const uint32_t n = *n_dev;
  if ((uint64_t)blockIdx.x * CO_TILE >= n) return;
  h[threadIdx.x] = 0;
  __syncthreads();
  uint32_t my[CO_TILE / 256], cv[CO_TILE / 256];
#pragma unroll
  for (int j = 0; j < CO_TILE / 256; j++) {
    const uint32_t k = blockIdx.x * CO_TILE + j * 256 + threadIdx.x;
    cv[j] = k < n ? (recs[2 * (size_t)k + 1].y & (CO_MAXCOV - 1)) : 0u;
    my[j] = k < n ? atomicAdd(&h[cv[j]], 1u) : 0u;
  }
  __syncthreads();
  base[threadIdx.x] = h[threadIdx.x] ? atomicAdd(&cursor[threadIdx.x], h[threadIdx.x]) : 0u;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < CO_TILE / 256; j++) {
    const uint32_t k = blockIdx.x * CO_TILE + j * 256 + threadIdx.x;
    if (k < n) {
      const size_t to = (size_t)base[cv[j]] + my[j];
      out[2 * to] = recs[2 * (size_t)k];
      out[2 * to + 1] = recs[2 * (size_t)k + 1];
    }
  }
}

// Covariate-split class-1 segments (RecOut, round 5): how many records segment (wave % groups) * ncs + covariate can receive at most - the
// reads of that covariate among the records the first prologue pass's workgroups of that group handle - and the
// segments' first slots as the prefix sums of those counts (or, fixed != 0: a fixed stride apart).  Workgroup b covers the records of the
// prologue's workgroup b.
__global__ __launch_bounds__(256) void k_c3_seg_hist(uint64_t n, const uint16_t *__restrict__ rgid, const uint16_t *__restrict__ rg_cov, uint32_t groups, uint32_t ncs,
                                                     uint32_t *__restrict__ seg_cap /* [groups * ncs] */) {
  __shared__ uint32_t h[C3_MAXSEG];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t i0 = (uint64_t)blockIdx.x * PF_TILES * 256 + threadIdx.x;
  for (int tile = 0; tile < PF_TILES; tile++) {
    const uint64_t i = i0 + (uint64_t)tile * 256;
    if (i0 - threadIdx.x + (uint64_t)tile * 256 >= n) break;  // (uniform: the whole tile lies behind the last record)
    const uint16_t rg = i < n ? rgid[i] : (uint16_t)ELP_NIL16;
    const uint32_t cov = rg == ELP_NIL16 ? 0xFFFFu : (uint32_t)(rg_cov[rg] & 0xFFu);
    // (every lane adds its one in the LDS: cheaper than forming the wave's groups by covariate first, apply3.hip k_apply_cov_hist)
    if (cov < ncs) atomicAdd(&h[(blockIdx.x % groups) * ncs + cov], 1u);  // (workgroup b covers the records of the prologue's workgroup b)
  }
  __syncthreads();
  if (threadIdx.x < groups * ncs && h[threadIdx.x]) atomicAdd(&seg_cap[threadIdx.x], h[threadIdx.x]);
}
__global__ void k_c3_seg_offsets(const uint32_t *__restrict__ seg_cap, uint32_t nseg, uint32_t fixed, uint32_t *__restrict__ seg_base /* [nseg + 1] */) {
  uint32_t at = 0;
  for (uint32_t s2 = 0; s2 < nseg; s2++) { seg_base[s2] = at; at += fixed ? fixed : seg_cap[s2]; }
  seg_base[nseg] = at;
}

// ---- the tables' and the LUT's rows form (round 5): with many read groups the dense tables / LUT are tens of megabytes of which only the
// rows of the qualities that occur hold anything; only those rows cross PCIe.
// packs the rows of `quals` (slot k = quality quals[k]) of the three dense tables behind each other: q [n_cov][nq][2] | c [n_cov][nq][ncyc][2]
// | x [n_cov][nq][16][2]; flags a row with observations whose quality is not among them (the caller then fetches the dense tables)
__global__ __launch_bounds__(256) void k_tables_pack_rows(const unsigned long long *__restrict__ tb, int n_cov, int ncyc, const uint8_t *__restrict__ quals, int nq,
                                                          unsigned long long *__restrict__ out, uint32_t *uncovered) {
  const size_t row_w = 2 + (size_t)ncyc * 2 + ELP_NCTX * 2;  // words of one (covariate, quality) row over the three tables
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x, n_rows = (size_t)n_cov * (size_t)nq;
  if (k < (size_t)n_cov * ELP_NQUAL) {  // (the first n_cov * 94 threads also check the qualities that were not asked for)
    const int q = (int)(k % ELP_NQUAL);
    bool asked = false;
    for (int j = 0; j < nq; j++) asked |= quals[j] == q;
    if (!asked && tb[2 * k] != 0) atomicOr(uncovered, 1u);
  }
  if (k >= n_rows * row_w) return;
  const size_t r = k / row_w, w = k - r * row_w;
  const size_t cv = r / (size_t)nq, q = quals[r % (size_t)nq], src_row = cv * ELP_NQUAL + q;
  const size_t nq_all = (size_t)n_cov * ELP_NQUAL * 2, nc_all = nq_all * (size_t)ncyc;
  const size_t oq = 0, oc = n_rows * 2, ox = oc + n_rows * (size_t)ncyc * 2;
  if (w < 2) out[oq + r * 2 + w] = tb[src_row * 2 + w];
  else if (w < 2 + (size_t)ncyc * 2) out[oc + r * (size_t)ncyc * 2 + (w - 2)] = tb[nq_all + src_row * (size_t)ncyc * 2 + (w - 2)];
  else out[ox + r * ELP_NCTX * 2 + (w - 2 - (size_t)ncyc * 2)] = tb[nq_all + nc_all + src_row * ELP_NCTX * 2 + (w - 2 - (size_t)ncyc * 2)];
}
// the dense LUT [n_cov][94][ncyc][17] from its rows form: rows [n_cov][nq][ncyc][17] for the qualities with a slot, the default byte else
__global__ __launch_bounds__(256) void k_lut_expand_rows(const uint8_t *__restrict__ rows, const uint8_t *__restrict__ defaults, const uint8_t *__restrict__ slot_of /* [94], 255 = none */,
                                                        int n_cov, int nq, int ncyc, uint8_t *__restrict__ lut) {
  const size_t row_b = (size_t)ncyc * 17, k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= (size_t)n_cov * ELP_NQUAL * row_b) return;
  const size_t r = k / row_b, w = k - r * row_b, cv = r / ELP_NQUAL, q = r % ELP_NQUAL;
  const uint8_t sl = slot_of[q];
  lut[k] = sl == 255 ? defaults[r] : rows[((cv * (size_t)nq + sl) * row_b) + w];
}

static int gather_impl(elp_ctx *c, int max_cycle, int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl) {
  for (int r = 0; r < c->n_ref; r++)
    if (!c->h_ref_seq[r]) return set_error(c, ELP_ERR_ARG, "elp_bqsr_gather: no reference sequence set for refid %d", r);
  for (int r = 0; r < c->n_ref; r++)
    if (!c->h_sites[r]) ELP_TRY(elp_bqsr_set_known_sites(c, r, nullptr, 0));
  ELP_TRY(sync_bqsr_ptrs(c));
  ELP_TRY(ensure_adapted(c, false));  // low-quality-tail bounds per read (adapt_score)
  ELP_TRY(ensure_flat_index(c));
  ELP_TRY(ensure_qual_present(c));  // sizing hint: the set of quality values seen in a sample of the column
  if (c->n_cov > 255) return set_error(c, ELP_ERR_UNSUPPORTED, "more than 255 read-group covariates");
  if (c->cigar_ops + 4 * c->n >= 0x7FFFFFFFull) return set_error(c, ELP_ERR_UNSUPPORTED, "CIGAR pool exceeds 2^31 operations per context");
  const int ncyc_g = 2 * max_cycle + 1;
  const size_t nq = (size_t)c->n_cov * ELP_NQUAL * 2, nc = nq * ncyc_g, nx = nq * ELP_NCTX;
  c->tables_n = 0;
  ELP_TRY(ensure(c, c->dev_tables, nq + nc + nx + elp_ctx::TABLES_TAIL));
  unsigned long long *tb = c->dev_tables.p;
  hipStream_t st = c->stream;
  ELP_HIP(c, hipMemsetAsync(tb, 0, (nq + nc + nx) * sizeof(unsigned long long), st));
  const uint64_t n = c->n;
  if (n && c->qual_bytes) {
    uint32_t *cs_pool;
    ELP_TRY(scratch(c, 1, 2 * (c->cigar_ops + 4 * n) + 64, &cs_pool));
    BqDesc *desc;
    ELP_TRY(scratch(c, 2, n + 4, &desc));
    uint4 *plain_rec;  // a 64-byte line per staged read, written for the reads with indels only (pf_record<false> -> k_bqsr_prologue_plain)
    ELP_TRY(scratch(c, 0, 4 * n + 8, &plain_rec));
    uint32_t *skipbits;
    const size_t skip_words = (size_t)((c->qual_bytes + 31) / 32 + 8);
    ELP_TRY(scratch(c, 3, skip_words, &skipbits));
    BqCols m{n, c->refid.p, c->pos.p, c->next_refid.p, c->pnext.p, c->tlen.p, c->flag.p, c->rgid.p, c->mapq.p, c->has_sr.p, c->l_seq.p,
             c->cigar_off.p, c->seq_off.p, c->qual_off.p, c->cigar.p, c->seq4.p, c->qual.p, c->ref_len.p, c->rg_cov.p, c->n_ref,
             c->d_ref_seq.p, c->d_ref_seq_len.p, c->d_sites.p, c->d_n_sites.p, c->d_site_idx.p, c->qbounds.p};
    // count3.hip (read sets of one length) works from 32-byte records the prologue kernels write instead of the descriptors; it takes
    // the count if the staged reads have one length (checked once per staged column), no read can exceed --max-cycle, and the quality
    // slots fit one table pass - k_bqsr_count otherwise (elp_set_tuning "count_kernel" = 1 forces it: A/B measurements)
    // How the one-length count kernel takes this read set: 0 not at all (k_bqsr_count), 1 one private table with the rows of
    // every covariate, or - if those do not fit, or only with little replication of the context cells - 2: split by covariate (records
    // in per-covariate segments, a workgroup counts ONE covariate at a time: the table needs n_q + 3 rows whatever the number of read groups)
    ELP_TRY(ensure_uniform_len(c));
    const int lmax0 = (int)std::max<uint32_t>(c->max_l_seq, 1);
    uint32_t ncs = 1;
    while ((int)ncs < c->n_cov) ncs <<= 1;
    const uint32_t nseg2 = std::max<uint32_t>((uint32_t)C3_NSEG, ncs);  // segments of the covariate split (ncs <= 256: n_cov <= 255 above)
    uint32_t *queue;
    ELP_TRY(scratch(c, 5, 2 * n + 128 + (size_t)(C3_MAXSEG + 1) * C3_CSTRIDE + 4 * CO_MAXCOV + 2 * C3_MAXSEG + 32, &queue));  // [0] = count, [4..] = records left to the general kernel; [1] = count, [n + 20..] = reads of the second pass;
    ELP_HIP(c, hipMemsetAsync(queue, 0, 16, st));  // [2 n + 48 ..] = the record counters (RecOut), the other region's sort words, the segments' sizes and first slots
    uint32_t *plist = queue + n + 20, *rec_cnt = queue + ((2 * n + 48 + 63) & ~(uint64_t)63);
    uint32_t *cw = rec_cnt + (size_t)(C3_MAXSEG + 1) * C3_CSTRIDE;  // [CO_MAXCOV] counts | [CO_MAXCOV + 1] offsets | [CO_MAXCOV] cursors
    uint32_t *seg_cap = cw + 3 * CO_MAXCOV + 1, *seg_base = seg_cap + C3_MAXSEG;  // [C3_MAXSEG] | [C3_MAXSEG + 1]
    // a wave of the first pass appends its class-1 records (at most PF_TILES * 64) to segment wave % C3_NSEG
    const unsigned pf_grid = blocks_for(n, 256 * PF_TILES);
    const uint64_t cap_s1 = ((uint64_t)pf_grid * 4 + C3_NSEG - 1) / C3_NSEG * (uint64_t)(PF_TILES * 64);
    auto c3_mode = [&](int nq) -> int {
      if (c->tune.count_kernel == 1 || c->uniform_len == 0 || lmax0 > max_cycle || lmax0 > 1022) return 0;
      int rsw3 = 0, rlog3 = 0;
      size_t dyn3 = 0;
      const bool all_fits = count3_plan(c->n_cov, nq, lmax0, &rsw3, &rlog3, &dyn3, c->tune.count3_rlog) == 0;
      if (all_fits && (rlog3 >= 3 || c->n_cov == 1) && c->tune.count_kernel != 3) return 1;
      if (c->n_cov > 1 && c->tune.count_kernel != 2 && count3_plan(1, nq, lmax0, &rsw3, &rlog3, &dyn3, c->tune.count3_rlog) == 0) return 2;
      return all_fits ? 1 : 0;
    };
    int nq0 = 0;
    for (int q = 6; q < ELP_NQUAL; q++) nq0 += (int)((q < 64 ? (c->qual_present[0] >> q) : (c->qual_present[1] >> (q - 64))) & 1ull);
    int mode = c3_mode(std::max(nq0, 1));
    BqRec *recs = nullptr;
    uint64_t other_at = 0;  // first slot of the other region = the class-1 area's capacity
    uint32_t nseg = C3_NSEG;
    // class-1 segments | the other region | (mode 2) the other region sorted by covariate; the counters and the segments' first slots
    auto rec_buffers = [&]() -> int {
      nseg = mode == 2 ? nseg2 : (uint32_t)C3_NSEG;
      other_at = mode == 2 ? n : (uint64_t)C3_NSEG * cap_s1;
      recs = nullptr;
      if (!mode) return 0;
      ELP_TRY(scratch(c, 4, (size_t)other_at + (mode == 2 ? 2 : 1) * (size_t)n + 64, &recs));
      ELP_HIP(c, hipMemsetAsync(rec_cnt, 0, (size_t)(C3_MAXSEG + 1) * C3_CSTRIDE * sizeof(uint32_t), st));
      if (mode == 2) {
        ELP_HIP(c, hipMemsetAsync(seg_cap, 0, (size_t)C3_MAXSEG * sizeof(uint32_t), st));
        ELP_LAUNCH(c, "bqsr_seg_hist", k_c3_seg_hist, dim3(pf_grid), dim3(256), 0, n, (const uint16_t *)c->rgid.p, (const uint16_t *)c->rg_cov.p, nseg / ncs, ncs, seg_cap);
      }
      ELP_LAUNCH(c, "bqsr_seg_offsets", k_c3_seg_offsets, dim3(1), dim3(1), 0, (const uint32_t *)seg_cap, nseg, mode == 2 ? 0u : (uint32_t)cap_s1, seg_base);
      return 0;
    };
    if ((uint64_t)C3_NSEG * cap_s1 + 2 * n >= 0xFFFFFFF0ull) return set_error(c, ELP_ERR_UNSUPPORTED, "BQSR: more than ~1.4 G records per context");
    ELP_TRY(rec_buffers());
    if (!recs) ELP_HIP(c, hipMemsetAsync(skipbits, 0, skip_words * 4, st));  // (with records the reads that use the column clear their own bits: clear_skip_bits)
    // the three prologue passes; with `r` they leave 32-byte records for count3.hip, without it the descriptors of k_bqsr_count
    auto run_prologues = [&](BqRec *r) -> int {
      const RecOut ro{r, rec_cnt, seg_base, other_at, nseg, (r && mode == 2) ? ncs : 0u};
      ELP_LAUNCH(c, "bqsr_prologue_fast", k_bqsr_prologue_fast, dim3(pf_grid), dim3(256), 0, m, desc, skipbits, queue + 4, queue, c->err_flag.p, ro, plist, plain_rec);
      // (sized for the worst case; workgroups beyond the list's end leave at once)
      ELP_LAUNCH(c, "bqsr_prologue_plain", k_bqsr_prologue_plain, dim3(std::min<unsigned>(blocks_for(n, 256), (unsigned)c->n_cu * 16)), dim3(256), 0, m, desc, skipbits,
                 (const uint32_t *)plist, queue + 4, queue, c->err_flag.p, ro, (const uint4 *)plain_rec);
      ELP_LAUNCH(c, "bqsr_prologue", k_bqsr_prologue, dim3(std::min<unsigned>(blocks_for(n, 256), (unsigned)c->n_cu * 16)), dim3(256), 0, m,
                 (const uint32_t *)(queue + 4), (const uint32_t *)queue, cs_pool, desc, skipbits, c->err_flag.p, ro);
      return 0;
    };
    ELP_TRY(run_prologues(recs));
    const int lmax = (int)std::max<uint32_t>(c->max_l_seq, 1);
    if (lmax > MAX_DESC_READ) return set_error(c, ELP_ERR_UNSUPPORTED, "BQSR: read longer than %d bases", MAX_DESC_READ);
    const bool check_cycle = lmax > max_cycle;
    const int rs = (CT_CYC + ((17 * 2 * lmax) >> 4) + 1 + 1) & ~1;  // words per row
    const size_t per_slot = (size_t)c->n_cov * (size_t)rs * 4;
    typedef CountBody<false, true> CB;
    typedef CountBody<false, true, 1024> CB1;
    const size_t static_lds = sizeof(FlatLds<CB::RMAX>) + (size_t)CB::RMAX * (sizeof(BqDesc) + 4) + 1024 + 96 + 64 + 8 + (size_t)CT_PAD * 4 + (size_t)REF_LDS * 16 + (size_t)CB::RMAX * 12;
    const size_t static_lds1 = sizeof(FlatLds<CB1::RMAX>) + (size_t)CB1::RMAX * (sizeof(BqDesc) + 4) + 1024 + 96 + 64 + 8 + (size_t)CT_PAD * 4 + (size_t)REF_LDS * 16 + (size_t)CB1::RMAX * 12;
    const size_t lds_cu = 160 * 1024;
    const uint64_t nsteps = flat_steps<CB>(c->qual_bytes);
    for (int attempt = 0;; attempt++) {
      // quality values to give table slots (>= 6, <= 93)
      std::vector<int> quals;
      for (int q = 6; q < ELP_NQUAL; q++)
        if ((q < 64 ? (c->qual_present[0] >> q) : (c->qual_present[1] >> (q - 64))) & 1ull) quals.push_back(q);
      if (quals.empty()) quals.push_back(6);
      // as many workgroups per CU (512 threads each) as still hold the rows of every covariate and quality slot in one pass; else ONE
      // workgroup of 1024 threads per CU around one table (as many waves per SIMD as two of 512) and, if that does not hold them either,
      // several passes: over quality subsets and - many read groups - over covariate subsets [cov0, cov0 + ncp) (a pass skips the reads of
      // the other covariates).  The reference's tables are maps that just grow (filters/bqsr.go:467-551): any number of read groups runs.
      int wg_per_cu = 1, qcap = 0, ncp = c->n_cov;
      bool big = false;
      for (int w = 3; w >= 2; w--) {
        const size_t budget = lds_cu / (size_t)w;
        if (budget <= static_lds + 256) continue;
        const int cap = (int)((budget - static_lds - 256) / per_slot) - CT_XROWS;  // minus the extra rows per covariate
        if (cap >= (int)quals.size()) { wg_per_cu = w; qcap = cap; break; }
      }
      bool mg = false;
      int rs_use = rs;
      if (!qcap) {
        big = true;
        // rows of `rsx` words that fit -> the (covariates, quality slots) per pass with the fewest passes; 0 = not even one covariate's four rows fit
        auto plan = [&](int rsx, int *ncp_out, int *qcap_out) -> long {
          const long rows_fit = (long)((lds_cu - static_lds1 - 256) / ((size_t)rsx * 4));
          long best = 0;
          for (int k = c->n_cov; k >= 1; k--) {
            const long qc = std::min<long>((long)quals.size(), rows_fit / k - CT_XROWS);
            if (qc < 1) continue;
            const long passes = (long)((c->n_cov + k - 1) / k) * (long)(((long)quals.size() + qc - 1) / qc);
            if (!best || passes < best) { best = passes; *ncp_out = k; *qcap_out = (int)qc; }
          }
          return best;
        };
        const long passes = plan(rs, &ncp, &qcap);
        // still several passes: observation-only cycle cells (half the bytes; mismatches by global atomics)
        if (passes != 1) {
          const int rs_mg = (CT_CYC + (((17 * 2 * lmax) >> 4) >> 1) + 1 + 1) & ~1;
          int ncp_mg = 0, qcap_mg = 0;
          const long passes_mg = plan(rs_mg, &ncp_mg, &qcap_mg);
          if (passes_mg && (!passes || passes_mg < passes)) { mg = true; rs_use = rs_mg; ncp = ncp_mg; qcap = qcap_mg; }
        }
        if (qcap < 1) return set_error(c, ELP_ERR_UNSUPPORTED, "BQSR: one covariate's private table rows do not fit in LDS (max read length=%d)", lmax);
      }
      const int grid = (int)std::min<uint64_t>(nsteps, (uint64_t)wg_per_cu * (uint64_t)c->n_cu);
      if (recs) {
        // the sampled hint chose a form of the one-length kernel; the exact set (taken after the count met a quality without a slot) may
        // need another one - the covariate split, or the general kernel with its several passes: the reference just runs
        // (filters/bqsr.go:467-551), so does this - the prologues once more, leaving what the new form reads
        const int mode2 = c3_mode((int)quals.size());
        if (mode2 != mode) {
          mode = mode2;
          ELP_TRY(rec_buffers());
          ELP_HIP(c, hipMemsetAsync(skipbits, 0, skip_words * 4, st));
          ELP_HIP(c, hipMemsetAsync(queue, 0, 16, st));
          ELP_TRY(run_prologues(recs));
        }
      }
      if (recs) {
        int rsw3 = 0, rlog3 = 0;
        size_t dyn3 = 0;
        (void)count3_plan(mode == 2 ? 1 : c->n_cov, (int)quals.size(), lmax, &rsw3, &rlog3, &dyn3, c->tune.count3_rlog);
        QMap qm;
        memset(qm.slot, 254, sizeof qm.slot);
        for (size_t s2 = 0; s2 < quals.size(); s2++) qm.slot[quals[s2]] = (uint8_t)s2;
        const uint4 *other = reinterpret_cast<const uint4 *>(recs) + 2 * (size_t)other_at;  // the other region (32-byte records)
        const uint32_t *n_other = rec_cnt + (size_t)nseg * C3_CSTRIDE;
        ELP_HIP(c, hipMemsetAsync(cw, 0, (3 * CO_MAXCOV + 1) * sizeof(uint32_t), st));
        uint32_t *ooff = cw + CO_MAXCOV;
        if (mode == 2) {
          // the other region sorted by covariate (behind it), counts and offsets per covariate
          uint4 *sorted = const_cast<uint4 *>(other) + 2 * (size_t)n;
          const unsigned og = blocks_for(n, CO_TILE);  // (launched for the worst case; blocks behind the region's end leave at once)
          ELP_LAUNCH(c, "bqsr_other_hist", k_c3_other_hist, dim3(og), dim3(256), 0, other, n_other, cw);
          ELP_LAUNCH(c, "bqsr_other_offsets", k_c3_other_offsets, dim3(1), dim3(1), 0, (const uint32_t *)cw, ooff, ooff + CO_MAXCOV + 1);
          ELP_LAUNCH(c, "bqsr_other_scatter", k_c3_other_scatter, dim3(og), dim3(256), 0, other, n_other, ooff + CO_MAXCOV + 1, sorted);
          other = sorted;
        }
        // launch 1: the reads that are one run of matches, in the class-1 segments; launch 2: the others (indels, clipped windows,
        // descriptors) - one segment (its first slot: the zero in front of the offsets... ooff[0], cleared above), or one per covariate
        Count3Args A3{rec_cnt, (uint32_t)C3_CSTRIDE, seg_base, reinterpret_cast<const uint4 *>(recs), nseg, 0, mode == 2 ? (int)ncs : 0, c->uniform_len, c->qual.p,
                      c->seq4.p + elp_ctx::SEQ_FRONT, reinterpret_cast<const uint8_t *>(skipbits), reinterpret_cast<const uint4 *>(desc), c->cigar.p, cs_pool,
                      c->d_ref_seq.p, c->d_ref_seq_len.p, c->n_cov, (int)quals.size(), lmax, max_cycle, rsw3, rlog3, tb + nq, tb + nq + nc, c->err_flag.p};
        ELP_TRY(count3_launch(c, A3, qm, dyn3));
        A3.other = 1;
        A3.srecs = other;
        A3.seg_base = ooff;
        if (mode == 2) { A3.seg_cnt = cw; A3.cnt_stride = 1; A3.nseg = ncs; }
        else { A3.seg_cnt = n_other; A3.cnt_stride = 1; A3.nseg = 1; }
        ELP_TRY(count3_launch(c, A3, qm, dyn3));
        goto counted;
      }
      for (int cov0 = 0; cov0 < c->n_cov; cov0 += ncp)
      for (size_t q0 = 0; q0 < quals.size(); q0 += (size_t)qcap) {
        const int nqs = (int)std::min<size_t>((size_t)qcap, quals.size() - q0), ncov_pass = std::min(ncp, c->n_cov - cov0);
        QMap qm;
        memset(qm.slot, 254, sizeof qm.slot);
        for (int q : quals) qm.slot[q] = 255;
        for (int s = 0; s < nqs; s++) qm.slot[quals[q0 + s]] = (uint8_t)s;
        const size_t dyn = ((size_t)ncov_pass * (nqs + CT_XROWS) * rs_use + CT_PAD) * 4;
        CountArgs A{n, c->qual_bytes, c->qual_off.p, c->seq_off.p, c->qual.p, c->seq4.p, desc, c->cigar.p, cs_pool,
                    reinterpret_cast<const uint8_t *>(skipbits), c->d_ref_seq.p, c->d_ref_seq_len.p, c->n_ref, ncov_pass, nqs, lmax, rs_use, max_cycle,
                    tb + nq, tb + nq + nc, c->err_flag.p, c->tile_first.p, cov0};
        const bool ref_lds = c->n_ref <= REF_LDS;
#define ELP_COUNT_LAUNCH(CC, RL)                                                                                                              \
  do {                                                                                                                                        \
    if (big && mg) {                                                                                                                          \
      ELP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bqsr_count<CC, RL, 1024, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)); \
      ELP_LAUNCH(c, "bqsr_count", (k_bqsr_count<CC, RL, 1024, true>), dim3(grid), dim3(1024), dyn, A, qm);                                    \
    } else if (big) {                                                                                                                         \
      ELP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bqsr_count<CC, RL, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)); \
      ELP_LAUNCH(c, "bqsr_count", (k_bqsr_count<CC, RL, 1024>), dim3(grid), dim3(1024), dyn, A, qm);                                          \
    } else {                                                                                                                                  \
      ELP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bqsr_count<CC, RL, FL_THREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn)); \
      ELP_LAUNCH(c, "bqsr_count", (k_bqsr_count<CC, RL, FL_THREADS>), dim3(grid), dim3(FL_THREADS), dyn, A, qm);                              \
    }                                                                                                                                         \
  } while (0)
        if (check_cycle && ref_lds) ELP_COUNT_LAUNCH(true, true);
        else if (check_cycle) ELP_COUNT_LAUNCH(true, false);
        else if (ref_lds) ELP_COUNT_LAUNCH(false, true);
        else ELP_COUNT_LAUNCH(false, false);
#undef ELP_COUNT_LAUNCH
      }
    counted:
      uint32_t e[4];
      ELP_TRY(fetch_err(c, e));
      if ((e[0] & ~128u) != 0) return bqsr_error(c, e[0] & ~128u);
      if (!(e[0] & 128u)) break;
      // a counted base had a quality the sampled hint did not contain: take the exact set (full scan) and redo the count
      ELP_HIP(c, hipMemsetAsync(c->err_flag.p, 0, 4, st));
      c->have_qual_present = false;
      ELP_TRY(ensure_qual_present(c, true));
      if (attempt >= 2) return set_error(c, ELP_ERR_HIP, "BQSR: quality-slot retry did not converge");
      ELP_HIP(c, hipMemsetAsync(tb, 0, (nq + nc + nx) * sizeof(unsigned long long), st));
    }
    ELP_LAUNCH(c, "bqsr_qual_from_cycle", k_bqsr_qual_from_cycle, dim3(c->n_cov * ELP_NQUAL), dim3(256), 0, c->n_cov * ELP_NQUAL, ncyc_g,
               (const unsigned long long *)(tb + nq), tb);
  }
  c->tables_n = nq + nc + nx;
  c->tables_max_cycle = max_cycle;
  c->tables_quals[0] = c->qual_present[0] & ~0x3Full;  // (qualities below 6 are never counted)
  c->tables_quals[1] = c->qual_present[1];
  ELP_TRY(tables_written(c));
  if (!qual_tbl) return 0;  // tables stay in HBM (the count loop above fetched the error word behind the last kernel that can raise one)
  // the three tables lie behind each other on the device: one copy into pinned memory, then into the caller's arrays
  const size_t bytes = (nq + nc + nx) * 8;
  if (bytes > c->h_pinned_cap) {
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    c->h_pinned = nullptr; c->h_pinned_cap = 0;
    ELP_HIP(c, hipHostMalloc(&c->h_pinned, bytes, hipHostMallocDefault));
    c->h_pinned_cap = bytes;
  }
  ELP_HIP(c, hipMemcpyAsync(c->h_pinned, tb, bytes, hipMemcpyDeviceToHost, st));
  uint32_t e[4];
  ELP_TRY(fetch_err(c, e));
  if (e[0]) return bqsr_error(c, e[0]);
  const int64_t *hp = static_cast<const int64_t *>(c->h_pinned);
  memcpy(qual_tbl, hp, nq * 8);
  memcpy(cycle_tbl, hp + nq, nc * 8);
  memcpy(ctx_tbl, hp + nq + nc, nx * 8);
  return 0;
}

}  // namespace elp

using namespace elp;

extern "C" {

int elp_bqsr_set_reference(elp_ctx *c, int32_t refid, const uint8_t *bases, int64_t len) {
  if (!c || !c->have_header || refid < 0 || refid >= c->n_ref || len < 0 || (len && !bases)) return set_error(c, ELP_ERR_ARG, "elp_bqsr_set_reference: bad arguments");
  ELP_HIP(c, hipSetDevice(c->device));
  if (c->h_ref_seq[refid]) { (void)elp::stream_wait(c->stream); (void)hipFree(c->h_ref_seq[refid]); c->h_ref_seq[refid] = nullptr; }
  // the contig is kept as 4-bit base codes (k_pack_reference); the ASCII bytes only pass through scratch
  const int64_t packed = (len + 1) / 2;
  uint8_t *d = nullptr;
  ELP_HIP(c, hipMalloc((void **)&d, (size_t)(packed + REF_PAD)));
  ELP_HIP(c, hipMemsetAsync(d, 0x88, (size_t)(packed + REF_PAD), c->stream));
  if (len) {
    uint8_t *tmp;
    ELP_TRY(scratch(c, 7, (size_t)len + 16, &tmp));
    ELP_HIP(c, hipMemcpyAsync(tmp, bases, (size_t)len, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_pack_reference, dim3(blocks_for((uint64_t)packed, 256)), dim3(256), 0, c->stream, (const uint8_t *)tmp, len, d, packed);
    ELP_HIP(c, hipGetLastError());
  }
  ELP_HIP(c, elp::stream_wait(c->stream));
  c->h_ref_seq[refid] = d;
  c->h_ref_seq_len[refid] = len;
  c->ref_flags_dirty[refid] = 1;
  c->bqsr_ptrs_dirty = true;
  return 0;
}

int elp_bqsr_set_known_sites(elp_ctx *c, int32_t refid, const int32_t *start_end, int64_t n) {
  if (!c || !c->have_header || refid < 0 || refid >= c->n_ref || n < 0 || (n && !start_end)) return set_error(c, ELP_ERR_ARG, "elp_bqsr_set_known_sites: bad arguments");
  for (int64_t k = 1; k < n; k++)
    if (!(start_end[2 * k] > start_end[2 * k - 1])) return set_error(c, ELP_ERR_ARG, "known sites of refid %d are not sorted and flattened at index %lld", refid, (long long)k);
  ELP_HIP(c, hipSetDevice(c->device));
  if (c->h_sites[refid]) { (void)elp::stream_wait(c->stream); (void)hipFree(c->h_sites[refid]); c->h_sites[refid] = nullptr; }
  if (c->h_site_idx[refid]) { (void)hipFree(c->h_site_idx[refid]); c->h_site_idx[refid] = nullptr; }
  int32_t *d = nullptr;
  ELP_HIP(c, hipMalloc((void **)&d, (size_t)(2 * n + 4) * sizeof(int32_t)));
  if (n) ELP_HIP(c, hipMemcpyAsync(d, start_end, (size_t)(2 * n) * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  const int64_t nbuck = ((int64_t)c->h_ref_len[refid] >> 6) + 1;
  uint32_t *ix = nullptr;
  ELP_HIP(c, hipMalloc((void **)&ix, (size_t)(nbuck + 4) * sizeof(uint32_t)));
  hipLaunchKernelGGL(k_site_index, dim3(blocks_for((uint64_t)nbuck, 256)), dim3(256), 0, c->stream, (const int32_t *)d, n, nbuck, ix);
  ELP_HIP(c, hipGetLastError());
  ELP_HIP(c, elp::stream_wait(c->stream));
  c->h_site_idx[refid] = ix;
  c->h_sites[refid] = d;
  c->h_n_sites[refid] = n;
  c->ref_flags_dirty[refid] = 1;
  c->bqsr_ptrs_dirty = true;
  return 0;
}

int elp_bqsr_gather(elp_ctx *c, int max_cycle, int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl) {
  if (!c || !qual_tbl || !cycle_tbl || !ctx_tbl || max_cycle < 1) return set_error(c, ELP_ERR_ARG, "elp_bqsr_gather: bad arguments");
  ELP_HIP(c, hipSetDevice(c->device));
  return gather_impl(c, max_cycle, qual_tbl, cycle_tbl, ctx_tbl);
}

int elp_bqsr_gather_device(elp_ctx *c, int max_cycle) {
  if (!c || max_cycle < 1) return set_error(c, ELP_ERR_ARG, "elp_bqsr_gather_device: bad arguments");
  ELP_HIP(c, hipSetDevice(c->device));
  return gather_impl(c, max_cycle, nullptr, nullptr, nullptr);
}

int elp_bqsr_tables_fetch(elp_ctx *c, int64_t *qual_tbl, int64_t *cycle_tbl, int64_t *ctx_tbl) {
  if (!c || !qual_tbl || !cycle_tbl || !ctx_tbl) return ELP_ERR_ARG;
  if (!c->tables_n) return set_error(c, ELP_ERR_ARG, "elp_bqsr_tables_fetch: no device tables (elp_bqsr_gather_device)");
  ELP_HIP(c, hipSetDevice(c->device));
  const size_t nq = (size_t)c->n_cov * ELP_NQUAL * 2, nc = nq * (size_t)(2 * c->tables_max_cycle + 1), nx = nq * ELP_NCTX;
  const size_t bytes = (nq + nc + nx) * 8;
  if (bytes > c->h_pinned_cap) {
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    c->h_pinned = nullptr; c->h_pinned_cap = 0;
    ELP_HIP(c, hipHostMalloc(&c->h_pinned, bytes, hipHostMallocDefault));
    c->h_pinned_cap = bytes;
  }
  // on the copy stream, behind the last writer of the tables: a host thread can fetch (and finalise) while another one runs the
  // next stage on the context's stream (the copy is 6 MB over PCIe: ~0.2 ms during which the GPU would otherwise sit idle)
  if (!c->copy_stream) ELP_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  if (c->tables_ev) ELP_HIP(c, hipStreamWaitEvent(c->copy_stream, c->tables_ev, 0));
  else ELP_HIP(c, elp::stream_wait(c->stream));
  ELP_HIP(c, hipMemcpyAsync(c->h_pinned, c->dev_tables.p, bytes, hipMemcpyDeviceToHost, c->copy_stream));
  ELP_HIP(c, elp::stream_wait(c->copy_stream));
  const int64_t *hp = static_cast<const int64_t *>(c->h_pinned);
  memcpy(qual_tbl, hp, nq * 8);
  memcpy(cycle_tbl, hp + nq, nc * 8);
  memcpy(ctx_tbl, hp + nq + nc, nx * 8);
  return 0;
}

// the qualities that had table slots in the gather that made the device tables: bit q of bits[q / 64]
int elp_bqsr_quals_counted(elp_ctx *c, uint64_t *bits) {
  if (!c || !bits) return ELP_ERR_ARG;
  if (!c->tables_n) return set_error(c, ELP_ERR_ARG, "elp_bqsr_quals_counted: no device tables (elp_bqsr_gather_device)");
  bits[0] = c->tables_quals[0];
  bits[1] = c->tables_quals[1];
  return 0;
}

// elp_bqsr_tables_fetch for the rows of the qualities `quals` only: q_rows [n_cov][n_quals][2], c_rows [n_cov][n_quals][2*max_cycle+1][2],
// x_rows [n_cov][n_quals][16][2] - packed on the device, one copy.  Returns 1 (and copies nothing) if a quality that was not asked for has
// observations (tables that were summed with another context's or rank's: the caller fetches the dense tables then).
int elp_bqsr_tables_fetch_rows(elp_ctx *c, const uint8_t *quals, int n_quals, int64_t *q_rows, int64_t *c_rows, int64_t *x_rows) {
  if (!c || n_quals < 0 || n_quals > ELP_NQUAL || (n_quals && (!quals || !q_rows || !c_rows || !x_rows))) return ELP_ERR_ARG;
  if (!c->tables_n) return set_error(c, ELP_ERR_ARG, "elp_bqsr_tables_fetch_rows: no device tables (elp_bqsr_gather_device)");
  for (int k = 0; k < n_quals; k++)
    if (quals[k] >= ELP_NQUAL) return set_error(c, ELP_ERR_ARG, "elp_bqsr_tables_fetch_rows: quality %d", (int)quals[k]);
  ELP_HIP(c, hipSetDevice(c->device));
  const int ncyc = 2 * c->tables_max_cycle + 1;
  const size_t n_rows = (size_t)c->n_cov * (size_t)n_quals, row_w = 2 + (size_t)ncyc * 2 + ELP_NCTX * 2, words = n_rows * row_w;
  const size_t bytes = words * 8 + 256;
  if (bytes > c->h_pinned_cap) {
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    c->h_pinned = nullptr; c->h_pinned_cap = 0;
    ELP_HIP(c, hipHostMalloc(&c->h_pinned, bytes, hipHostMallocDefault));
    c->h_pinned_cap = bytes;
  }
  ELP_TRY(ensure(c, c->tables_pack, words + 64));
  if (!c->copy_stream) ELP_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  if (c->tables_ev) ELP_HIP(c, hipStreamWaitEvent(c->copy_stream, c->tables_ev, 0));
  else ELP_HIP(c, elp::stream_wait(c->stream));
  // the quality list and the "uncovered" word ride in the pack buffer's tail
  uint8_t *d_quals = reinterpret_cast<uint8_t *>(c->tables_pack.p + words);
  uint32_t *d_unc = reinterpret_cast<uint32_t *>(c->tables_pack.p + words + 16);
  uint8_t *hq = static_cast<uint8_t *>(c->h_pinned) + words * 8;
  memset(hq, 0, 256);
  if (n_quals) memcpy(hq, quals, (size_t)n_quals);
  ELP_HIP(c, hipMemcpyAsync(d_quals, hq, 128 + 8, hipMemcpyHostToDevice, c->copy_stream));  // (also clears the word)
  const size_t work = std::max(words, (size_t)c->n_cov * ELP_NQUAL);
  hipLaunchKernelGGL(k_tables_pack_rows, dim3(blocks_for(work, 256)), dim3(256), 0, c->copy_stream, (const unsigned long long *)c->dev_tables.p, c->n_cov, ncyc,
                     (const uint8_t *)d_quals, n_quals, c->tables_pack.p, d_unc);
  ELP_HIP(c, hipGetLastError());
  ELP_HIP(c, hipMemcpyAsync(c->h_pinned, c->tables_pack.p, words * 8, hipMemcpyDeviceToHost, c->copy_stream));
  uint32_t unc = 0;
  ELP_HIP(c, hipMemcpyAsync(&unc, d_unc, 4, hipMemcpyDeviceToHost, c->copy_stream));
  ELP_HIP(c, elp::stream_wait(c->copy_stream));
  if (unc) return 1;
  const int64_t *hp = static_cast<const int64_t *>(c->h_pinned);
  if (n_rows) {
    memcpy(q_rows, hp, n_rows * 2 * 8);
    memcpy(c_rows, hp + n_rows * 2, n_rows * (size_t)ncyc * 2 * 8);
    memcpy(x_rows, hp + n_rows * 2 + n_rows * (size_t)ncyc * 2, n_rows * ELP_NCTX * 2 * 8);
  }
  return 0;
}



// Distinct rows of the resident part of the dense LUT (three small kernels): wk = counters | slots | slot ids | row -> slot | t1 | t2.
struct LutDict { uint32_t *slots, *slot_id, *row_slot, *counter; uint16_t *t1; uint8_t *t2; uint32_t n_slots; size_t n_rows, n1; };
constexpr uint32_t LUT_T2_CAP = 3855;  // 16-bit byte offsets
static size_t lut_dict_words(int n_cov, int n_qi, int lmax, uint32_t *n_slots_out) {
  const size_t w = 2 * (size_t)lmax + 1, n_rows = (size_t)n_cov * (size_t)std::max(n_qi, 0) * w, n1 = (size_t)n_cov * (size_t)(n_qi + 1) * w;
  uint32_t n_slots = 1024;
  while (n_slots < 2 * n_rows) n_slots <<= 1;
  *n_slots_out = n_slots;
  const size_t t2_bytes = std::max<size_t>((size_t)(LUT_T2_CAP + 1) * 17, (size_t)n_cov * LUT_PC_ROWS * 17);  // one dictionary, or one per covariate
  return 256 + (size_t)2 * n_slots + n_rows + n1 + t2_bytes / 4 + 64 + 16;
}
static LutDict lut_dict_layout(uint32_t *wk, int n_cov, int n_qi, int lmax, uint32_t n_slots) {
  const size_t w = 2 * (size_t)lmax + 1, n_rows = (size_t)n_cov * (size_t)std::max(n_qi, 0) * w, n1 = (size_t)n_cov * (size_t)(n_qi + 1) * w;
  LutDict D;
  D.n_slots = n_slots; D.n_rows = n_rows; D.n1 = n1;
  D.counter = wk;  // [256] (own words: the err_flag mailbox other stages use is not touched from the upload thread)
  D.slots = wk + 256; D.slot_id = D.slots + n_slots; D.row_slot = D.slots + 2 * (size_t)n_slots;
  D.t1 = reinterpret_cast<uint16_t *>(D.row_slot + n_rows);
  D.t2 = reinterpret_cast<uint8_t *>(D.t1 + ((n1 + 1) & ~(size_t)1));
  return D;
}
// (plain launches, no profiling brackets: also called from the upload thread while the context's own stream is busy)
static int lut_dict_build(elp_ctx *c, hipStream_t st, const LutDict &D, const uint8_t *dl, int qlo, int n_qi, int lmax, int max_cycle, bool per_cov) {
  ELP_HIP(c, hipMemsetAsync(D.slots, 0xFF, (size_t)D.n_slots * 4, st));
  ELP_HIP(c, hipMemsetAsync(D.counter, 0, 256 * 4, st));
  LutRows R{dl, c->n_cov, qlo, n_qi, lmax, max_cycle, per_cov ? 1 : 0};
  const uint32_t cap = per_cov ? LUT_PC_ROWS : LUT_T2_CAP;
  hipLaunchKernelGGL(k_lut_rows_insert, dim3(blocks_for(D.n_rows, 256)), dim3(256), 0, st, R, (int)D.n_rows, D.slots, D.n_slots - 1, D.row_slot);
  hipLaunchKernelGGL(k_lut_rows_number, dim3(blocks_for(D.n_slots, 256)), dim3(256), 0, st, R, (const uint32_t *)D.slots, D.n_slots, D.slot_id, D.counter, D.t2, cap);
  hipLaunchKernelGGL(k_lut_rows_index, dim3(blocks_for(std::max<size_t>(D.n1, 17 * 256), 256)), dim3(256), 0, st, R, (const uint32_t *)D.row_slot, (const uint32_t *)D.slot_id,
                     (const uint32_t *)D.counter, D.t1, D.t2, cap);
  ELP_HIP(c, hipGetLastError());
  return 0;
}
// How apply3.hip (read sets of one length) takes a LUT: 0 not at all, 1 the level-1 tables of every covariate in one workgroup's LDS,
// 2 split by covariate (their level 1 does not fit - many read groups -, or elp_set_tuning "apply_kernel" = 3): a workgroup holds one
// covariate's tables at a time
static int apply3_mode(const elp_ctx *c, int n_qi, int lmax, size_t *dyn_out) {
  if (c->tune.apply_kernel != 3 && apply3_bytes(c->n_cov, n_qi, lmax, dyn_out) == 0) return 1;
  if (apply3_bytes(1, n_qi, lmax, dyn_out) == 0) return 2;
  return 0;
}
// the resident quality range of ApplyBQSR's LDS tables, from the quality hint (-1: no quality >= 6 seen)
static void lut_quality_range(const elp_ctx *c, int *qlo, int *qhi) {
  *qlo = 0; *qhi = -1;
  for (int q = 6; q < ELP_NQUAL; q++)
    if ((q < 64 ? (c->qual_present[0] >> q) : (c->qual_present[1] >> (q - 64))) & 1ull) { if (*qhi < 0) *qlo = q; *qhi = q; }
}

// The LUT's way to the device ahead of the apply call: from the thread that built it, on the context's copy stream, while the context's
// own stream still runs the sort / metrics pass (6.4 MB at --max-cycle 500: ~0.2 ms that elp_bqsr_apply otherwise spends in front of its
// first kernel).  The LUT lives in a buffer of its own (not in the scratch pool: other stages are running).
static int lut_uploaded(elp_ctx *c, int max_cycle);
int elp_bqsr_lut_upload(elp_ctx *c, int max_cycle, const uint8_t *lut, const uint8_t *cov_present) {
  if (!c || !lut || !cov_present || max_cycle < 1) return set_error(c, ELP_ERR_ARG, "elp_bqsr_lut_upload: bad arguments");
  ELP_HIP(c, hipSetDevice(c->device));
  const size_t lut_bytes = (size_t)c->n_cov * ELP_NQUAL * (2 * (size_t)max_cycle + 1) * 17, all = lut_bytes + (size_t)c->n_cov;
  if (c->lut_ev) ELP_HIP(c, hipEventSynchronize(c->lut_ev));  // (a previous upload still in flight reads the pinned buffer)
  // a LUT that already sits in page-locked memory (elp_pinned_alloc) is copied from where it is - with many read groups the LUT is tens of
  // megabytes and the staging copy below was the longest part of the host's table path; the caller then leaves it alone until the
  // elp_bqsr_apply that uses it has been called and the context synchronised
  hipPointerAttribute_t pa;
  const bool caller_pinned = hipPointerGetAttributes(&pa, lut) == hipSuccess && pa.type == hipMemoryTypeHost;
  if (!caller_pinned) (void)hipGetLastError();  // (ordinary memory: the query fails, by design)
  const size_t staged = caller_pinned ? (size_t)c->n_cov : all;
  if (staged > c->lut_pinned_cap) {
    if (c->lut_pinned) (void)hipHostFree(c->lut_pinned);
    c->lut_pinned = nullptr; c->lut_pinned_cap = 0;
    ELP_HIP(c, hipHostMalloc(&c->lut_pinned, staged, hipHostMallocDefault));
    c->lut_pinned_cap = staged;
  }
  ELP_TRY(ensure(c, c->lut_dev, all + 64));
  if (!caller_pinned) memcpy(c->lut_pinned, lut, lut_bytes);
  memcpy(static_cast<uint8_t *>(c->lut_pinned) + (caller_pinned ? 0 : lut_bytes), cov_present, (size_t)c->n_cov);
  if (!c->lut_ev) ELP_HIP(c, hipEventCreateWithFlags(&c->lut_ev, hipEventDisableTiming));
  if (!c->copy_stream) ELP_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));  // (not the NULL stream all contexts share)
  if (c->apply_ev) ELP_HIP(c, hipStreamWaitEvent(c->copy_stream, c->apply_ev, 0));  // an apply that still reads the previous LUT
  if (caller_pinned) {
    ELP_HIP(c, hipMemcpyAsync(c->lut_dev.p, lut, lut_bytes, hipMemcpyHostToDevice, c->copy_stream));
    ELP_HIP(c, hipMemcpyAsync(c->lut_dev.p + lut_bytes, c->lut_pinned, (size_t)c->n_cov, hipMemcpyHostToDevice, c->copy_stream));
  } else {
    ELP_HIP(c, hipMemcpyAsync(c->lut_dev.p, c->lut_pinned, all, hipMemcpyHostToDevice, c->copy_stream));
  }
  return lut_uploaded(c, max_cycle);
}

// behind the LUT's arrival in lut_dev on the copy stream: the row dictionary apply3 works from - if what it depends on is known now (the
// quality hint of the gather that produced these tables, a read set of one length): 0.25 ms that elp_bqsr_apply otherwise spends in front of
// its kernel - and the event the apply waits for
static int lut_uploaded(elp_ctx *c, int max_cycle) {
  c->dict_ready = false;
  const bool force_old = c->tune.apply_kernel == 1;
  if (!force_old && c->have_qual_present && c->uniform_n == c->n && c->uniform_len >= 16 && c->n > 0 && (int64_t)c->max_l_seq <= (int64_t)max_cycle) {
    int qlo, qhi;
    lut_quality_range(c, &qlo, &qhi);
    const int lmax = (int)std::max<uint32_t>(c->max_l_seq, 1);
    size_t dyn3 = 0;
    const int a3 = qhi >= 0 ? apply3_mode(c, qhi - 6 + 1, lmax, &dyn3) : 0;
    if (a3) {
      const int n_qi = qhi - 6 + 1;  // (apply3: resident from quality 6 on)
      uint32_t n_slots = 0;
      const size_t words = lut_dict_words(c->n_cov, n_qi, lmax, &n_slots);
      if ((size_t)c->n_cov * (size_t)n_qi * (size_t)(2 * lmax + 1) < (1u << 22)) {
        ELP_TRY(ensure(c, c->lut_wk, words));
        const LutDict D = lut_dict_layout(c->lut_wk.p, c->n_cov, n_qi, lmax, n_slots);
        ELP_TRY(lut_dict_build(c, c->copy_stream, D, c->lut_dev.p, 6, n_qi, lmax, max_cycle, a3 == 2));
        c->dict_qlo = 6; c->dict_nqi = n_qi; c->dict_lmax = lmax; c->dict_cycle = max_cycle; c->dict_ncov = c->n_cov; c->dict_per_cov = a3 == 2;
        c->dict_ready = true;
      }
    }
  }
  ELP_HIP(c, hipEventRecord(c->lut_ev, c->copy_stream));
  c->lut_uploaded_cycle = max_cycle;
  return 0;
}

// elp_bqsr_lut_upload for the LUT in rows form (the host library's elp_bqsr_tables_build_lut_rows): n_cov x n_quals rows + one default byte
// per other row instead of n_cov x 94 rows - with 16 read groups 1.9 MB instead of 25.6 MB over PCIe (and 13 x less for the host to fill);
// a kernel on the copy stream expands it into the dense LUT every apply kernel reads.
int elp_bqsr_lut_upload_rows(elp_ctx *c, int max_cycle, const uint8_t *quals, int n_quals, const uint8_t *rows, const uint8_t *defaults, const uint8_t *cov_present) {
  if (!c || max_cycle < 1 || n_quals < 0 || n_quals > ELP_NQUAL || (n_quals && (!quals || !rows)) || !defaults || !cov_present)
    return set_error(c, ELP_ERR_ARG, "elp_bqsr_lut_upload_rows: bad arguments");
  uint8_t slot_of[ELP_NQUAL];
  memset(slot_of, 255, sizeof slot_of);
  for (int k = 0; k < n_quals; k++) {
    if (quals[k] >= ELP_NQUAL || slot_of[quals[k]] != 255) return set_error(c, ELP_ERR_ARG, "elp_bqsr_lut_upload_rows: quality list");
    slot_of[quals[k]] = (uint8_t)k;
  }
  ELP_HIP(c, hipSetDevice(c->device));
  const size_t ncyc = 2 * (size_t)max_cycle + 1, row_b = ncyc * 17;
  const size_t rows_bytes = (size_t)c->n_cov * (size_t)n_quals * row_b, def_bytes = (size_t)c->n_cov * ELP_NQUAL;
  const size_t lut_bytes = (size_t)c->n_cov * ELP_NQUAL * row_b, small = def_bytes + ELP_NQUAL + (size_t)c->n_cov;  // defaults | slot_of | cov_present
  if (c->lut_ev) ELP_HIP(c, hipEventSynchronize(c->lut_ev));  // (a previous upload still in flight reads the pinned buffer)
  hipPointerAttribute_t pa;
  const bool caller_pinned = rows_bytes && hipPointerGetAttributes(&pa, rows) == hipSuccess && pa.type == hipMemoryTypeHost;
  if (!caller_pinned) (void)hipGetLastError();
  const size_t staged = small + (caller_pinned ? 0 : rows_bytes);
  if (staged > c->lut_pinned_cap) {
    if (c->lut_pinned) (void)hipHostFree(c->lut_pinned);
    c->lut_pinned = nullptr; c->lut_pinned_cap = 0;
    ELP_HIP(c, hipHostMalloc(&c->lut_pinned, staged, hipHostMallocDefault));
    c->lut_pinned_cap = staged;
  }
  uint8_t *hp = static_cast<uint8_t *>(c->lut_pinned);
  memcpy(hp, defaults, def_bytes);
  memcpy(hp + def_bytes, slot_of, ELP_NQUAL);
  memcpy(hp + def_bytes + ELP_NQUAL, cov_present, (size_t)c->n_cov);
  if (!caller_pinned && rows_bytes) memcpy(hp + small, rows, rows_bytes);
  ELP_TRY(ensure(c, c->lut_dev, lut_bytes + (size_t)c->n_cov + 64));
  ELP_TRY(ensure(c, c->lut_rows_dev, rows_bytes + small + 64));
  if (!c->lut_ev) ELP_HIP(c, hipEventCreateWithFlags(&c->lut_ev, hipEventDisableTiming));
  if (!c->copy_stream) ELP_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  if (c->apply_ev) ELP_HIP(c, hipStreamWaitEvent(c->copy_stream, c->apply_ev, 0));  // an apply that still reads the previous LUT
  uint8_t *d_small = c->lut_rows_dev.p, *d_rows = c->lut_rows_dev.p + ((small + 63) & ~(size_t)63);
  ELP_HIP(c, hipMemcpyAsync(d_small, hp, small, hipMemcpyHostToDevice, c->copy_stream));
  if (rows_bytes) ELP_HIP(c, hipMemcpyAsync(d_rows, caller_pinned ? rows : hp + small, rows_bytes, hipMemcpyHostToDevice, c->copy_stream));
  hipLaunchKernelGGL(k_lut_expand_rows, dim3(blocks_for(lut_bytes, 256)), dim3(256), 0, c->copy_stream, (const uint8_t *)d_rows, (const uint8_t *)d_small,
                     (const uint8_t *)(d_small + def_bytes), c->n_cov, n_quals, (int)ncyc, c->lut_dev.p);
  ELP_HIP(c, hipGetLastError());
  ELP_HIP(c, hipMemcpyAsync(c->lut_dev.p + lut_bytes, d_small + def_bytes + ELP_NQUAL, (size_t)c->n_cov, hipMemcpyDeviceToDevice, c->copy_stream));
  return lut_uploaded(c, max_cycle);
}

static int bqsr_apply_impl(elp_ctx *c, int max_cycle, const uint8_t *lut, const uint8_t *cov_present);
int elp_bqsr_apply(elp_ctx *c, int max_cycle, const uint8_t *lut, const uint8_t *cov_present) {
  const int rc = bqsr_apply_impl(c, max_cycle, lut, cov_present);
  if (c && !lut && rc == 0) {  // the kernels just queued read the uploaded LUT: the next upload waits for them
    if (!c->apply_ev && hipEventCreateWithFlags(&c->apply_ev, hipEventDisableTiming) != hipSuccess) return set_error(c, ELP_ERR_HIP, "hipEventCreate failed");
    ELP_HIP(c, hipEventRecord(c->apply_ev, c->stream));
  }
  return rc;
}
static int bqsr_apply_impl(elp_ctx *c, int max_cycle, const uint8_t *lut, const uint8_t *cov_present) {
  if (!c || max_cycle < 1 || (lut != nullptr) != (cov_present != nullptr)) return set_error(c, ELP_ERR_ARG, "elp_bqsr_apply: bad arguments");
  if (!lut && c->lut_uploaded_cycle != max_cycle) return set_error(c, ELP_ERR_ARG, "elp_bqsr_apply: no LUT given and none uploaded for this --max-cycle (elp_bqsr_lut_upload)");
  ELP_HIP(c, hipSetDevice(c->device));
  if (c->n_cov > 255) return set_error(c, ELP_ERR_UNSUPPORTED, "more than 255 read-group covariates");
  const size_t ncyc = 2 * (size_t)max_cycle + 1;
  const size_t lut_bytes = (size_t)c->n_cov * ELP_NQUAL * ncyc * 17;
  ELP_TRY(ensure_adapted(c, false));  // low-quality-tail bounds per read (adapt_score)
  uint8_t *dl;
  if (lut) {
    ELP_TRY(scratch(c, 0, lut_bytes + (size_t)c->n_cov + 64, &dl));
    ELP_HIP(c, hipMemcpyAsync(dl, lut, lut_bytes, hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipMemcpyAsync(dl + lut_bytes, cov_present, (size_t)c->n_cov, hipMemcpyHostToDevice, c->stream));
  } else {
    dl = c->lut_dev.p;  // uploaded ahead of the call: this stream waits for the copy, not the host
    ELP_HIP(c, hipStreamWaitEvent(c->stream, c->lut_ev, 0));
  }
  const uint64_t n = c->n;
  if (n) {
    if (c->qual_bytes) {
      typedef ApplyBody<false, 1> AB;
      const uint64_t nsteps = flat_steps<AB>(c->qual_bytes);
      ELP_TRY(ensure_flat_index(c));
      ELP_TRY(ensure_qual_present(c));  // the resident quality range comes from a sample of the column (a hint: qualities outside it take the fix-up path)
      int qlo = 0, qhi = -1;
      lut_quality_range(c, &qlo, &qhi);
      const int lmax = (int)std::max<uint32_t>(c->max_l_seq, 1);
      const bool chk = (int64_t)c->max_l_seq > (int64_t)max_cycle;
      // apply3.hip takes read sets of one length (elp_set_tuning "apply_kernel" = 1 forces k_bqsr_apply_flat: A/B measurements); its level-1 table is
      // resident from quality 6 on, whatever the smallest sampled quality was
      const bool force_old = c->tune.apply_kernel == 1;  // elp_set_tuning
      ELP_TRY(ensure_uniform_len(c));
      const bool want3 = !force_old && !chk && c->uniform_len >= 16 && qhi >= 0;  // (apply3 works in whole 16-byte blocks)
      if (want3) qlo = 6;
      ApplyArgs A{n, c->qual_bytes, c->qual_off.p, c->seq_off.p, c->qual.p, c->seq4.p, c->flag.p, c->rgid.p, c->rg_cov.p, c->l_seq.p, c->qbounds.p,
                  dl + lut_bytes, c->tile_first.p, dl, max_cycle, c->err_flag.p, nullptr, nullptr, c->n_cov, 0, 0, lmax, 0};
      int mode = 0;
      size_t dyn = 0;
      const int n_qi = qhi - qlo + 1, w = 2 * lmax + 1;
      const size_t static_lds = sizeof(FlatLds<AB::RMAX>) + (size_t)AB::RMAX * 12 + 512;
      const size_t n_rows = (size_t)c->n_cov * (size_t)std::max(n_qi, 0) * (size_t)w, n1 = (size_t)c->n_cov * (size_t)(n_qi + 1) * (size_t)w;
      // apply3.hip first: the distinct rows of the resident part of the LUT (one dictionary, or - covariate split - one per covariate), built
      // behind the LUT's upload if that was possible (elp_bqsr_lut_upload), else here; the number(s) of distinct rows stay on the device: if
      // there are more than the one-byte ids hold the kernel says so and leaves without touching a byte - the next form takes over
      size_t dyn3 = 0;
      int a3 = (want3 && lmax <= max_cycle && n_rows < (1u << 22)) ? apply3_mode(c, n_qi, lmax, &dyn3) : 0;
      int dict_form = 0;  // the dictionary in `wk`: 0 none, 1 one for all covariates, 2 one per covariate
      uint32_t n_slots = 0;
      const size_t dict_words = lut_dict_words(c->n_cov, std::max(n_qi, 0), lmax, &n_slots);
      const bool prebuilt = !lut && c->dict_ready && c->dict_qlo == qlo && c->dict_nqi == n_qi && c->dict_lmax == lmax && c->dict_cycle == max_cycle &&
                            c->dict_ncov == c->n_cov;
      uint32_t *wk = c->lut_wk.p;
      if (prebuilt) dict_form = c->dict_per_cov ? 2 : 1;
      else if (a3 || (!chk && qhi >= 0 && lmax <= max_cycle && n1 + static_lds <= 160 * 1024 && n_rows < (1u << 22))) ELP_TRY(scratch(c, 4, dict_words, &wk));
      while (a3) {
        const LutDict D = lut_dict_layout(wk, c->n_cov, n_qi, lmax, n_slots);
        if (dict_form != a3) {
          c->dict_ready = false;  // (a prebuilt dictionary of the other form is overwritten)
          ELP_TRY(lut_dict_build(c, c->stream, D, dl, qlo, n_qi, lmax, max_cycle, a3 == 2));
          dict_form = a3;
        }
        ELP_TRY(apply3_launch(c, max_cycle, dl, dl + lut_bytes, D.t1, D.t2, D.counter, n_qi, lmax, dyn3, a3 == 2));
        uint32_t e3[4];
        ELP_TRY(fetch_err(c, e3));
        if ((e3[0] & ~512u) != 0) return bqsr_error(c, e3[0] & ~512u);
        if (!(e3[0] & 512u)) {
          c->adapted = false;
          c->have_qual_present = false;
          return 0;
        }
        ELP_HIP(c, hipMemsetAsync(c->err_flag.p, 0, 4, c->stream));
        // too many distinct rows for one dictionary: one per covariate (a covariate's rows are the n_cov-th part); else the general kernel
        a3 = (a3 == 1 && c->n_cov > 1 && apply3_bytes(1, n_qi, lmax, &dyn3) == 0) ? 2 : 0;
      }
      if (!chk && qhi >= 0 && lmax <= max_cycle && n1 + static_lds <= 160 * 1024 && n_rows < (1u << 22)) {
        const LutDict D = lut_dict_layout(wk, c->n_cov, n_qi, lmax, n_slots);
        if (dict_form != 1) {
          c->dict_ready = false;
          ELP_TRY(lut_dict_build(c, c->stream, D, dl, qlo, n_qi, lmax, max_cycle, false));
        }
        uint32_t *counter = D.counter;
        uint16_t *t1 = D.t1;
        uint8_t *t2 = D.t2;
        const uint32_t t2_cap = LUT_T2_CAP;
        uint32_t n_dict = 0;
        ELP_HIP(c, hipMemcpyAsync(&n_dict, counter, 4, hipMemcpyDeviceToHost, c->stream));
        ELP_HIP(c, elp::stream_wait(c->stream));
        if (n_dict < t2_cap) {
          const int m = n_dict + 1 <= 256 ? 1 : 2;
          const size_t bytes = ((n1 * (size_t)m + 15) & ~(size_t)15) + (size_t)(n_dict + 1) * (m == 1 ? 32 : 17) + 16;
          if (bytes + static_lds <= 160 * 1024) {
            mode = m;
            dyn = bytes;
            A.t1 = t1; A.t2 = t2; A.n_qi = n_qi; A.qlo = qlo; A.n_dict = (int)n_dict;
          }
        }
      }
      if (mode) {
        const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(3, (160 * 1024) / (dyn + static_lds)));
        const unsigned g1 = (unsigned)std::min<uint64_t>(nsteps, (uint64_t)c->n_cu * per_cu);
        if (mode == 1) {
          ELP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bqsr_apply_flat<false, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
          ELP_LAUNCH(c, "bqsr_apply", (k_bqsr_apply_flat<false, 1>), dim3(g1), dim3(FL_THREADS), dyn, A);
        } else {
          ELP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bqsr_apply_flat<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
          ELP_LAUNCH(c, "bqsr_apply", (k_bqsr_apply_flat<false, 2>), dim3(g1), dim3(FL_THREADS), dyn, A);
        }
      } else {
        const unsigned grid = (unsigned)std::min<uint64_t>(nsteps, (uint64_t)c->n_cu * 8);
        if (chk) ELP_LAUNCH(c, "bqsr_apply", (k_bqsr_apply_flat<true, 0>), dim3(grid), dim3(FL_THREADS), 0, A);
        else ELP_LAUNCH(c, "bqsr_apply", (k_bqsr_apply_flat<false, 0>), dim3(grid), dim3(FL_THREADS), 0, A);
      }
    }
  }
  uint32_t e[4];
  ELP_TRY(fetch_err(c, e));
  if (e[0]) return bqsr_error(c, e[0]);
  c->adapted = false;  // scores depend on QUAL
  c->have_qual_present = false;
  return 0;
}

}  // extern "C"
