// sort.hip — adapt (unclipped position, Phred-sum score, coordinate key) and the coordinate sort.
//
// Reference: sam.CoordinateLess + modFlag (sam/sam-types.go:408-473), By.ParallelStableSort (:639-641),
// computeUnclippedPosition / computePhredScore (filters/mark-duplicates.go:57-110).
//
// Sort = (1) 64-bit primary key {refid (unmapped last), POS, strand} sorted by the LSD radix sort of radix.hip,
//        (2) tie resolution inside runs of equal primary key with the comparator tail
//            {QNAME bytes, modFlag, MAPQ, [NextREFID, PNEXT if paired], TLEN}, ties keep staging order:
//            runs <= TIE_SMALL records: every record computes its rank in the run directly (all-pairs);
//            larger runs (the unmapped block, pile-ups): LSD radix over the live bytes of the zero-padded comparator string,
//            8 bytes per round, then one stable round on the run id.
#include <cstdlib>

#include "apply_rec.hpp"
#include "common.hpp"
#include "flat.hpp"

namespace elp {

constexpr int TIE_SMALL = 48;

// ------------------------------------------------------------------ adapt
// fixed-field part: unclipped 5' position (computeUnclippedPosition :79-110) and the CoordinateLess primary key
__global__ __launch_bounds__(256) void k_adapt_fixed(uint64_t n, const int32_t *__restrict__ pos, const int32_t *__restrict__ refid,
                                                     const uint16_t *__restrict__ flag, const uint64_t *__restrict__ cigar_off,
                                                     const uint32_t *__restrict__ cigar, int32_t *__restrict__ upos, int32_t *__restrict__ score,
                                                     uint64_t *__restrict__ key, uint32_t n_ref, int pos_bits,
                                                     const uint8_t *__restrict__ has_sr) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint16_t f = flag[i];
  const int32_t p = pos[i];
  const int32_t r = refid[i];
  // CoordinateLess primary key: REFID ascending with negative last (:429-432), POS (:433-436), forward before reverse (:437-438)
  // The key is packed into as few bits as the data needs (unmapped = one code above the last contig; POS in pos_bits bits, the
  // width of the largest staged POS), so that the LSD radix sort has as few live digit positions as possible.
  // Records with the sr tag are dropped behind the mark-duplicates filter (RemoveOptionalReads, cmd/filter.go:803) and never reach
  // the sort: they get one more code above "unmapped", i.e. the tail of the permutation (elp_num_sorted()).
  const uint64_t ru = has_sr[i] ? (uint64_t)n_ref + 1 : (r < 0 ? (uint64_t)n_ref : (uint64_t)(uint32_t)r);
  key[i] = (ru << (pos_bits + 1)) | ((uint64_t)(uint32_t)p << 1) | ((f & F_REVERSED) ? 1ull : 0ull);
  int32_t up = 0;
  if ((f & (F_UNMAPPED | F_SECONDARY | F_SUPPLEMENTARY)) == 0) {  // mark-duplicates.go:427,436
    const uint64_t c0 = cigar_off[i], c1 = cigar_off[i + 1];
    up = p;
    if (c1 > c0) {
      if (f & F_REVERSED) {  // :90-100
        int32_t clipped = 1;
        up--;
        for (uint64_t k = c1; k-- > c0;) {
          const uint32_t c = cigar[k];
          const uint32_t op = c & 0xF;
          const int32_t isclip = (op == OP_S || op == OP_H) ? 1 : 0;
          const int32_t isref = op_consumes_ref(op) ? 1 : 0;
          clipped *= isclip;
          up += (isref | clipped) * (int32_t)(c >> 4);
        }
      } else {  // :101-108
        for (uint64_t k = c0; k < c1; k++) {
          const uint32_t c = cigar[k];
          const uint32_t op = c & 0xF;
          if (!(op == OP_S || op == OP_H)) break;
          up -= (int32_t)(c >> 4);
        }
      }
    }
  }
  upos[i] = up;
  score[i] = 0;
}

// byte mask (0xFF per byte) for the first n bytes (n unclamped) of a 32-bit word
__device__ __forceinline__ uint32_t first_bytes32(int n) { return n <= 0 ? 0u : (n >= 4 ? 0xFFFFFFFFu : (0xFFFFFFFFu >> (32 - 8 * n))); }

// computePhredScore :57-68 as a flat stream over the QUAL column: sum of qualities >= 15 per duplicate-marking candidate;
// any quality > 93 in a candidate is an error.  SWAR over the lane's 16 bytes, v_sad_u8 for the byte sums.
struct ScoreBody {
  static constexpr int NT = FL_THREADS, TILES = 4, RMAX = 1024;
  const uint16_t *__restrict__ flag;
  const uint8_t *__restrict__ qual;
  int32_t *score;
  uint64_t *qbounds;
  int32_t *acc;    // LDS [RMAX]: per-read partial sums of this group (LDS atomics; one global store per read)
  uint32_t *lo;    // LDS [RMAX]: index of the first quality > 2 of the read (0xFFFFFFFF: none yet)
  uint32_t *hi;    // LDS [RMAX]: 1 + index of the last quality > 2 (0: none yet)
  uint8_t *cand;   // LDS [RMAX]: duplicate-marking candidate?
  const uint4 *mask;  // LDS [17]: byte masks of the first nb bytes of a block, one 16-byte read instead of sixteen instructions
  uint32_t bad;

  __device__ __forceinline__ void stage(uint32_t g0, uint32_t ng) {
    for (uint32_t k = threadIdx.x; k < ng; k += NT) {
      acc[k] = 0;
      lo[k] = 0xFFFFFFFFu;
      hi[k] = 0u;
      cand[k] = (flag[g0 + k] & (F_UNMAPPED | F_SECONDARY | F_SUPPLEMENTARY)) == 0;
    }
  }
  // Byte tests of a word by ONE add: for bytes below 128 - 113 (every legal quality is) adding 113 / 125 / 34 sets bit 7 of exactly the
  // bytes >= 15 / > 2 / >= 94 without a carry into the next byte; a byte that does carry is >= 94 in the first place and is reported as
  // such (`bad`, checked on the same word).  The words are masked to the block's bytes first, so no test needs the mask again.
  static __device__ __forceinline__ uint32_t part(uint32_t x, uint32_t s) {  // s + the sum of the bytes >= 15
    const uint32_t ge15 = ((x + 0x71717171u) & 0x80808080u) >> 7;
    return __builtin_amdgcn_sad_u8(x & (ge15 * 0xFFu), 0u, s);
  }
  static __device__ __forceinline__ uint32_t gt2(uint32_t x) { return (((x | 0x80808080u) - 0x03030303u) | x) & 0x80808080u; }  // bit 7 of every byte > 2, whatever the bytes (a missing-QUAL filler of 0xFF in a non-candidate must not spill into its neighbour's test)
  static __device__ __forceinline__ uint32_t ge94(uint32_t x) { return x | (x + 0x22222222u); }           // bit 7 of every byte >= 94 (or >= 128)
  struct Pre { Chunk ch; uint32_t rl; int nb; int k0; };
  __device__ __forceinline__ bool prefetch(uint32_t rl, int k0, int nb, uint64_t qpos, uint32_t, Pre &p) {
    p.rl = rl; p.nb = nb; p.k0 = k0;
    p.ch.load(qual + qpos);
    return true;
  }
  __device__ __forceinline__ void process(Pre &p) {
    const uint4 mk = mask[p.nb];
    const uint32_t x0 = p.ch.w0 & mk.x, x1 = p.ch.w1 & mk.y, x2 = p.ch.w2 & mk.z, x3 = p.ch.w3 & mk.w;
    // low-quality-tail bounds: first / last quality > 2 of the read
    const uint64_t glo = (uint64_t)gt2(x0) | ((uint64_t)gt2(x1) << 32);
    const uint64_t ghi = (uint64_t)gt2(x2) | ((uint64_t)gt2(x3) << 32);
    if (glo | ghi) {
      const int first = glo ? (__builtin_ctzll(glo) >> 3) : 8 + (__builtin_ctzll(ghi) >> 3);
      const int last = ghi ? 8 + ((63 - __builtin_clzll(ghi)) >> 3) : ((63 - __builtin_clzll(glo)) >> 3);
      atomicMin(&lo[p.rl], (uint32_t)(p.k0 + first));
      atomicMax(&hi[p.rl], (uint32_t)(p.k0 + last + 1));
    }
    if (!cand[p.rl]) return;
    // a quality >= 94 (or a byte whose carry could have spoilt a neighbour's test) in a duplicate-marking candidate: the error bit
    bad |= (ge94(x0) | ge94(x1) | ge94(x2) | ge94(x3)) & 0x80808080u;
    const uint32_t s = part(x3, part(x2, part(x1, part(x0, 0u))));
    if (s) atomicAdd(&acc[p.rl], (int32_t)s);
  }
  __device__ __forceinline__ void group_end(uint32_t g0, uint32_t ng) {
    for (uint32_t k = threadIdx.x; k < ng; k += NT) {  // a read belongs to exactly one group
      score[g0 + k] = acc[k];
      qbounds[g0 + k] = (uint64_t)hi[k] | ((uint64_t)lo[k] << 32);  // all-zero = no quality > 2
    }
  }
  __device__ __forceinline__ void slots(uint32_t) {}
  __device__ __forceinline__ void retire() {}
  __device__ __forceinline__ void tile_end(uint32_t, uint64_t) {}
};

__global__ __launch_bounds__(FL_THREADS) void k_score_flat(uint64_t n, const uint64_t *__restrict__ qual_off, const uint8_t *__restrict__ qual,
                                                           uint64_t qual_bytes, const uint32_t *__restrict__ tile_first,
                                                           const uint16_t *__restrict__ flag, int32_t *score, uint64_t *qbounds, uint32_t *err) {
  __shared__ FlatLds<ScoreBody::RMAX> L;
  __shared__ int32_t acc[ScoreBody::RMAX];
  __shared__ uint32_t lo[ScoreBody::RMAX], hi[ScoreBody::RMAX];
  __shared__ uint8_t cand[ScoreBody::RMAX];
  __shared__ uint4 mask[17];
  if (threadIdx.x <= 16) {
    const int nb = (int)threadIdx.x;
    mask[nb] = make_uint4(first_bytes32(nb), first_bytes32(nb - 4), first_bytes32(nb - 8), first_bytes32(nb - 12));
  }
  __syncthreads();
  ScoreBody B{flag, qual, score, qbounds, acc, lo, hi, cand, mask, 0u};
  flat_run(qual_off, n, qual_bytes, tile_first, L, B);
  if (__any(B.bad != 0) && (threadIdx.x & 63) == 0) atomicOr(&err[0], 1u);
}

// computePhredScore + the low-quality-tail bounds for read sets of ONE length (what a sequencer writes; ensure_uniform_len): a READ per
// lane.  A wave copies the QUAL bytes of 64 consecutive reads - one contiguous, 64-byte-aligned span - into its own LDS tile with
// coalesced 16-byte loads and every lane then walks the words of its read: the sums, the error test and the positions of the first /
// last quality > 2 stay in the lane's registers.  k_score_flat spreads a read over ~10 lanes and combines them through three LDS
// atomics per 16-byte block on the read's cells (PMC, round 3: 79 % of its LDS cycles were bank conflicts of exactly those).
constexpr int SU_WAVES = 8, SU_MAX_LEN = 250;  // a read touches at most 64 words (the bitmap of words with a quality > 2)
__device__ __forceinline__ uint32_t su_gt2(uint32_t x) { return (((x | 0x80808080u) - 0x03030303u) | x) & 0x80808080u; }  // bit 7 of every byte > 2, any byte value
template <int R>
__global__ __launch_bounds__(64 * SU_WAVES) void k_score_uniform(uint64_t n, uint32_t L, const uint8_t *__restrict__ qual, const uint16_t *__restrict__ flag,
                                                                 int32_t *__restrict__ score, uint64_t *__restrict__ qbounds, uint32_t *err,
                                                                 uint32_t qstride /* every qstride-th group of 64 reads notes the quality values it holds */,
                                                                 unsigned long long *qmask, const uint16_t *__restrict__ rgid, const uint16_t *__restrict__ rg_cov,
                                                                 uint2 *__restrict__ arecs /* not null: ApplyBQSR's per-read records (apply_rec.hpp) */) {
  extern __shared__ __attribute__((aligned(16))) uint32_t su_lds[];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t tile_bytes = 64u * L, tile_words = tile_bytes / 4u + 4u;
  uint32_t *tile = su_lds + (size_t)wave * tile_words;
  const uint32_t s = lane * L, w0 = s >> 2, skip = s & 3u;  // the read's first word and the bytes of it in front of the read
  const uint32_t nw = (skip + L + 3u) >> 2, tail = (skip + L) & 3u;
  const uint32_t m_first = 0xFFFFFFFFu << (8u * skip), m_last = tail ? 0xFFFFFFFFu >> (32u - 8u * tail) : 0xFFFFFFFFu;
  uint32_t bad = 0;
  uint32_t qm[3] = {0, 0, 0};  // bits 0 .. 95: quality values seen by the sampled groups (the BQSR gather's sizing hint, ensure_qual_present)
  const uint64_t ngroups = (n + 63) / 64;
  for (uint64_t g = (uint64_t)blockIdx.x * SU_WAVES + wave; g < ngroups; g += (uint64_t)gridDim.x * SU_WAVES) {
    const uint64_t r0 = g * 64;
    const uint8_t *base = qual + r0 * L;
    const uint64_t have = (n - r0 < 64 ? n - r0 : 64) * (uint64_t)L;  // the last group may be short
    uint4 v[R];
#pragma unroll
    for (int k = 0; k < R; k++) {
      const uint32_t off = (uint32_t)k * 1024u + lane * 16u;
      v[k] = make_uint4(0u, 0u, 0u, 0u);
      if (off < have) v[k] = *reinterpret_cast<const uint4 *>(base + off);  // (the column is padded: the last chunk may read past `have`)
    }
#pragma unroll
    for (int k = 0; k < R; k++) {
      const uint32_t off = (uint32_t)k * 1024u + lane * 16u;
      if (off < tile_bytes) *reinterpret_cast<uint4 *>(tile + off / 4u) = v[k];
    }
    // (the tile is the wave's own, and a wave's LDS operations execute in order: no barrier)
    const uint64_t r = r0 + lane;
    if (r < n) {
      const uint16_t f = flag[r];
      const bool cand = (f & (F_UNMAPPED | F_SECONDARY | F_SUPPLEMENTARY)) == 0;
      const uint16_t rg = arecs ? rgid[r] : (uint16_t)ELP_NIL16;
      const uint32_t *rw = tile + w0;
      uint32_t x = rw[0] & m_first;
      if (nw == 1) x &= m_last;
      // per word: the sum of the bytes >= 15 (six operations), the OR of all words (the > 93 test is made ONCE on it: a byte of the OR is
      // at least the byte of every word, so a clean OR clears the read - legal qualities stay below 64, whose ORs stay below 94 - and only a
      // flagged OR is looked at word by word), and the first / last word that holds a quality > 2 as two running indices (round 6: five
      // vector instructions fewer per word than a 64-bit bitmap of such words and a > 93 test per word; measured on the same box: no change
      // in the kernel's time - 73 % vector issue by PMC, but the LDS tile's round trip is what a wave waits for)
      constexpr uint32_t NONE = 0xFFFFFFFFu;
      uint32_t sum = ScoreBody::part(x, 0u), orx = x;
      uint32_t kf = su_gt2(x) ? 0u : NONE, kl = 0u;
#pragma unroll 8
      for (uint32_t k = 1; k + 1 < nw; k++) {
        x = rw[k];
        sum = ScoreBody::part(x, sum);
        orx |= x;
        const bool gk = su_gt2(x) != 0u;
        kf = min(kf, gk ? k : NONE);
        kl = gk ? k : kl;
      }
      if (nw > 1) {
        x = rw[nw - 1] & m_last;
        sum = ScoreBody::part(x, sum);
        orx |= x;
        const bool gk = su_gt2(x) != 0u;
        kf = min(kf, gk ? nw - 1 : NONE);
        kl = gk ? nw - 1 : kl;
      }
      uint32_t bd = ScoreBody::ge94(orx) & 0x80808080u;
      if (bd) {  // (rare) exactly, word by word
        bd = 0;
        for (uint32_t k = 0; k < nw; k++) {
          uint32_t xw = rw[k];
          if (k == 0) xw &= m_first;
          if (k == nw - 1) xw &= m_last;
          bd |= ScoreBody::ge94(xw);
        }
      }
      uint32_t lo = 0xFFFFFFFFu, hi = 0u;
      if (kf != NONE) {
        uint32_t xf = rw[kf], xl = rw[kl];
        if (kf == 0) xf &= m_first;
        if (kf == nw - 1) xf &= m_last;
        if (kl == 0) xl &= m_first;
        if (kl == nw - 1) xl &= m_last;
        lo = kf * 4u + ((uint32_t)__builtin_ctz(su_gt2(xf)) >> 3) - skip;
        hi = kl * 4u + ((31u - (uint32_t)__builtin_clz(su_gt2(xl))) >> 3) - skip + 1u;
      }
      score[r] = cand ? (int32_t)sum : 0;
      qbounds[r] = (uint64_t)hi | ((uint64_t)lo << 32);
      if (cand) bad |= bd & 0x80808080u;
      if (arecs) arecs[r] = rg == ELP_NIL16 ? make_uint2(AR_NO_RG, 0u) : apply_record(L, (int)L, f, (uint64_t)hi | ((uint64_t)lo << 32), rg_cov[rg]);
      if ((uint32_t)g % qstride == 0) {  // (wave-uniform)
        for (uint32_t k = 0; k < nw; k++) {
          const uint32_t vm = (k == 0 ? m_first : 0xFFFFFFFFu) & (k + 1 == nw ? m_last : 0xFFFFFFFFu), xw = rw[k];
#pragma unroll
          for (int b = 0; b < 4; b++) {
            const uint32_t q = (xw >> (8 * b)) & 0xFFu, bit = ((vm >> (8 * b)) & 1u) << (q & 31u), ws = q >> 5;
            qm[0] |= ws == 0 ? bit : 0u;
            qm[1] |= ws == 1 ? bit : 0u;
            qm[2] |= ws == 2 ? bit : 0u;
          }
        }
      }
    }
  }
  if (__any(bad != 0) && lane == 0) atomicOr(&err[0], 1u);
  if (__any((qm[0] | qm[1] | qm[2]) != 0)) {
    for (int d = 32; d >= 1; d >>= 1) {
      qm[0] |= __shfl_xor(qm[0], d, 64);
      qm[1] |= __shfl_xor(qm[1], d, 64);
      qm[2] |= __shfl_xor(qm[2], d, 64);
    }
    if (lane == 0) {
      const unsigned long long lo = (unsigned long long)qm[0] | ((unsigned long long)qm[1] << 32);
      if (lo) atomicOr(&qmask[0], lo);
      if (qm[2]) atomicOr(&qmask[1], (unsigned long long)qm[2]);
    }
  }
}
// LDS bank conflicts of the per-lane walk: lanes are L bytes apart; if that is a whole number of words with a large power of two in it,
// many lanes sit on one bank - those lengths stay with k_score_flat
static bool score_uniform_ok(uint32_t L) {
  if (L < 16 || L > (uint32_t)SU_MAX_LEN) return false;
  if (L % 4u) return true;
  return ((L / 4u) % 4u) != 0;  // at most two lanes more per bank than the 64-lanes-on-32-banks minimum
}
template <int R>
static int score_uniform_launch(elp_ctx *c) {
  const uint32_t L = c->uniform_len;
  const size_t dyn = (size_t)SU_WAVES * ((size_t)64 * L + 16);
  const unsigned per_cu = (unsigned)std::max<size_t>(1, (160 * 1024) / (dyn + 256));
  const uint64_t ngroups = (c->n + 63) / 64;
  const unsigned grid = (unsigned)std::min<uint64_t>((ngroups + SU_WAVES - 1) / SU_WAVES, (uint64_t)c->n_cu * per_cu);
  ELP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_score_uniform<R>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
  // ApplyBQSR's records on the way (apply3.hip takes whole 16-byte blocks of reads of one length)
  uint2 *arecs = nullptr;
  if (L >= 16 && L <= 0x7FFF && c->n_rg > 0) {
    ELP_TRY(ensure(c, c->apply_recs, c->n + 4));
    arecs = c->apply_recs.p;
  }
  // the sample of the quality values: everything up to ~500 K reads (the hint is then exact, as k_qual_present_sample's is), one group in
  // up to 128 of longer columns
  const uint32_t qstride = (uint32_t)std::min<uint64_t>(128, std::max<uint64_t>(1, ngroups / 8192));
  ELP_LAUNCH(c, "adapt_score", k_score_uniform<R>, dim3(grid), dim3(64 * SU_WAVES), dyn, c->n, L, (const uint8_t *)c->qual.p, (const uint16_t *)c->flag.p,
             c->score.p, c->qbounds.p, c->adapt_err.p, qstride, reinterpret_cast<unsigned long long *>(c->adapt_err.p + 2), (const uint16_t *)c->rgid.p,
             (const uint16_t *)c->rg_cov.p, arecs);
  c->adapt_sampled = true;
  c->apply_recs_valid = arecs != nullptr;
  c->apply_recs_lmax = (int)L;
  return 0;
}
static int score_uniform(elp_ctx *c) {
  switch ((c->uniform_len + 15u) / 16u) {  // 16-byte load rounds of a wave's tile
#define ELP_SU(R) case R: return score_uniform_launch<R>(c);
    ELP_SU(1) ELP_SU(2) ELP_SU(3) ELP_SU(4) ELP_SU(5) ELP_SU(6) ELP_SU(7) ELP_SU(8) ELP_SU(9) ELP_SU(10) ELP_SU(11) ELP_SU(12) ELP_SU(13) ELP_SU(14)
    ELP_SU(15) ELP_SU(16)
#undef ELP_SU
  }
  return set_error(c, ELP_ERR_ARG, "score_uniform: read length %u", c->uniform_len);
}

// Set of quality values present, from a sample of the tiles (every `stride`-th).  It is a sizing hint for the BQSR gather's
// LDS tables, not a correctness input: k_bqsr_count reports any counted quality it has no table slot for and the host retries.
// `span`: bytes looked at per sampled tile (the exact set after a miss: stride 1, whole tiles)
__global__ __launch_bounds__(256) void k_qual_present_sample(const uint8_t *__restrict__ qual, uint64_t qual_bytes, uint64_t stride, uint64_t span,
                                                             unsigned long long *qmask) {
  const uint64_t ntiles = (qual_bytes + FL_TILE - 1) / FL_TILE;
  uint32_t m[3] = {0, 0, 0};  // bits 0..95
  for (uint64_t t = (uint64_t)blockIdx.x * stride; t < ntiles; t += (uint64_t)gridDim.x * stride) {
    const uint64_t tb = t * FL_TILE, te = (tb + span < qual_bytes) ? tb + span : qual_bytes;
    for (uint64_t p = tb + (uint64_t)threadIdx.x * 16; p < te; p += 256 * 16) {
      Chunk ch;
      ch.load(qual + p);
      const int nv = (int)((te - p) < 16 ? (te - p) : 16);
#pragma unroll
      for (int i = 0; i < 16; i++) {
        const uint32_t wv = (i >> 2) == 0 ? ch.w0 : ((i >> 2) == 1 ? ch.w1 : ((i >> 2) == 2 ? ch.w2 : ch.w3));
        const uint32_t q = (wv >> (8 * (i & 3))) & 0xFFu;
        const uint32_t bit = (i < nv) ? (1u << (q & 31u)) : 0u;
        const uint32_t wsel = q >> 5;
        m[0] |= wsel == 0 ? bit : 0u;
        m[1] |= wsel == 1 ? bit : 0u;
        m[2] |= wsel == 2 ? bit : 0u;
      }
    }
  }
  for (int d = 32; d >= 1; d >>= 1) {
    m[0] |= __shfl_xor(m[0], d, 64);
    m[1] |= __shfl_xor(m[1], d, 64);
    m[2] |= __shfl_xor(m[2], d, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    const unsigned long long lo = (unsigned long long)m[0] | ((unsigned long long)m[1] << 32);
    if (lo) atomicOr(&qmask[0], lo);
    if (m[2]) atomicOr(&qmask[1], (unsigned long long)m[2]);
  }
}

// tile_first[t] = first read r (0 <= r <= n) with qual_off[r] >= t * FL_TILE, for t in [0, ntiles); tile_first[ntiles] = n
__global__ __launch_bounds__(256) void k_flat_index(const uint64_t *__restrict__ qual_off, uint64_t n_reads, uint64_t ntiles,
                                                    uint32_t *__restrict__ tile_first) {
  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t > ntiles) return;
  if (t == ntiles) { tile_first[t] = (uint32_t)n_reads; return; }
  const uint64_t x = t * FL_TILE;
  uint64_t lo = 0, hi = n_reads;  // answer in [lo, hi]: qual_off[n_reads] = qual_bytes > x
  while (lo < hi) {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (qual_off[mid] >= x) hi = mid; else lo = mid + 1;
  }
  tile_first[t] = (uint32_t)lo;
}

int ensure_flat_index(elp_ctx *c) {
  if (c->flat_index_n == c->n && c->flat_index_bytes == c->qual_bytes && c->n) return 0;
  const uint64_t ntiles = (c->qual_bytes + FL_TILE - 1) / FL_TILE;
  ELP_TRY(ensure(c, c->tile_first, ntiles + 2));
  if (c->n)
    ELP_LAUNCH(c, "flat_index", k_flat_index, dim3(blocks_for(ntiles + 1, 256)), dim3(256), 0, (const uint64_t *)c->qual_off.p, c->n, ntiles,
               c->tile_first.p);
  c->flat_index_n = c->n;
  c->flat_index_bytes = c->qual_bytes;
  return 0;
}

// Do all staged reads have one length?  (every QUAL offset i * len, every SEQ offset SEQ_FRONT + i * ((len + 1) / 2)): then the per-base
// kernels need no offsets at all (count3.hip).  A fact of the staged columns: checked once per staging, like the tile index.
__global__ __launch_bounds__(256) void k_uniform_check(uint64_t n, const uint64_t *__restrict__ qual_off, const uint64_t *__restrict__ seq_off,
                                                       const uint32_t *__restrict__ l_seq, uint64_t len, uint64_t sb, uint64_t seq_front, uint32_t *bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool b = false;
  if (i < n) b = qual_off[i] != i * len || seq_off[i] != seq_front + i * sb || (uint64_t)l_seq[i] != len;
  if (i == n) b = qual_off[n] != n * len || seq_off[n] != seq_front + n * sb;
  if (__any(b) && (threadIdx.x & 63) == 0) atomicOr(bad, 1u);
}
int ensure_uniform_len(elp_ctx *c) {
  if (c->uniform_n == c->n && c->uniform_bytes == c->qual_bytes && c->n) return 0;
  c->uniform_len = 0;
  c->uniform_n = c->n;
  c->uniform_bytes = c->qual_bytes;
  if (!c->n || c->qual_bytes % c->n) return 0;
  const uint64_t len = c->qual_bytes / c->n;
  if (len == 0 || len > 0x3FFFFFull) return 0;
  uint32_t *bad = c->err_flag.p + 2;
  ELP_HIP(c, hipMemsetAsync(bad, 0, 4, c->stream));
  ELP_LAUNCH(c, "uniform_check", k_uniform_check, dim3(blocks_for(c->n + 1, 256)), dim3(256), 0, c->n, (const uint64_t *)c->qual_off.p,
             (const uint64_t *)c->seq_off.p, (const uint32_t *)c->l_seq.p, len, (len + 1) / 2, (uint64_t)elp_ctx::SEQ_FRONT, bad);
  uint32_t hb = 0;
  ELP_HIP(c, hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, c->stream));
  ELP_HIP(c, elp::stream_wait(c->stream));
  ELP_HIP(c, hipMemsetAsync(bad, 0, 4, c->stream));
  if (!hb) c->uniform_len = (uint32_t)len;
  return 0;
}

int adapt_quality_error(elp_ctx *c) {
  return set_error(c, ELP_ERR_DATA, "Invalid QUAL character (phred > 93) in a duplicate-marking candidate (reference: log.Panic, filters/mark-duplicates.go:64-66)");
}

// check_quals: report a quality > 93 in a duplicate-marking candidate (computePhredScore panics); the BQSR entry points, which
// only need the per-read low-quality-tail bounds, pass false and leave that error to a later elp_mark_duplicates
// the quality-error word of the score kernel, read when somebody needs it (check_quals) - or by elp_mark_duplicates together with its
// own first read-back (adapt_note): the adapt stage has no synchronisation of its own
static int adapt_resolve(elp_ctx *c) {
  if (!c->adapt_pending) return 0;
  uint32_t w[ADAPT_WORDS];
  ELP_HIP(c, hipMemcpyAsync(w, c->adapt_err.p, sizeof w, hipMemcpyDeviceToHost, c->stream));
  ELP_HIP(c, elp::stream_wait(c->stream));
  adapt_note(c, w);
  return 0;
}
// words: [0] the score kernel's error word, [2 .. 5] the quality values its sampled groups saw (k_score_uniform only)
void adapt_note(elp_ctx *c, const uint32_t *words) {
  c->adapt_pending = false;
  if (words[0] & 1u) c->adapt_bad_qual = true;
  if (c->adapt_sampled) {
    c->adapt_qmask[0] = (unsigned long long)words[2] | ((unsigned long long)words[3] << 32);
    c->adapt_qmask[1] = (unsigned long long)words[4] | ((unsigned long long)words[5] << 32);
    c->adapt_qmask_valid = true;
  }
}
int adapt_begin(elp_ctx *c, int *pos_bits_out) {
  ELP_HIP(c, hipSetDevice(c->device));
  const uint64_t n = c->n;
  ELP_TRY(ensure_flat_index(c));
  ELP_TRY(ensure(c, c->upos, n + 1));
  ELP_TRY(ensure(c, c->score, n + 1));
  ELP_TRY(ensure(c, c->key, n + 1));
  ELP_TRY(ensure(c, c->qbounds, n + 1));
  ELP_TRY(ensure(c, c->adapt_err, ADAPT_WORDS + 2));
  ELP_HIP(c, hipMemsetAsync(c->adapt_err.p, 0, ADAPT_WORDS * sizeof(uint32_t), c->stream));
  c->adapt_bad_qual = false;
  c->adapt_pending = false;
  c->adapt_sampled = c->adapt_qmask_valid = false;
  c->apply_recs_valid = false;
  c->adapt_epoch++;  // (the key column is about to be rewritten: sorted words made from it are stale)
  int pos_bits = 1;
  while (pos_bits < 32 && (c->max_pos >> pos_bits) != 0) pos_bits++;
  int ref_bits = 1;  // contig codes 0 .. n_ref + 1 (unmapped, then the records that are not sorted at all)
  while (ref_bits < 32 && (((uint32_t)c->n_ref + 1u) >> ref_bits) != 0) ref_bits++;
  c->key_bits = ref_bits + pos_bits + 1;
  *pos_bits_out = pos_bits;
  return 0;
}
// every record's score and low-quality-tail bounds (records without QUAL bytes: zero)
int adapt_scores(elp_ctx *c) {
  const uint64_t n = c->n;
  if (!n) return 0;
  if (!c->qual_bytes) {  // no QUAL bytes at all: no tile, no kernel
    ELP_HIP(c, hipMemsetAsync(c->qbounds.p, 0, n * sizeof(uint64_t), c->stream));
    ELP_HIP(c, hipMemsetAsync(c->score.p, 0, n * sizeof(int32_t), c->stream));
    return 0;
  }
  ELP_TRY(ensure_uniform_len(c));
  if (c->uniform_len && score_uniform_ok(c->uniform_len) && c->tune.score_kernel != 1) {
    ELP_TRY(score_uniform(c));  // (writes the score of every record)
  } else {
    ELP_HIP(c, hipMemsetAsync(c->score.p, 0, n * sizeof(int32_t), c->stream));  // (a read without bases belongs to no tile's group)
    const unsigned grid = (unsigned)std::min<uint64_t>(flat_steps<ScoreBody>(c->qual_bytes), (uint64_t)c->n_cu * 4);
    ELP_LAUNCH(c, "adapt_score", k_score_flat, dim3(grid), dim3(FL_THREADS), 0, n, (const uint64_t *)c->qual_off.p, (const uint8_t *)c->qual.p,
               c->qual_bytes, (const uint32_t *)c->tile_first.p, (const uint16_t *)c->flag.p, c->score.p, c->qbounds.p, c->adapt_err.p);
  }
  c->adapt_pending = true;
  return 0;
}
int ensure_adapted(elp_ctx *c, bool check_quals) {
  if (c->adapted) {
    if (check_quals) ELP_TRY(adapt_resolve(c));
    return (check_quals && c->adapt_bad_qual) ? adapt_quality_error(c) : 0;
  }
  int pos_bits = 1;
  ELP_TRY(adapt_begin(c, &pos_bits));
  const uint64_t n = c->n;
  if (n) {
    ELP_LAUNCH(c, "adapt_fixed", k_adapt_fixed, dim3(blocks_for(n, 256)), dim3(256), 0, n, (const int32_t *)c->pos.p, (const int32_t *)c->refid.p,
               (const uint16_t *)c->flag.p, (const uint64_t *)c->cigar_off.p, (const uint32_t *)c->cigar.p, c->upos.p, c->score.p, c->key.p,
               (uint32_t)c->n_ref, pos_bits, (const uint8_t *)c->has_sr.p);
    ELP_TRY(adapt_scores(c));
  }
  c->adapted = true;
  if (check_quals) ELP_TRY(adapt_resolve(c));
  return (check_quals && c->adapt_bad_qual) ? adapt_quality_error(c) : 0;
}

// c->qual_present = quality values seen in a sample of the QUAL column (a sizing hint, see k_qual_present_sample)
int ensure_qual_present(elp_ctx *c, bool exact) {
  if (c->have_qual_present) return 0;
  c->qual_present[0] = c->qual_present[1] = 0;
  // elp_set_tuning "qual_hint" = 1: the hint stays empty, which forces the gather's report-and-retry path (tests)
  if (!exact && c->tune.qual_hint != 1 && c->adapted && c->adapt_sampled) {
    // the score kernel sampled the column as it went (k_score_uniform): no pass, and no wait of its own if the error word is in already
    ELP_TRY(adapt_resolve(c));
    c->qual_present[0] = c->adapt_qmask[0];
    c->qual_present[1] = c->adapt_qmask[1];
  } else if (c->qual_bytes && (exact || c->tune.qual_hint != 1)) {
    unsigned long long *qm;
    ELP_TRY(scratch(c, 6, 4, &qm));
    ELP_HIP(c, hipMemsetAsync(qm, 0, 16, c->stream));
    const uint64_t ntiles = (c->qual_bytes + FL_TILE - 1) / FL_TILE;
    const uint64_t stride = exact ? 1 : std::min<uint64_t>(16, std::max<uint64_t>(1, ntiles / 2048));
    const unsigned grid = (unsigned)std::min<uint64_t>((ntiles + stride - 1) / stride, 2048);
    ELP_LAUNCH(c, "qual_present_sample", k_qual_present_sample, dim3(grid), dim3(256), 0, (const uint8_t *)c->qual.p, c->qual_bytes, stride,
               (exact || stride == 1) ? FL_TILE : (uint64_t)4096, qm);
    ELP_HIP(c, hipMemcpyAsync(c->qual_present, qm, 16, hipMemcpyDeviceToHost, c->stream));
    ELP_HIP(c, elp::stream_wait(c->stream));
  }
  // elp_set_tuning "qual_hint_drop" = q removes one quality from the hint (tests: the kernels' no-slot paths)
  if (!exact && c->tune.qual_hint_drop >= 0) {
    const int q = c->tune.qual_hint_drop;
    if (q < 64) c->qual_present[0] &= ~(1ull << q);
    else if (q < 128) c->qual_present[1] &= ~(1ull << (q - 64));
  }
  c->have_qual_present = true;
  return 0;
}

// ------------------------------------------------------------------ tie-break
struct TieCols {
  const uint64_t *qname_off;
  const uint8_t *qname;
  const uint16_t *flag;
  const uint8_t *mapq;
  const int32_t *next_refid, *pnext, *tlen;
  const uint8_t *rows = nullptr;  // the radix tie-break's comparator strings written out (k_material_rows): rows[m * rw + j] = byte j of member m
  uint32_t rw = 0;
};

// comparator tail of CoordinateLess (:439-472) for two records with equal (refid, pos, strand)
__device__ inline bool tie_less(const TieCols &t, uint32_t a, uint32_t b) {
  const uint32_t la = (uint32_t)(t.qname_off[a + 1] - t.qname_off[a]), lb = (uint32_t)(t.qname_off[b + 1] - t.qname_off[b]);
  if (la != 0 && lb != 0) {
    int c = qname_cmp(t.qname, t.qname_off, a, b);
    if (c < 0) return true;
    if (c > 0) return false;
  }
  const uint16_t fa = mod_flag(t.flag[a]), fb = mod_flag(t.flag[b]);
  if (fa < fb) return true;
  if (fa > fb) return false;
  const uint8_t ma = t.mapq[a], mb = t.mapq[b];
  if (ma < mb) return true;
  if (ma > mb) return false;
  if ((t.flag[a] & F_MULTIPLE) && (t.flag[b] & F_MULTIPLE)) {
    const int32_t na = t.next_refid[a], nb = t.next_refid[b];
    if (na < nb) return true;
    if (na > nb) return false;
    const int32_t pa = t.pnext[a], pb = t.pnext[b];
    if (pa < pb) return true;
    if (pa > pb) return false;
  }
  return t.tlen[a] < t.tlen[b];
}

__global__ __launch_bounds__(256) void k_iota(uint32_t *v, uint64_t n) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (uint32_t)i;
}

// the sorted array: (key, index) pairs in two arrays, or - idx_bits > 0 - words key << idx_bits | index (radix_sort_fused)
struct SortedView {
  const uint64_t *__restrict__ keys;
  const uint32_t *__restrict__ vals;
  uint32_t idx_bits;
  __device__ __forceinline__ uint64_t key(uint64_t i) const { return idx_bits ? keys[i] >> idx_bits : keys[i]; }
  __device__ __forceinline__ uint32_t val(uint64_t i) const { return idx_bits ? (uint32_t)(keys[i] & ((1ull << idx_bits) - 1ull)) : vals[i]; }
};

// Runs of equal primary key.  k_tie_scan (every record, three loads issued together): a record whose neighbours both differ is
// final; members of runs go to a compact list.  k_tie_small (list members only, all lanes busy with the same kind of work): runs of
// <= TIE_SMALL records are ranked by all-pairs comparison, members of longer runs are flagged for the radix tie-break.
constexpr int TS_TILES = 32;  // (16: 12 K workgroups queue at the one list counter for 0.15 ms of the kernel's 0.30)
__global__ __launch_bounds__(256) void k_tie_scan(uint64_t n, SortedView sv, uint32_t *__restrict__ perm_out, uint32_t *__restrict__ list, uint32_t *list_n,
                                                  uint32_t *__restrict__ bounds) {
  if (blockIdx.x == 0 && threadIdx.x < 4) bounds[threadIdx.x] = 0;  // the counters of k_tie_small's lists (it runs behind this kernel)
  // a workgroup handles TS_TILES * 256 consecutive records and collects its run members in LDS: ONE global atomic per workgroup
  // (a global atomic per wave on the single list counter serialises at ~12 ns each: 9 ms for 50 M records, measured)
  __shared__ uint32_t lq[TS_TILES * 256];
  __shared__ uint32_t lcount, gbase;
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
#pragma unroll 1
  for (int tile = 0; tile < TS_TILES; tile++) {
    const uint64_t i = ((uint64_t)blockIdx.x * TS_TILES + (uint64_t)tile) * 256 + threadIdx.x;
    bool in_run = false;
    if (i < n) {
      const uint64_t w = sv.keys[i], wp = i > 0 ? sv.keys[i - 1] : 0ull, wn = i + 1 < n ? sv.keys[i + 1] : 0ull;
      const uint64_t k = w >> sv.idx_bits, kp = i > 0 ? wp >> sv.idx_bits : ~k, kn = i + 1 < n ? wn >> sv.idx_bits : ~k;
      const uint32_t me = sv.idx_bits ? (uint32_t)(w & ((1ull << sv.idx_bits) - 1ull)) : sv.vals[i];
      in_run = kp == k || kn == k;
      if (!in_run) perm_out[i] = me;
    }
    const unsigned long long mask = __ballot(in_run);
    if (mask) {
      const int lane = threadIdx.x & 63, leader = __ffsll((long long)mask) - 1;
      uint32_t base = 0;
      if (lane == leader) base = atomicAdd(&lcount, (uint32_t)__popcll(mask));
      base = __shfl(base, leader, 64);
      if (in_run) lq[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)i;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) gbase = lcount ? atomicAdd(list_n, lcount) : 0u;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < lcount; k += 256) list[gbase + k] = lq[k];
}
__global__ __launch_bounds__(256) void k_tie_small(uint64_t n, SortedView sv, uint32_t *__restrict__ perm_out, uint32_t *__restrict__ bounds /* [0] starts, [1] ends counted; [4 + k] / [4 + cap + k] the positions */,
                                                   uint32_t bounds_cap, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n, TieCols t) {
  const uint64_t j0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j0 >= (uint64_t)*list_n) return;
  const uint64_t i = list[j0];
  const uint64_t k = sv.key(i);
  const uint32_t me = sv.val(i);
  uint64_t s = i, e = i + 1;  // run [s, e)
  bool large = false;
  while (s > 0 && sv.key(s - 1) == k) {
    s--;
    if (i - s >= (uint64_t)TIE_SMALL) { large = true; break; }
  }
  if (!large) {
    while (e < n && sv.key(e) == k) {
      e++;
      if (e - s > (uint64_t)TIE_SMALL) { large = true; break; }
    }
  }
  if (large) {
    // a run of more than TIE_SMALL records goes to the radix tie-break as a RANGE of the sorted array: its first member reports where
    // it starts, its last one where it ends (two short lists the host pairs up) - no flag per record, no scan over all records
    perm_out[i] = me;
    if (i == 0 || sv.key(i - 1) != k) { const uint32_t at = atomicAdd(&bounds[0], 1u); if (at < bounds_cap) bounds[4 + at] = (uint32_t)i; }
    if (i + 1 == n || sv.key(i + 1) != k) { const uint32_t at = atomicAdd(&bounds[1], 1u); if (at < bounds_cap) bounds[4 + bounds_cap + at] = (uint32_t)i; }
    return;
  }
  if (e - s == 2) {
    // a run of two (nine runs in ten): its first member places both with ONE comparison, the second has nothing to do
    if (i != s) return;
    const uint32_t other = sv.val(s + 1);
    const bool swap = tie_less(t, other, me);  // `me` came first: it stays in front unless the other one is strictly less
    perm_out[s] = swap ? other : me;
    perm_out[s + 1] = swap ? me : other;
    return;
  }
  uint32_t rank = 0;
  for (uint64_t j = s; j < e; j++) {
    if (j == i) continue;
    const uint32_t other = sv.val(j);
    // other precedes me iff other < me, or neither is less and other came first (radix sort is stable: j < i <=> earlier staging index)
    if (tie_less(t, other, me)) rank++;
    else if (j < i && !tie_less(t, me, other)) rank++;
  }
  perm_out[s + rank] = me;
}

// members of the large runs, compacted: run r = positions [start[r], start[r] + (first[r + 1] - first[r])) of the sorted array, its members are
// numbers first[r] .. first[r + 1] - 1; u_seg = 1 + r (the key of the last, stable round: runs stay apart and in order)
__global__ __launch_bounds__(256) void k_large_fill(uint32_t nu, uint32_t nr, const uint32_t *__restrict__ start, const uint32_t *__restrict__ first,
                                                    SortedView sv, uint32_t *__restrict__ u_pos, uint32_t *__restrict__ u_read,
                                                    uint32_t *__restrict__ u_seg) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nu) return;
  uint32_t lo = 0, hi = nr;  // the run with first[r] <= j < first[r + 1]
  while (hi - lo > 1) {
    const uint32_t mid = (lo + hi) >> 1;
    if (first[mid] <= j) lo = mid; else hi = mid;
  }
  const uint32_t pos = start[lo] + (j - first[lo]);
  u_pos[j] = pos;
  u_read[j] = sv.val(pos);
  u_seg[j] = lo + 1u;
}

// comparator byte string of record r: QNAME zero-padded to maxq, then modFlag(2, BE), MAPQ(1), NextREFID^msb(4, BE), PNEXT^msb(4, BE)
// (both zero when unpaired; paired-ness is part of modFlag so it is constant within equal prefixes), TLEN^msb(4, BE).
__device__ inline uint32_t material_byte(const TieCols &t, uint32_t r, uint32_t j, uint32_t maxq, uint32_t m = 0) {
  if (t.rows) return t.rows[(size_t)m * t.rw + j];  // m = the member's number
  if (j < maxq) {
    const uint64_t o = t.qname_off[r];
    const uint32_t l = (uint32_t)(t.qname_off[r + 1] - o);
    return j < l ? t.qname[o + j] : 0u;
  }
  const uint32_t f = j - maxq;
  const uint16_t fl = t.flag[r];
  if (f < 2) { const uint16_t m = mod_flag(fl); return f == 0 ? (m >> 8) : (m & 0xFF); }
  if (f == 2) return t.mapq[r];
  if (f < 11) {
    if (!(fl & F_MULTIPLE)) return 0u;
    const uint32_t v = (f < 7 ? (uint32_t)t.next_refid[r] : (uint32_t)t.pnext[r]) ^ 0x80000000u;
    const uint32_t b = (f < 7) ? (f - 3) : (f - 7);
    return (v >> (8 * (3 - b))) & 0xFF;
  }
  if (f < 15) {
    const uint32_t v = (uint32_t)t.tlen[r] ^ 0x80000000u;
    return (v >> (8 * (3 - (f - 11)))) & 0xFF;
  }
  return 0u;
}

// The comparator strings of the large runs' members written out once, `rw` bytes per member (a thread writes eight bytes): the kernels
// of the LSD rounds then read a member's bytes from one or two cache lines instead of chasing QNAME offsets, names and four columns per
// byte (k_material_values and the key kernels took 0.5 ms for the 500 K unmapped reads of the bench workload).
__global__ __launch_bounds__(256) void k_material_rows(uint32_t nu, const uint32_t *__restrict__ u_read, uint32_t maxq, uint32_t nbytes, uint32_t rw,
                                                       uint8_t *__restrict__ rows, TieCols t /* t.rows == nullptr */) {
  const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t per = rw >> 3;
  const uint64_t m = g / per;
  if (m >= nu) return;
  const uint32_t k = (uint32_t)(g - m * per) * 8u, r = u_read[m];
  uint64_t v = 0;
  for (uint32_t i = 0; i < 8; i++)
    if (k + i < nbytes) v |= (uint64_t)material_byte(t, r, k + i, maxq) << (8 * i);
  reinterpret_cast<uint64_t *>(rows + m * rw)[k >> 3] = v;
}

// bit j of live[]: byte j of the comparator string is not the same for all members of large runs (compared with member 0's).
// QNAME bytes are compared eight at a time (zero-padded behind the name's end, as material_byte pads them).
__global__ __launch_bounds__(256) void k_material_live(uint32_t nu, const uint32_t *__restrict__ u_read, uint32_t maxq, uint32_t nbytes,
                                                       uint32_t *live, TieCols t) {
  __shared__ uint32_t acc[elp_ctx::TIE_LIVE_WORDS];
  if (threadIdx.x < elp_ctx::TIE_LIVE_WORDS) acc[threadIdx.x] = 0;
  __syncthreads();
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m < nu && t.rows) {
    // the strings are written out: member m's row against member 0's, eight bytes at a time
    const uint64_t *a8 = reinterpret_cast<const uint64_t *>(t.rows + (size_t)m * t.rw), *b8 = reinterpret_cast<const uint64_t *>(t.rows);
    for (uint32_t k = 0; k < nbytes; k += 8) {
      uint64_t x = a8[k >> 3] ^ b8[k >> 3];
      if (x) {
        x |= x >> 4; x |= x >> 2; x |= x >> 1;
        x &= 0x0101010101010101ull;
        const uint32_t bits = (uint32_t)((x * 0x0102040810204080ull) >> 56);
        for (uint32_t i = 0; i < 8 && k + i < nbytes; i++)
          if ((bits >> i) & 1u) atomicOr(&acc[(k + i) >> 5], 1u << ((k + i) & 31));
      }
    }
  } else if (m < nu) {
    const uint32_t r = u_read[m], r0 = u_read[0];
    const uint64_t oa = t.qname_off[r], ob = t.qname_off[r0];
    const uint32_t la = (uint32_t)(t.qname_off[r + 1] - oa), lb = (uint32_t)(t.qname_off[r0 + 1] - ob);
    for (uint32_t k = 0; k < maxq; k += 8) {
      const uint64_t a = k < la ? low_bytes(load8(t.qname + oa + k), la - k) : 0ull;
      const uint64_t b = k < lb ? low_bytes(load8(t.qname + ob + k), lb - k) : 0ull;
      uint64_t x = a ^ b;
      if (x) {
        x |= x >> 4; x |= x >> 2; x |= x >> 1;  // bit 0 of every byte: the byte is not zero
        x &= 0x0101010101010101ull;
        const uint32_t bits = (uint32_t)((x * 0x0102040810204080ull) >> 56);  // bit i: byte i differs
        for (uint32_t i = 0; i < 8 && k + i < maxq; i++)
          if ((bits >> i) & 1u) atomicOr(&acc[(k + i) >> 5], 1u << ((k + i) & 31));
      }
    }
    for (uint32_t j = maxq; j < nbytes; j++)
      if (material_byte(t, r, j, maxq) != material_byte(t, r0, j, maxq)) atomicOr(&acc[j >> 5], 1u << (j & 31));
  }
  __syncthreads();
  if (threadIdx.x < elp_ctx::TIE_LIVE_WORDS && acc[threadIdx.x]) atomicOr(&live[threadIdx.x], acc[threadIdx.x]);
}

// which byte values occur at every live position (bitmap of 256 bits per position, collected in LDS)
constexpr int TIE_MAX_PACKED = 96;  // live positions up to which the packed keys below are used
struct TiePosList { uint16_t pos[TIE_MAX_PACKED]; uint32_t n; };
__global__ __launch_bounds__(256) void k_material_values(uint32_t nu, const uint32_t *__restrict__ u_read, TiePosList lp, uint32_t maxq,
                                                         uint32_t *vals /* [n][8] */, TieCols t) {
  __shared__ uint32_t acc[TIE_MAX_PACKED * 8];
  for (uint32_t k = threadIdx.x; k < lp.n * 8; k += blockDim.x) acc[k] = 0;
  __syncthreads();
  for (uint32_t m = blockIdx.x * blockDim.x + threadIdx.x; m < nu; m += gridDim.x * blockDim.x) {
    const uint32_t r = u_read[m];
    for (uint32_t k = 0; k < lp.n; k++) {
      const uint32_t v = material_byte(t, r, lp.pos[k], maxq, m);
      const uint32_t bit = 1u << (v & 31u);
      uint32_t *w = &acc[k * 8 + (v >> 5)];
      if (!(*w & bit)) atomicOr(w, bit);
    }
  }
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < lp.n * 8; k += blockDim.x)
    if (acc[k]) atomicOr(&vals[k], acc[k]);
}
// key of a member = the RANKS of its bytes (among the values that occur at the position: same order, fewer bits) at up to 64
// positions, most significant first: pos[j] is shifted to bit `shift[j]`; rank = lut[slot[j] * 256 + byte]
struct TiePacked { uint16_t pos[64]; uint8_t shift[64]; uint8_t slot[64]; uint32_t n; };
__global__ __launch_bounds__(256) void k_material_keys_packed(uint32_t nu, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ u_read,
                                                              TiePacked sel, const uint8_t *__restrict__ lut, uint32_t maxq, uint64_t *__restrict__ keys,
                                                              TieCols t, const uint32_t *__restrict__ u_seg, uint32_t seg_shift) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nu) return;
  const uint32_t mem = vals[j], r = u_read[mem];
  uint64_t k = u_seg ? (uint64_t)u_seg[mem] << seg_shift : 0ull;  // the run id in front: runs stay apart and in order
  for (uint32_t b = 0; b < sel.n; b++) k |= (uint64_t)lut[(uint32_t)sel.slot[b] * 256u + material_byte(t, r, sel.pos[b], maxq, mem)] << sel.shift[b];
  keys[j] = k;
}

// key of a member = the bytes of its comparator string at the (up to eight) positions pos[0] < pos[1] < ..., most significant first
struct TieSel { uint16_t pos[8]; uint32_t n; };
__global__ __launch_bounds__(256) void k_material_keys_sel(uint32_t nu, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ u_read,
                                                           TieSel sel, uint32_t maxq, uint64_t *__restrict__ keys, TieCols t) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nu) return;
  const uint32_t mem = vals[j], r = u_read[mem];
  uint64_t k = 0;
  for (uint32_t b = 0; b < sel.n; b++) k = (k << 8) | material_byte(t, r, sel.pos[b], maxq, mem);
  keys[j] = k;
}

__global__ __launch_bounds__(256) void k_seg_keys(uint32_t nu, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ u_seg_incl,
                                                  uint64_t *__restrict__ keys) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nu) keys[j] = (uint64_t)u_seg_incl[vals[j]];
}


// The comparator strings of two members, all positions (the order the LSD rounds over every live position give).
__device__ inline int material_cmp(const TieCols &t, uint32_t ma, uint32_t ra, uint32_t mb, uint32_t rb, uint32_t nbytes, uint32_t maxq) {
  if (t.rows) {
    const uint64_t *a8 = reinterpret_cast<const uint64_t *>(t.rows + (size_t)ma * t.rw), *b8 = reinterpret_cast<const uint64_t *>(t.rows + (size_t)mb * t.rw);
    for (uint32_t k = 0; k < nbytes; k += 8) {  // bytes behind nbytes are zero in every row
      const uint64_t x = a8[k >> 3], y = b8[k >> 3];
      if (x != y) return __builtin_bswap64(x) < __builtin_bswap64(y) ? -1 : 1;
    }
    return 0;
  }
  for (uint32_t j = 0; j < nbytes; j++) {
    const uint32_t x = material_byte(t, ra, j, maxq), y = material_byte(t, rb, j, maxq);
    if (x != y) return x < y ? -1 : 1;
  }
  return 0;
}
// Members of the large runs sorted on ONE key: the run id and the ranks of the most significant live positions that fit 64 bits
// (read names differ early: nearly every key is unique or shared by the two mates of a pair).  What the key leaves open is settled
// here: a member whose neighbours have other keys is final; the members of a group of equal keys rank themselves by comparing the
// whole strings (equal strings: the earlier member first, as the stable rounds leave them).  A group of more than LT_CAP members
// raises `over`: the host then runs the LSD rounds over all positions instead.
constexpr uint32_t LT_CAP = 1024;
__global__ __launch_bounds__(256) void k_large_ties(uint32_t nu, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ u_read, uint32_t nbytes, uint32_t maxq,
                                                    uint32_t *over, TieCols t) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nu) return;
  const uint64_t k = keys[j];
  const uint32_t me = vals[j];
  const bool eq_prev = j > 0 && keys[j - 1] == k, eq_next = j + 1 < nu && keys[j + 1] == k;
  if (!eq_prev && !eq_next) { vals_out[j] = me; return; }
  uint32_t s = j, e = j + 1;  // group [s, e)
  bool large = false;
  while (s > 0 && keys[s - 1] == k) {
    s--;
    if (j - s >= LT_CAP) { large = true; break; }
  }
  while (!large && e < nu && keys[e] == k) {
    e++;
    if (e - s > LT_CAP) large = true;
  }
  if (large) {
    vals_out[j] = me;
    if (!eq_prev) atomicOr(over, 1u);
    return;
  }
  const uint32_t rme = u_read[me];
  uint32_t rank = 0;
  for (uint32_t i = s; i < e; i++) {
    if (i == j) continue;
    const uint32_t other = vals[i];
    const int c = material_cmp(t, other, u_read[other], me, rme, nbytes, maxq);
    if (c < 0 || (c == 0 && other < me)) rank++;
  }
  vals_out[s + rank] = me;
}

__global__ __launch_bounds__(256) void k_large_scatter(uint32_t nu, const uint32_t *__restrict__ vals_sorted, const uint32_t *__restrict__ u_pos,
                                                       const uint32_t *__restrict__ u_read, uint32_t *__restrict__ perm_out) {
  uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < nu) perm_out[u_pos[j]] = u_read[vals_sorted[j]];
}

static int sort_impl(elp_ctx *c, uint64_t *presorted = nullptr /* the key passes' result, made ahead (sort_presort) */) {
  const uint64_t n = c->n;
  ELP_TRY(ensure_adapted(c, false));
  ELP_TRY(ensure(c, c->perm, n + 1));
  if (n == 0) { c->sorted = true; return 0; }
  uint64_t *kbuf;
  uint32_t *vbuf, *flags;
  ELP_TRY(scratch(c, 0, 2 * n + 8, &kbuf));
  ELP_TRY(scratch(c, 1, 2 * n + 8, &vbuf));
  ELP_TRY(scratch(c, 2, 2 * n + 8, &flags));
  uint64_t *k0 = kbuf, *k1 = kbuf + n;
  uint32_t *v0 = vbuf, *v1 = vbuf + n;
  // the key column holds key_bits live bits (adapt packs it): that many digit passes, no histogram read-back; the first pass reads
  // the column itself and numbers the records as it goes (no copy, no index array)
  uint64_t *ks;
  uint32_t *vs = nullptr;
  // where key << b | index fits one word (a genome's coordinate key has ~31-34 live bits), the passes move words, not pairs
  int idx_bits = 1;
  while (idx_bits < 32 && (n >> idx_bits) != 0) idx_bits++;
  const bool words = c->tune.sort_pairs != 1 && c->key_bits >= 1 && c->key_bits + idx_bits <= 64;
  if (words && presorted) ks = presorted;
  else if (words) ELP_TRY(radix_sort_fused(c, c->key.p, n, c->key_bits, idx_bits, k0, k1, &ks));
  else ELP_TRY(radix_sort_pairs_low(c, k0, v0, k1, v1, n, (c->key_bits + 7) / 8, &ks, &vs, c->key.p, true));
  const SortedView sv{ks, vs, words ? (uint32_t)idx_bits : 0u};
  TieCols t{c->qname_off.p, c->qname.p, c->flag.p, c->mapq.p, c->next_refid.p, c->pnext.p, c->tlen.p};
  // run members -> compact list (the other half of `vbuf` is free: the radix sort left its result in one half); bounds of the runs
  // longer than TIE_SMALL -> two short lists in `flags` ([0], [1] their lengths)
  const uint32_t bounds_cap = (uint32_t)std::min<uint64_t>(n / TIE_SMALL + 16, (2 * n + 8 - 4) / 2);
  uint32_t *bounds = flags;
  {
    uint32_t *list = (vs == v0) ? v1 : v0, *list_n = c->err_flag.p + 3;  // the scan-total mailbox (sorted words: `vbuf` is free altogether)
    ELP_HIP(c, hipMemsetAsync(list_n, 0, 4, c->stream));
    ELP_LAUNCH(c, "tie_scan", k_tie_scan, dim3(blocks_for(n, 256 * TS_TILES)), dim3(256), 0, n, sv, c->perm.p, list, list_n, bounds);
    // sized for the worst case; workgroups beyond the list's end leave at once
    ELP_LAUNCH(c, "tie_small", k_tie_small, dim3(blocks_for(n, 256)), dim3(256), 0, n, sv, c->perm.p, bounds, bounds_cap, (const uint32_t *)list, (const uint32_t *)list_n, t);
    ELP_HIP(c, hipMemsetAsync(list_n, 0, 4, c->stream));
  }
  // ONE read-back: the two counts and the first run bounds (all of them unless there are more than TIE_HEAD runs: then a second copy)
  constexpr uint32_t TIE_HEAD = 256;
  std::vector<uint32_t> hb(4 + 2 * (size_t)TIE_HEAD), starts, ends;
  {
    const uint32_t head = std::min<uint32_t>(TIE_HEAD, bounds_cap);
    ELP_HIP(c, hipMemcpyAsync(hb.data(), bounds, (4 + (size_t)head) * 4, hipMemcpyDeviceToHost, c->stream));
    ELP_HIP(c, hipMemcpyAsync(hb.data() + 4 + TIE_HEAD, bounds + 4 + bounds_cap, (size_t)head * 4, hipMemcpyDeviceToHost, c->stream));
    ELP_HIP(c, elp::stream_wait(c->stream));
  }
  const uint32_t nr = hb[0];
  if (nr != hb[1] || nr > bounds_cap) return set_error(c, ELP_ERR_HIP, "coordinate sort: %u starts and %u ends of long runs (internal)", hb[0], hb[1]);
  uint32_t nu = 0;
  if (nr > 0) {
    starts.resize(nr); ends.resize(nr);
    if (nr <= TIE_HEAD) {
      std::copy(hb.begin() + 4, hb.begin() + 4 + nr, starts.begin());
      std::copy(hb.begin() + 4 + TIE_HEAD, hb.begin() + 4 + TIE_HEAD + nr, ends.begin());
    } else {
      ELP_HIP(c, hipMemcpyAsync(starts.data(), bounds + 4, (size_t)nr * 4, hipMemcpyDeviceToHost, c->stream));
      ELP_HIP(c, hipMemcpyAsync(ends.data(), bounds + 4 + bounds_cap, (size_t)nr * 4, hipMemcpyDeviceToHost, c->stream));
      ELP_HIP(c, elp::stream_wait(c->stream));
    }
    std::sort(starts.begin(), starts.end());
    std::sort(ends.begin(), ends.end());  // runs are disjoint ranges: the r-th start belongs to the r-th end
    std::vector<uint32_t> first(nr + 1);
    uint64_t total = 0;
    for (uint32_t r = 0; r < nr; r++) {
      if (ends[r] < starts[r] || (r + 1 < nr && starts[r + 1] <= ends[r])) return set_error(c, ELP_ERR_HIP, "coordinate sort: long runs overlap (internal)");
      first[r] = (uint32_t)total;
      total += (uint64_t)ends[r] - starts[r] + 1;
    }
    first[nr] = (uint32_t)total;
    nu = (uint32_t)total;
    // compacted large-run members
    // (+ the members' comparator strings written out, if that takes at most 512 MB: else every kernel reads the columns)
    const uint32_t rw = (c->max_qname_len + 15u + 15u) & ~15u;
    const bool use_rows = (uint64_t)nu * rw <= (512ull << 20);
    const size_t u_words = ((size_t)5 * nu + 2 * (size_t)nr + 64 + 3) & ~(size_t)3;
    uint32_t *u;
    ELP_TRY(scratch(c, 3, u_words + (use_rows ? ((size_t)nu * rw + 64) / 4 : 0), &u));
    uint32_t *u_pos = u, *u_read = u + nu, *u_seg = u + 2 * (size_t)nu, *uv0 = u + 3 * (size_t)nu, *uv1 = u + 4 * (size_t)nu, *d_start = u + 5 * (size_t)nu,
             *d_first = d_start + nr;
    uint64_t *uk;
    ELP_TRY(scratch(c, 4, (size_t)2 * nu + 16, &uk));
    uint64_t *uk0 = uk, *uk1 = uk + nu;
    ELP_HIP(c, hipMemcpyAsync(d_start, starts.data(), (size_t)nr * 4, hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipMemcpyAsync(d_first, first.data(), ((size_t)nr + 1) * 4, hipMemcpyHostToDevice, c->stream));
    ELP_LAUNCH(c, "large_fill", k_large_fill, dim3(blocks_for(nu, 256)), dim3(256), 0, nu, nr, (const uint32_t *)d_start, (const uint32_t *)d_first, sv, u_pos, u_read, u_seg);
    ELP_LAUNCH(c, "iota", k_iota, dim3(blocks_for(nu, 256)), dim3(256), 0, uv0, (uint64_t)nu);
    const uint32_t maxq = c->max_qname_len;
    const uint32_t m_bytes = maxq + 15;  // <= MAX_QNAME + 15 <= 32 * TIE_LIVE_WORDS positions (elp_stage enforces the QNAME limit)
    // which positions of the comparator string differ at all among the members: one kernel, one read-back; the LSD rounds then
    // take eight live positions each and run every pass (no histogram read-back per round, no rounds over constant bytes)
    static_assert(elp_ctx::MAX_QNAME + 15 <= 32 * elp_ctx::TIE_LIVE_WORDS, "live-position bitmap too small");
    ELP_TRY(ensure(c, c->tie_live, elp_ctx::TIE_LIVE_WORDS));
    ELP_HIP(c, hipMemsetAsync(c->tie_live.p, 0, elp_ctx::TIE_LIVE_WORDS * sizeof(uint32_t), c->stream));
    if (use_rows) {
      uint8_t *rows = reinterpret_cast<uint8_t *>(u + u_words);
      ELP_LAUNCH(c, "material_rows", k_material_rows, dim3(blocks_for((uint64_t)nu * (rw >> 3), 256)), dim3(256), 0, nu, (const uint32_t *)u_read, maxq, m_bytes, rw,
                 rows, t);
      t.rows = rows;
      t.rw = rw;
    }
    ELP_LAUNCH(c, "material_live", k_material_live, dim3(blocks_for(nu, 256)), dim3(256), 0, nu, (const uint32_t *)u_read, maxq, m_bytes,
               c->tie_live.p, t);
    uint32_t live[elp_ctx::TIE_LIVE_WORDS];
    ELP_HIP(c, hipMemcpyAsync(live, c->tie_live.p, sizeof live, hipMemcpyDeviceToHost, c->stream));
    ELP_HIP(c, elp::stream_wait(c->stream));
    std::vector<uint16_t> lp;
    for (uint32_t j = 0; j < m_bytes; j++)
      if ((live[j >> 5] >> (j & 31)) & 1u) lp.push_back((uint16_t)j);
    uint32_t *vcur = uv0, *vtmp = uv1;
    bool packed_done = false;
    if (!lp.empty() && lp.size() <= (size_t)TIE_MAX_PACKED) {
      // Few distinct byte values per live position (read names: digits and ':'): sort on their ranks instead of the bytes, as
      // many positions per 64-bit key as fit - typically two LSD rounds of ~10 passes instead of five rounds of eight.
      TiePosList pl;
      pl.n = (uint32_t)lp.size();
      for (size_t k = 0; k < lp.size(); k++) pl.pos[k] = lp[k];
      uint32_t *d_vals;
      ELP_TRY(scratch(c, 5, (size_t)TIE_MAX_PACKED * 8 + (size_t)TIE_MAX_PACKED * 64 + 64, &d_vals));
      uint8_t *d_lut = reinterpret_cast<uint8_t *>(d_vals + TIE_MAX_PACKED * 8);
      ELP_HIP(c, hipMemsetAsync(d_vals, 0, (size_t)pl.n * 8 * 4, c->stream));
      ELP_LAUNCH(c, "material_values", k_material_values, dim3(std::min(blocks_for(nu, 256), 1024u)), dim3(256), 0, nu, (const uint32_t *)u_read, pl, maxq, d_vals, t);
      std::vector<uint32_t> hv((size_t)pl.n * 8);
      ELP_HIP(c, hipMemcpyAsync(hv.data(), d_vals, hv.size() * 4, hipMemcpyDeviceToHost, c->stream));
      ELP_HIP(c, elp::stream_wait(c->stream));
      std::vector<uint8_t> lut((size_t)pl.n * 256, 0);
      std::vector<int> bits(pl.n);
      for (uint32_t k = 0; k < pl.n; k++) {
        int rank = 0;
        for (int v = 0; v < 256; v++)
          if ((hv[(size_t)k * 8 + (v >> 5)] >> (v & 31)) & 1u) lut[(size_t)k * 256 + v] = (uint8_t)rank++;
        int b = 1;
        while ((1 << b) < rank) b++;
        bits[k] = b;
      }
      ELP_HIP(c, hipMemcpyAsync(d_lut, lut.data(), lut.size(), hipMemcpyHostToDevice, c->stream));
      // ONE round if everything fits a key, or - round 4 - on the most significant positions that do, with the rest settled by
      // comparison inside the groups of equal keys (k_large_ties); the run id rides in front of the key (no last round on it)
      int segbits = 0;
      while (nr > 1 && (1u << segbits) <= nr) segbits++;  // u_seg = 1 .. nr
      if (c->tune.tie_rounds != 1 && segbits + bits[0] <= 64) {
        TiePacked sel;
        memset(&sel, 0, sizeof sel);
        int total = 0, hi = 0;
        while (hi < (int)pl.n && hi < 64 && segbits + total + bits[hi] <= 64) total += bits[hi++];
        int sh = total;
        for (int k = 0; k < hi; k++) {
          sh -= bits[k];
          sel.pos[sel.n] = pl.pos[k]; sel.shift[sel.n] = (uint8_t)sh; sel.slot[sel.n] = (uint8_t)k;
          sel.n++;
        }
        ELP_LAUNCH(c, "material_keys", k_material_keys_packed, dim3(blocks_for(nu, 256)), dim3(256), 0, nu, (const uint32_t *)vcur, (const uint32_t *)u_read,
                   sel, (const uint8_t *)d_lut, maxq, uk0, t, (const uint32_t *)(nr > 1 ? u_seg : nullptr), (uint32_t)total);
        uint64_t *ko;
        uint32_t *vo;
        ELP_TRY(radix_sort_pairs_low(c, uk0, vcur, uk1, vtmp, nu, (segbits + total + 7) / 8, &ko, &vo));
        if (vo != vcur) { vtmp = vcur; vcur = vo; }
        bool settled = hi == (int)pl.n;  // every live position is in the key: equal keys are equal strings, the stable passes kept their order
        if (!settled) {
          uint32_t *over = c->err_flag.p + 3;  // the scan-total mailbox (zero here)
          ELP_LAUNCH(c, "large_ties", k_large_ties, dim3(blocks_for(nu, 256)), dim3(256), 0, nu, (const uint64_t *)ko, (const uint32_t *)vcur, vtmp,
                     (const uint32_t *)u_read, m_bytes, maxq, over, t);
          std::swap(vcur, vtmp);
          ELP_LAUNCH(c, "large_scatter", k_large_scatter, dim3(blocks_for(nu, 256)), dim3(256), 0, nu, (const uint32_t *)vcur, (const uint32_t *)u_pos,
                     (const uint32_t *)u_read, c->perm.p);
          uint32_t h_over = 0;
          ELP_HIP(c, hipMemcpyAsync(&h_over, over, 4, hipMemcpyDeviceToHost, c->stream));
          ELP_HIP(c, elp::stream_wait(c->stream));
          settled = h_over == 0;
          if (!settled) {  // a group of > LT_CAP equal keys: the rounds over all positions, from the members' first order
            ELP_HIP(c, hipMemsetAsync(over, 0, 4, c->stream));
            ELP_LAUNCH(c, "iota", k_iota, dim3(blocks_for(nu, 256)), dim3(256), 0, uv0, (uint64_t)nu);
            vcur = uv0; vtmp = uv1;
          }
        } else {
          ELP_LAUNCH(c, "large_scatter", k_large_scatter, dim3(blocks_for(nu, 256)), dim3(256), 0, nu, (const uint32_t *)vcur, (const uint32_t *)u_pos,
                     (const uint32_t *)u_read, c->perm.p);
        }
        if (settled) {
          c->radix_check_pending = true;
          c->sorted = true;
          return 0;
        }
      }
      // rounds from the least significant position upwards, each filling at most 64 bits / 64 positions
      for (int hi = (int)pl.n; hi > 0;) {
        TiePacked sel;
        memset(&sel, 0, sizeof sel);
        int total = 0, lo = hi;
        while (lo > 0 && total + bits[lo - 1] <= 64 && hi - lo < 64) { lo--; total += bits[lo]; }
        int sh = total;
        for (int k = lo; k < hi; k++) {
          sh -= bits[k];
          sel.pos[sel.n] = pl.pos[k]; sel.shift[sel.n] = (uint8_t)sh; sel.slot[sel.n] = (uint8_t)k;
          sel.n++;
        }
        ELP_LAUNCH(c, "material_keys", k_material_keys_packed, dim3(blocks_for(nu, 256)), dim3(256), 0, nu, (const uint32_t *)vcur, (const uint32_t *)u_read,
                   sel, (const uint8_t *)d_lut, maxq, uk0, t, (const uint32_t *)nullptr, 0u);
        uint64_t *ko;
        uint32_t *vo;
        ELP_TRY(radix_sort_pairs_low(c, uk0, vcur, uk1, vtmp, nu, (total + 7) / 8, &ko, &vo));
        if (vo != vcur) { vtmp = vcur; vcur = vo; }
        hi = lo;
      }
      packed_done = true;
    }
    for (size_t hi = packed_done ? 0 : lp.size(); hi > 0;) {  // least significant positions first
      const size_t lo = hi >= 8 ? hi - 8 : 0;
      TieSel sel;
      sel.n = (uint32_t)(hi - lo);
      for (size_t b = 0; b < 8; b++) sel.pos[b] = b < sel.n ? lp[lo + b] : 0;
      ELP_LAUNCH(c, "material_keys", k_material_keys_sel, dim3(blocks_for(nu, 256)), dim3(256), 0, nu, (const uint32_t *)vcur,
                 (const uint32_t *)u_read, sel, maxq, uk0, t);
      uint64_t *ko;
      uint32_t *vo;
      ELP_TRY(radix_sort_pairs_low(c, uk0, vcur, uk1, vtmp, nu, (int)sel.n, &ko, &vo));
      if (vo != vcur) { vtmp = vcur; vcur = vo; }
      hi = lo;
    }
    // the last, stable round on the run id keeps the runs apart and in order - with ONE long run (the unmapped block of a file without
    // pile-ups) there is nothing to keep apart
    if (nr > 1) ELP_LAUNCH(c, "seg_keys", k_seg_keys, dim3(blocks_for(nu, 256)), dim3(256), 0, nu, (const uint32_t *)vcur, (const uint32_t *)u_seg, uk0);
    if (nr > 1) {
      uint64_t *ko;
      uint32_t *vo;
      ELP_TRY(radix_sort_pairs(c, uk0, vcur, uk1, vtmp, nu, &ko, &vo));
      if (vo != vcur) { vtmp = vcur; vcur = vo; }
    }
    ELP_LAUNCH(c, "large_scatter", k_large_scatter, dim3(blocks_for(nu, 256)), dim3(256), 0, nu, (const uint32_t *)vcur, (const uint32_t *)u_pos,
               (const uint32_t *)u_read, c->perm.p);
  }
  // (a look-back that timed out leaves a wrong permutation: the bit is read by whoever reads the error words next - the following
  // stage, elp_sync, elp_get_permutation, the emit calls: fetch_err - instead of a synchronisation of its own here)
  c->radix_check_pending = true;
  c->sorted = true;
  return 0;
}

}  // namespace elp

namespace elp {
// The sort on the context's side lane 1 (common.hpp): the shadow context sees the key column, the comparator's columns and the
// permutation's buffer as views for the duration of the call, runs on its own stream behind what the context's stream holds now, and
// before it returns the context's stream is made to wait for whatever the lane still has queued - later stages find the permutation.
static int sort_on_side(elp_ctx *c) {
  ELP_TRY(ensure_adapted(c, false));  // (on the context: the shadow takes the key column as it is)
  ELP_TRY(ensure(c, c->perm, c->n + 1));
  elp_ctx *s = nullptr;
  ELP_TRY(side_lane(c, 1, &s));
  s->n = c->n; s->n_sr = c->n_sr; s->n_filtered = c->n_filtered; s->key_bits = c->key_bits; s->max_qname_len = c->max_qname_len; s->max_pos = c->max_pos;
  s->n_ref = c->n_ref; s->adapted = true; s->adapt_pending = false; s->sorted = false;
  s->key.p = c->key.p; s->flag.p = c->flag.p; s->mapq.p = c->mapq.p; s->next_refid.p = c->next_refid.p; s->pnext.p = c->pnext.p; s->tlen.p = c->tlen.p;
  s->qname_off.p = c->qname_off.p; s->qname.p = c->qname.p; s->has_sr.p = c->has_sr.p;
  s->perm.p = c->perm.p; s->perm.cap = c->perm.cap;
  // the key passes may have been queued ahead (elp_sort_ahead, from inside elp_mark_duplicates): same keys, same length, same lane
  int ib = 1;
  while (ib < 32 && (c->n >> ib) != 0) ib++;
  uint64_t *pre = (c->presort_ks && c->presort_epoch == c->adapt_epoch && c->presort_n == c->n && c->presort_idx_bits == ib && c->tune.sort_pairs != 1) ? c->presort_ks : nullptr;
  c->presort_epoch = ~0ull;  // (used once: the tie-break leaves its marks in the buffer's other half)
  int rc = sort_impl(s, pre);
  s->key.p = nullptr; s->flag.p = nullptr; s->mapq.p = nullptr; s->next_refid.p = nullptr; s->pnext.p = nullptr; s->tlen.p = nullptr;
  s->qname_off.p = nullptr; s->qname.p = nullptr; s->has_sr.p = nullptr;
  s->perm.p = nullptr; s->perm.cap = 0;
  if (rc == 0) rc = radix_check(s);  // (the lane's own error words: nothing of the sort stays unread)
  if (rc != 0) {
    (void)elp::stream_wait(s->stream);
    c->err = s->err;
    return rc;
  }
  ELP_TRY(side_join(c, 1));
  c->sorted = s->sorted;
  return 0;
}
}  // namespace elp

namespace elp {
int sort_presort(elp_ctx *c) {
  const uint64_t n = c->n;
  c->presort_epoch = ~0ull;
  if (!c->sort_ahead || n < 2 || !c->adapted || c->tune.sort_pairs == 1) return 0;
  int idx_bits = 1;
  while (idx_bits < 32 && (n >> idx_bits) != 0) idx_bits++;
  if (c->key_bits < 1 || c->key_bits + idx_bits > 64) return 0;  // (pairs, not words: the sort makes its passes itself)
  elp_ctx *s = nullptr;
  ELP_TRY(side_lane(c, 1, &s));
  s->n = n; s->key_bits = c->key_bits;
  const int tile_saved = s->tune.radix_tile;
  if (!s->tune.radix_tile) s->tune.radix_tile = c->tune.presort_tile >= 1 && c->tune.presort_tile <= 3 ? c->tune.presort_tile : 2;
  uint64_t *kbuf, *ks = nullptr;
  int rc = scratch(s, 0, 2 * n + 8, &kbuf);
  if (rc == 0) rc = radix_sort_fused(s, c->key.p, n, c->key_bits, idx_bits, kbuf, kbuf + n, &ks);
  s->tune.radix_tile = tile_saved;
  if (rc != 0) { c->err = s->err; return rc; }
  c->presort_ks = ks; c->presort_n = n; c->presort_idx_bits = idx_bits; c->presort_epoch = c->adapt_epoch;
  return 0;
}
}  // namespace elp

extern "C" int elp_sort_ahead(elp_ctx *c, int on) {
  if (!c) return ELP_ERR_ARG;
  c->sort_ahead = on != 0;
  if (!on) c->presort_epoch = ~0ull;
  return 0;
}

extern "C" int elp_sort_coordinate(elp_ctx *c) {
  if (!c) return ELP_ERR_ARG;
  ELP_HIP(c, hipSetDevice(c->device));
  c->sorted = false;
  return elp::sort_on_side(c);
}
