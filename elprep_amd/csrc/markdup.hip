// markdup.hip — duplicate marking on the HBM column store.
//
// Reference: filters/mark-duplicates.go (classifyFragment :210-254, classifyPair :329-396, MarkDuplicates :398-445).
// The reference runs three pargo sync.Maps with CAS tournaments whose outcome depends on goroutine interleaving only
// for exact (score, QNAME) ties; this implementation computes the outcome of the single-threaded execution (records
// arrive in staging order), which is a deterministic function of the data:
//
//   fragments  key {lib, refid, unclipped 5' pos, strand}: if the group holds a read of a true pair, every true fragment
//              of the group is a duplicate; otherwise all but the best (score desc, QNAME asc, later arrival) are.
//   mates      key {lib, QNAME}: reads pair up in arrival order (DeleteOrStore toggling, :336-340): of the records r0 < r1 < r2 ...
//              (staging order) that share a key, (r0, r1), (r2, r3), ... are pairs, an odd last one stays alone.
//   pairs      key {lib, refid1, refid2, pos1, pos2, strand1, strand2} with ends ordered as in :347-353: all but the best
//              pair (score sum desc, QNAME asc, later completion) have both reads flagged.
//
// Grouping uses open-addressing tables in HBM whose entries are record indices (keys are compared by dereferencing the
// representative's columns, so there are no fingerprint collisions); group payloads are indexed by the representative.
// All cross-workgroup words are touched with agent-scope atomics; stale plain reads cannot occur because table entries
// only ever change EMPTY -> value and tournaments are re-checked through the CAS return value.
#include "common.hpp"

namespace elp {

constexpr uint32_t EMPTY = 0xFFFFFFFFu;
constexpr unsigned long long PB_EMPTY = ~0ull;  // an empty slot of the LDS tables (k_pair_bucket, k_mate_bucket)

struct MdCols {
  uint64_t n;
  const int32_t *refid;
  const uint16_t *flag_in;  // flags as staged (before this call)
  const uint16_t *rgid;
  const uint16_t *rg_lib;
  const uint16_t *split;    // elp_batch.split: records of different splits never share a key
  const int32_t *upos, *score;
  const uint64_t *qname_off;
  const uint8_t *qname;
};

__device__ __forceinline__ bool is_candidate(uint16_t f) { return (f & (F_UNMAPPED | F_SECONDARY | F_SUPPLEMENTARY)) == 0; }
__device__ __forceinline__ bool is_true_pair(uint16_t f) { return (f & (F_MULTIPLE | F_NEXT_UNMAPPED)) == F_MULTIPLE; }  // :182-184
__device__ __forceinline__ uint16_t lib_of(const MdCols &m, uint32_t i) {
  const uint16_t rg = m.rgid[i];
  return rg == ELP_NIL16 ? (uint16_t)ELP_NIL16 : m.rg_lib[rg];  // addLIBID :142-150
}
__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent64(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------- fragments
// The fragment key of a record in one 16-byte word: {REFID, unclipped 5' position, LIBID << 1 | reversed, split id}.  A probe that lands on
// an occupied slot compares against ONE random 16-byte load instead of four column gathers (the tables are at the random-access
// limit of the memory system, so accesses are what counts).
// the FLAG column the tournaments read: the flags as staged, with records the fused predicates rejected (state 2, filter.hip) made
// non-candidates - in the reference they never reach the MarkDuplicates filter (cmd/filter.go:696-773)
__global__ __launch_bounds__(256) void k_md_flag_in(uint64_t n, const uint16_t *__restrict__ flag, const uint8_t *__restrict__ state,
                                                    uint16_t *__restrict__ flag_in) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag_in[i] = state[i] == 2 ? (uint16_t)(flag[i] | F_SECONDARY) : flag[i];
}

// Also lists the TRUE FRAGMENTS (candidates that are not true pairs: unpaired reads and reads whose mate is unmapped).  They are the
// only records the fragment map can flag (classifyFragment, :210-251: a true pair is never flagged there, it only turns every true
// fragment at its key into a duplicate), so only they need grouping; the true pairs look their key up afterwards.  A workgroup
// handles MK_TILES * 256 records and appends its fragments with one global atomic.
constexpr int MK_TILES = 16;
__global__ __launch_bounds__(256) void k_md_keys(MdCols m, uint4 *__restrict__ fkey, uint32_t *__restrict__ flist, uint32_t *nf) {
  __shared__ uint32_t lq[MK_TILES * 256];
  __shared__ uint32_t lcount, gbase;
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
#pragma unroll 2
  for (int tile = 0; tile < MK_TILES; tile++) {
    const uint64_t i = ((uint64_t)blockIdx.x * MK_TILES + (uint64_t)tile) * 256 + threadIdx.x;
    bool frag = false;
    if (i < m.n) {
      const uint16_t f = m.flag_in[i];
      fkey[i] = make_uint4((uint32_t)m.refid[i], (uint32_t)m.upos[i], ((uint32_t)lib_of(m, (uint32_t)i) << 1) | ((f & F_REVERSED) ? 1u : 0u), (uint32_t)m.split[i]);
      frag = is_candidate(f) && !is_true_pair(f);
    }
    const unsigned long long mask = __ballot(frag);
    if (mask) {
      const int lane = threadIdx.x & 63, leader = __ffsll((long long)mask) - 1;
      uint32_t at = 0;
      if (lane == leader) at = atomicAdd(&lcount, (uint32_t)__popcll(mask));
      at = __shfl(at, leader, 64);
      if (frag) lq[at + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)i;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) gbase = lcount ? atomicAdd(nf, lcount) : 0u;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < lcount; k += 256) flist[gbase + k] = lq[k];
}
__device__ __forceinline__ bool key_eq(const uint4 &a, const uint4 &b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }
__device__ __forceinline__ uint64_t frag_hash(const uint4 &k) {
  const uint64_t h = ((uint64_t)k.x << 32) | k.y;
  return mix64(mix64(h) ^ (((uint64_t)k.w << 32) | (uint64_t)k.z));
}

// generic find-or-insert; returns the representative record of i's group (EMPTY: no free slot within `limit` probes - only
// possible for a table that was sized by an estimate)
template <class Eq>
__device__ __forceinline__ uint32_t find_or_insert(uint32_t *table, uint64_t mask, uint64_t h, uint32_t i, Eq eq, uint64_t limit = ~0ull) {
  uint64_t s = h & mask;
  for (uint64_t probes = 0;; probes++) {
    if (probes > limit) return EMPTY;
    // (the CAS at once, no load in front of it: nine probes in ten land on an empty slot, and a random load + a random atomic cost more
    // than the atomic alone - the tables run at the random-access rate of the memory system)
    const uint32_t cur = atomicCAS(&table[s], EMPTY, i);
    if (cur == EMPTY) return i;
    if (cur == i || eq(cur, i)) return cur;
    s = (s + 1) & mask;
  }
}

// the true fragments (flist) are grouped by key; group payloads (best score | pair bit, winner) are indexed by the representative
__global__ __launch_bounds__(256) void k_frag_insert(MdCols m, const uint4 *__restrict__ fkey, const uint32_t *__restrict__ flist, uint32_t nf,
                                                     uint32_t *table, uint64_t mask, uint32_t *__restrict__ frep, unsigned long long *fbest) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nf) return;
  const uint32_t i = flist[j];
  const uint4 mine = fkey[i];
  const uint32_t rep = find_or_insert(table, mask, frag_hash(mine), i, [&](uint32_t a, uint32_t) { return key_eq(fkey[a], mine); });
  frep[i] = rep;
  atomicMax(&fbest[rep], (unsigned long long)(uint32_t)m.score[i]);
}
__global__ __launch_bounds__(256) void k_frag_init(const uint32_t *__restrict__ flist, uint32_t nf, unsigned long long *__restrict__ fbest,
                                                   uint32_t *__restrict__ fwinner) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nf) return;
  const uint32_t i = flist[j];  // any fragment can become its group's representative
  fbest[i] = 0;
  fwinner[i] = EMPTY;
}
// every true pair looks its own key up (plain loads: the table is final and small - sized by the fragments, not by the records):
// a group of fragments at the key of a pair loses as a whole
// occupancy bits of the fragment table (one word per 32 slots): few hundred KB that stay in every L2, so that the pairs' look-ups
// (94 % of which end at an empty slot) rarely leave it
__global__ __launch_bounds__(256) void k_frag_bits(const uint32_t *__restrict__ table, uint64_t nwords, uint32_t *__restrict__ bits) {
  const uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwords) return;
  const uint4 *t = reinterpret_cast<const uint4 *>(table + 32 * w);
  uint32_t b = 0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint4 v = t[k];
    b |= (v.x != EMPTY ? 1u : 0u) << (4 * k) | (v.y != EMPTY ? 2u : 0u) << (4 * k) | (v.z != EMPTY ? 4u : 0u) << (4 * k) | (v.w != EMPTY ? 8u : 0u) << (4 * k);
  }
  bits[w] = b;
}
// (QNAME asc, later arrival wins) tournament among contenders
__device__ __forceinline__ void tournament(const MdCols &m, uint32_t *winner, uint32_t me) {
  uint32_t w = ld_agent(winner);
  for (;;) {
    if (w == EMPTY) {
      const uint32_t old = atomicCAS(winner, EMPTY, me);
      if (old == EMPTY) return;
      w = old;
      continue;
    }
    const int cq = qname_cmp(m.qname, m.qname_off, me, w);
    const bool better = cq < 0 || (cq == 0 && me > w);  // :232-238: on equal QNAME the later arrival replaces the holder
    if (!better) return;
    const uint32_t old = atomicCAS(winner, w, me);
    if (old == w) return;
    w = old;
  }
}

__global__ __launch_bounds__(256) void k_frag_tie(MdCols m, const uint32_t *__restrict__ flist, uint32_t nf, const uint32_t *__restrict__ frep,
                                                  const unsigned long long *__restrict__ fbest, uint32_t *fwinner) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nf) return;
  const uint32_t i = flist[j];
  const uint32_t rep = frep[i];
  const unsigned long long b = fbest[rep];
  if ((b >> 63) || (unsigned long long)(uint32_t)m.score[i] != b) return;
  tournament(m, &fwinner[rep], (uint32_t)i);
}

__global__ __launch_bounds__(256) void k_frag_flag(MdCols m, const uint32_t *__restrict__ flist, uint32_t nf, const uint32_t *__restrict__ frep,
                                                   const unsigned long long *__restrict__ fbest, const uint32_t *__restrict__ fwinner,
                                                   uint16_t *__restrict__ flag_out) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nf) return;
  const uint32_t i = flist[j];
  const uint32_t rep = frep[i];
  const unsigned long long b = fbest[rep];
  const bool dup = (b >> 63) || (unsigned long long)(uint32_t)m.score[i] < b || fwinner[rep] != (uint32_t)i;
  if (dup) flag_out[i] = (uint16_t)(flag_out[i] | F_DUPLICATE);  // (a fragment's flag_in is its staged flag)
}

// ---------------- pair keys (used by the mate pass as well: it writes the entries of the neighbour pairs)
struct PairKey { uint4 k1, k2; };  // fragment keys of the two ends, ordered (:347-353)
// `second` arrived later than `first`
__device__ __forceinline__ PairKey pair_key(const uint4 &second, const uint4 &first) {
  const int32_t r1 = (int32_t)second.x, r2 = (int32_t)first.x, p1 = (int32_t)second.y, p2 = (int32_t)first.y;
  const bool v1 = second.z & 1u, v2 = first.z & 1u;
  const bool swap = r1 > r2 || (r1 == r2 && (p1 > p2 || (p1 == p2 && v1 && !v2)));
  return swap ? PairKey{first, second} : PairKey{second, first};
}
// same two ends (positions, orientations) in the same library (the library of the first end stands for the pair, :355)
__device__ __forceinline__ bool pair_key_eq(const PairKey &a, const PairKey &b) {
  return a.k1.x == b.k1.x && a.k2.x == b.k2.x && a.k1.y == b.k1.y && a.k2.y == b.k2.y && a.k1.z == b.k1.z && ((a.k2.z ^ b.k2.z) & 1u) == 0 &&
         a.k1.w == b.k1.w;  // mates share their split id
}

__device__ __forceinline__ uint64_t pair_hash(const PairKey &k) {
  uint64_t h = mix64(((uint64_t)k.k1.x << 32) | k.k2.x);
  h = mix64(h ^ (((uint64_t)k.k1.y << 32) | k.k2.y));
  return mix64(h ^ (((uint64_t)k.k1.z << 1) | (k.k2.z & 1u)) ^ ((uint64_t)k.k1.w << 40));
}

// ---------------- mates
__device__ __forceinline__ uint64_t qname_hash(const MdCols &m, uint32_t i) {
  const uint64_t o = m.qname_off[i];
  const uint32_t l = (uint32_t)(m.qname_off[i + 1] - o);
  uint64_t h = 0x9e3779b97f4a7c15ull ^ l;
  for (uint32_t k = 0; k < l; k += 8) h = (h ^ low_bytes(load8(m.qname + o + k), l - k)) * 0xff51afd7ed558ccdull + (h >> 29);
  return mix64(h ^ ((uint64_t)lib_of(m, i) << 48) ^ ((uint64_t)m.split[i] << 24));
}
// {split, library, QNAME} of records a and b.  Library and split come from the records' packed fragment keys (round 6: ONE 16-byte load
// per record instead of READ GROUP -> library and split: two dependent loads and a third)
__device__ __forceinline__ bool mate_key_eq(const MdCols &m, const uint4 *__restrict__ fkey, uint32_t a, const uint4 &kb, uint32_t b) {
  const uint4 ka = fkey[a];
  return (ka.z >> 1) == (kb.z >> 1) && ka.w == kb.w && qname_eq(m.qname, m.qname_off, a, b);
}
__device__ __forceinline__ bool is_mate_candidate(uint16_t f) { return is_candidate(f) && is_true_pair(f); }

// Mate matching reproduces DeleteOrStore toggling in staging order (:336-340).  Three paths:
//  (1) neighbours: a run of exactly two neighbouring candidates with the same {split, library, QNAME} (the order an aligner writes
//      mates in) is a pair WITHOUT touching the hash table - provided no other record shares the key.  Records that are not part of
//      such a run announce their key in a Bloom filter first (k_mate_scan); a neighbour pair whose key hits the filter takes path (2).
//  (2) table: the first record of a key to arrive at its slot becomes the slot's representative, the second claims it with one CAS on
//      mate[representative]; both are the only records of the key, so they are a pair whatever their order.
//  (3) a third record of a key fails that CAS and marks the key's group BIG; the host then lists the members of the big groups,
//      sorts them by (representative, staging index) and pairs them up 0-1, 2-3, ... within every group (k_big_collect, k_big_pair).
// k_mate_scan: code[i] = 0 not a mate candidate, 1 table path, 2 leader / 3 follower of a neighbour pair; hash32 of the leader = the
// high half of its key hash (the bits the table index does not use).
enum : uint8_t { MC_NONE = 0, MC_TABLE = 1, MC_LEAD = 2, MC_FOLLOW = 3, MC_KIND = 3, MC_TABBED = 4 /* k_mate_pairs: went through the mate table */ };
constexpr uint32_t MATE_BIG = 0xFFFFFFFEu;

// The kernel is bound by memory latency (a wave waits for ~40 dependent loads otherwise), so every load of a record and its
// neighbour is issued up front in two rounds - fixed columns and QNAME offsets, then library ids and the first 32 bytes of both
// names - and the tests follow without further loads (names longer than 32 bytes finish in a loop).  Threads 256..258 of the
// 320-thread workgroup make the three neighbour tests across the block's borders the same way.
constexpr int MS_THREADS = 320;
__global__ __launch_bounds__(MS_THREADS) void k_mate_scan(MdCols m, uint8_t *__restrict__ code, uint32_t *__restrict__ hash32, uint32_t *__restrict__ hash_lo, uint32_t *bloom,
                                                          uint32_t bloom_mask, uint32_t *n_table /* records that announce their key (sizes the table) */) {
  const uint64_t base = (uint64_t)blockIdx.x * 256;
  const uint32_t t = threadIdx.x;
  // a = the record this thread tests against its right neighbour; s_join[a - (base - 2)]
  const int64_t a_s = t < 256 ? (int64_t)(base + t) : (t == 256 ? (int64_t)base - 2 : (t == 257 ? (int64_t)base - 1 : (t == 258 ? (int64_t)base + 256 : -1)));
  const bool valid = a_s >= 0 && (uint64_t)a_s + 1 < m.n;          // both a and a + 1 exist
  const bool own = a_s >= 0 && (uint64_t)a_s < m.n;                 // a exists
  const uint64_t a = own ? (uint64_t)a_s : 0, b = valid ? a + 1 : a;
  // round 1
  const uint16_t fa = m.flag_in[a], fb = m.flag_in[b], ra = m.rgid[a], rb = m.rgid[b], sa = m.split[a], sb = m.split[b];
  const uint64_t oa = m.qname_off[a], ob = m.qname_off[b], oe = m.qname_off[b + 1];
  const uint32_t la = (uint32_t)((valid ? ob : oe) - oa), lb = (uint32_t)(oe - ob);
  // round 2
  const uint16_t lia = ra == ELP_NIL16 ? (uint16_t)ELP_NIL16 : m.rg_lib[ra], lib_b = rb == ELP_NIL16 ? (uint16_t)ELP_NIL16 : m.rg_lib[rb];
  const uint8_t *pa = m.qname + oa, *pb = m.qname + ob;
  const uint64_t a0 = load8(pa), a1 = load8(pa + 8), a2 = load8(pa + 16), a3 = load8(pa + 24);
  // the neighbour's name is what the next lane loaded as its own: only the last lane of a wave (and the border threads) load it
  // themselves - the names are the bulk of this kernel's traffic, and every one used to be fetched twice
  uint64_t b0 = __shfl_down(a0, 1, 64), b1 = __shfl_down(a1, 1, 64), b2 = __shfl_down(a2, 1, 64), b3 = __shfl_down(a3, 1, 64);
  if ((t & 63u) == 63u || t >= 256u) { b0 = load8(pb); b1 = load8(pb + 8); b2 = load8(pb + 16); b3 = load8(pb + 24); }
  // first 32 bytes of a's name, zero behind its end (the hash takes them as they are)
  const uint64_t w0 = la > 0 ? low_bytes(a0, la) : 0ull, w1 = la > 8 ? low_bytes(a1, la - 8) : 0ull, w2 = la > 16 ? low_bytes(a2, la - 16) : 0ull,
                 w3 = la > 24 ? low_bytes(a3, la - 24) : 0ull;
  bool join = valid && is_mate_candidate(fa) && is_mate_candidate(fb) && lia == lib_b && sa == sb && la == lb;
  if (join) {
    uint64_t diff = w0 ^ (lb > 0 ? low_bytes(b0, lb) : 0ull);
    diff |= w1 ^ (lb > 8 ? low_bytes(b1, lb - 8) : 0ull);
    diff |= w2 ^ (lb > 16 ? low_bytes(b2, lb - 16) : 0ull);
    diff |= w3 ^ (lb > 24 ? low_bytes(b3, lb - 24) : 0ull);
    join = diff == 0;
    for (uint32_t k = 32; join && k < la; k += 8) join = low_bytes(load8(pa + k), la - k) == low_bytes(load8(pb + k), la - k);
  }
  // s_join[u] = joins(base - 2 + u) for u in [0, 259): every neighbour test of the block is made once
  __shared__ uint8_t s_join[264];
  __shared__ uint32_t s_ntab;
  if (t == 0) s_ntab = 0;
  if (a_s >= 0 || t >= 256) {
    const int64_t u = a_s - ((int64_t)base - 2);
    if (t < 259) s_join[t < 256 ? t + 2 : (t == 256 ? 0 : (t == 257 ? 1 : 258))] = join;
    (void)u;
  }
  __syncthreads();
  uint8_t cd = MC_NONE;
  const uint64_t i = base + t;
  if (t < 256 && i < m.n && is_mate_candidate(fa)) {
    const uint8_t *j = s_join + t + 2;  // j[0] = joins(i)
    const bool nx = j[0], pv = j[-1];
    cd = MC_TABLE;
    if (nx && !pv && !j[1]) cd = MC_LEAD;
    else if (pv && !nx && !j[-2]) cd = MC_FOLLOW;
    if (cd != MC_FOLLOW) {
      // qname_hash(m, i) from the words at hand
      uint64_t h = 0x9e3779b97f4a7c15ull ^ la;
      const uint64_t K = 0xff51afd7ed558ccdull;
      if (la > 0) h = (h ^ w0) * K + (h >> 29);
      if (la > 8) h = (h ^ w1) * K + (h >> 29);
      if (la > 16) h = (h ^ w2) * K + (h >> 29);
      if (la > 24) h = (h ^ w3) * K + (h >> 29);
      for (uint32_t k = 32; k < la; k += 8) h = (h ^ low_bytes(load8(pa + k), la - k)) * K + (h >> 29);
      h = mix64(h ^ ((uint64_t)lia << 48) ^ ((uint64_t)sa << 24));
      const uint32_t hi = (uint32_t)(h >> 32);
      hash32[i] = hi;  // (a follower's key is its leader's: hash32[i - 1])
      if (cd == MC_TABLE) hash_lo[i] = (uint32_t)h;  // (the table's index bits: a neighbour pair that goes there after all computes them again)
      if (cd != MC_LEAD) atomicOr(&bloom[(hi >> 5) & bloom_mask], 1u << (hi & 31u));
    }
  }
  if (t < 256 && i < m.n) code[i] = cd;
  // one global atomic per workgroup at most (and none in aligner order, where nearly every record is part of a neighbour pair)
  const unsigned long long tb = __ballot(cd == MC_TABLE);
  if ((t & 63) == 0 && tb) atomicAdd(&s_ntab, (uint32_t)__popcll(tb));
  __syncthreads();
  // (64 counters, a cache line each: with a few per cent of the records on the table path - the sr-tagged copies of an sfm context, whose
  // mates sit in another split - every workgroup adds here, and ~200 K adds to ONE word serialise at ~12 ns each: 2 ms, measured)
  if (t == 0 && s_ntab) atomicAdd(&n_table[(blockIdx.x & 63u) * 16u], s_ntab);
}

// one bit per 64-byte line of the Bloom filter: set if any announcement landed in the line
__global__ __launch_bounds__(256) void k_bloom_coarse(const uint32_t *__restrict__ bloom, uint32_t n_lines, uint32_t *__restrict__ coarse) {
  const uint32_t line = blockIdx.x * 256 + threadIdx.x;  // n_lines is a multiple of 64
  bool nz = false;
  if (line < n_lines) {
    const uint4 *l = reinterpret_cast<const uint4 *>(bloom + (size_t)line * 16);
    const uint4 a = l[0], b = l[1], c2 = l[2], d = l[3];
    nz = (a.x | a.y | a.z | a.w | b.x | b.y | b.z | b.w | c2.x | c2.y | c2.z | c2.w | d.x | d.y | d.z | d.w) != 0;
  }
  const unsigned long long bits = __ballot(nz);
  if ((threadIdx.x & 63) == 0 && line < n_lines) {
    coarse[line >> 5] = (uint32_t)bits;
    coarse[(line >> 5) + 1] = (uint32_t)(bits >> 32);
  }
}

// path (2) / (3) of the mate matching for record i: the first record of a key at its table slot becomes the representative, the second
// claims it with one CAS on mate[representative], a third marks the group BIG
// h = the record's key hash as k_md_front / k_mate_scan stored it (round 6: it was computed again here - the record's name, its offsets,
// read group and split, five random sectors per record of input whose mates are not neighbours)
__device__ __forceinline__ void mate_table_insert(const MdCols &m, const uint4 *__restrict__ fkey, uint32_t i, uint64_t h, uint32_t *table, uint64_t mask, uint32_t *mate,
                                                  uint32_t *rep_of, uint32_t *err) {
  const uint4 mine = fkey[i];
  const uint32_t rep = find_or_insert(table, mask, h, i, [&](uint32_t a, uint32_t b) { return mate_key_eq(m, fkey, a, mine, b); }, mask);
  if (rep == EMPTY) atomicOr(&err[1], 2u);  // the estimated table is full: the host repeats the pass with the full-size one
  else if (rep != i) {                      // (the first of its key at the slot waits for the second)
    rep_of[i] = rep;
    const uint32_t old = atomicCAS(&mate[rep], EMPTY, i);
    if (old == EMPTY) mate[i] = rep;
    else {
      rep_of[rep] = MATE_BIG;  // more than two records share {split, library, QNAME}
      atomicOr(&err[1], 1u);
    }
  }
}
// the table inserts k_mate_pairs listed (64 lists of `cap` entries, their lengths 16 words apart), every lane busy: thread g takes entry
// g of the lists laid end to end (a prefix sum of the 64 lengths per workgroup)
__global__ __launch_bounds__(256) void k_mate_table(MdCols m, const uint4 *__restrict__ fkey, const uint8_t *__restrict__ code, const uint32_t *__restrict__ hash32,
                                                    const uint32_t *__restrict__ hash_lo, const uint32_t *__restrict__ tab_list, const uint32_t *__restrict__ tab_cnt,
                                                    uint32_t cap, uint32_t *table, uint64_t mask, uint32_t *mate, uint32_t *rep_of, uint32_t *err) {
  __shared__ uint32_t first[65];
  if (threadIdx.x < 64) {
    const uint32_t c = tab_cnt[threadIdx.x * 16u];
    uint32_t incl = c;  // inclusive prefix over the wave's 64 lanes
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(incl, d, 64);
      if ((int)threadIdx.x >= d) incl += up;
    }
    first[threadIdx.x + 1] = incl;
    if (threadIdx.x == 0) first[0] = 0;
  }
  __syncthreads();
  // (the grid is sized by the host's estimate of the lists' lengths; whatever they hold is walked)
  for (uint32_t g = blockIdx.x * 256u + threadIdx.x; g < first[64]; g += gridDim.x * 256u) {
    uint32_t lo = 0, hi = 64;  // the list with first[l] <= g < first[l + 1]
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (first[mid] <= g) lo = mid; else hi = mid;
    }
    const uint32_t i = tab_list[(size_t)lo * cap + (g - first[lo])];
    const uint64_t h = (code[i] & MC_KIND) == MC_TABLE ? ((uint64_t)hash32[i] << 32) | hash_lo[i] : qname_hash(m, i);  // (stored for the records that were table-bound from the start)
    mate_table_insert(m, fkey, i, h, table, mask, mate, rep_of, err);
  }
}

// k_mate_pairs - ONE pass over the records behind k_mate_scan does what three passes did (md_mate_insert, md_frag_probe,
// md_pair_list):
//  * every true pair looks its own fragment key up in the (final, small) table of the true fragments - plain loads behind the
//    L2-resident occupancy bits - and sets the pair bit of the group it finds (classifyFragment :210-251: a fragment group that
//    holds a read of a true pair loses as a whole);
//  * mates: a neighbour pair nobody else announced is a pair at once; the other mate candidates go through the table (paths 2 and 3
//    above) and are marked in the code column (MC_TABBED);
//  * the follower of a neighbour pair is the pair's owner (the later arrival, :336-340) and has both keys and both scores at hand: it
//    writes the pair's entry {score sum | hash bits, owner} for the pair phase.  Pairs that form in the table are entered by
//    k_pair_list_table afterwards.
// mate[] and rep_of[] must be EMPTY-initialised.  rep_of[i] = representative of i's key for records that went through the table and
// are not the representative themselves; rep_of[representative] = MATE_BIG iff its key has more than two records.
// `fixed`: the entry of the pair owned by record i goes to slot i >> 1 of the list (two neighbouring followers cannot both own a
// pair, so the slot is the owner's alone); a slot nobody owns gets a hole {score 0xFFFFFFFF | scattered hash bits, EMPTY} that the
// bucket kernel skips - no list counter, no staging, a record per thread.  Holes are ~5 % of the slots in aligner order.  Without
// `fixed` (the host saw that most candidates take the table path) the followers are marked like table pairs and entered afterwards.
// A thread handles MP_R records (one per 256-record tile of its workgroup) and issues the loads of all of them level by level - codes,
// keys and name hashes; then occupancy bits, Bloom words, the neighbour's key and the scores; then the fragment table's entry; then
// that fragment's key: a wave otherwise waits four dependent round trips per record, one after the other, for the sake of the one
// lane in sixteen whose look-up reaches the table (the one-record-per-thread form ran at the latency of that chain, not at bandwidth).
constexpr int MP_R = 1;
__global__ __launch_bounds__(256) void k_mate_pairs(MdCols m, const uint4 *__restrict__ fkey, uint8_t *__restrict__ code,
                                                    const uint32_t *__restrict__ hash32, const uint32_t *__restrict__ hash_lo, const uint32_t *__restrict__ bloom,
                                                    uint32_t bloom_mask, const uint32_t *__restrict__ coarse /* null: nobody announced a key */,
                                                    uint32_t *table, uint64_t mask, uint32_t *mate, uint32_t *rep_of, uint32_t *err,
                                                    const uint32_t *__restrict__ ftable, const uint32_t *__restrict__ fbits, uint64_t fmask /* 0: no fragments */,
                                                    unsigned long long *fbest, int fixed /* 2: every candidate is matched by the partitioned pass
                                                    (k_mate_bucket) - only the fragment look-ups and the marks are made here */,
                                                    uint64_t *__restrict__ pk, uint32_t *__restrict__ pv,
                                                    uint32_t *__restrict__ tab_list /* null: table inserts are made here */, uint32_t *tab_cnt, uint32_t tab_cap) {
  const uint64_t base = (uint64_t)blockIdx.x * (256 * MP_R) + threadIdx.x;
  uint8_t cd[MP_R];
  uint4 mine[MP_R], prev[MP_R], kc[MP_R];
  uint32_t hi[MP_R], hl[MP_R], bl[MP_R], fw[MP_R], cur[MP_R];
  int32_t sc[MP_R];
  uint64_t fs[MP_R];
  bool hit[MP_R];
  // level 0
#pragma unroll
  for (int r = 0; r < MP_R; r++) {
    const uint64_t i = base + (uint64_t)r * 256;
    const bool in = i < m.n;
    cd[r] = in ? (uint8_t)(code[i] & MC_KIND) : (uint8_t)MC_NONE;
    mine[r] = in ? fkey[i] : make_uint4(0, 0, 0, 0);
    const uint32_t h0 = in ? hash32[i] : 0u, h1 = (in && i > 0) ? hash32[i - 1] : 0u;  // (only a leader's entry is defined; the unused one is dropped)
    hi[r] = cd[r] == MC_LEAD ? h0 : h1;
    hl[r] = (in && cd[r] == MC_TABLE) ? hash_lo[i] : 0u;
  }
  // level 1
#pragma unroll
  for (int r = 0; r < MP_R; r++) {
    const uint64_t i = base + (uint64_t)r * 256;
    const bool cand = cd[r] != MC_NONE;
    fs[r] = frag_hash(mine[r]) & fmask;
    fw[r] = (cand && fmask) ? fbits[fs[r] >> 5] : 0u;
    // the filter's word, behind a coarse level (one bit per 64-byte line of the filter, 8 KB that stay in the L1): in aligner order
    // next to nobody announces a key, and 48 M look-ups of a 4 MB filter in the L2 cost as much as streaming this kernel's columns
    bl[r] = ~0u;
    if (cand && cd[r] != MC_TABLE) {
      const uint32_t wi = (hi[r] >> 5) & bloom_mask, line = wi >> 4;
      bl[r] = (coarse && ((coarse[line >> 5] >> (line & 31u)) & 1u)) ? bloom[wi] : 0u;
    }
    const bool fol = cd[r] == MC_FOLLOW;
    prev[r] = fol ? fkey[i - 1] : make_uint4(0, 0, 0, 0);
    sc[r] = fol ? m.score[i] + m.score[i - 1] : 0;
  }
  // level 2, 3: the first probe of the fragment table
#pragma unroll
  for (int r = 0; r < MP_R; r++) {
    hit[r] = (fw[r] >> (fs[r] & 31)) & 1u;
    cur[r] = hit[r] ? ftable[fs[r]] : 0u;
  }
#pragma unroll
  for (int r = 0; r < MP_R; r++) kc[r] = hit[r] ? fkey[cur[r]] : make_uint4(0, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < MP_R; r++) {
    const uint64_t i = base + (uint64_t)r * 256;
    if (hit[r]) {
      if (key_eq(kc[r], mine[r])) atomicMax(&fbest[cur[r]], 1ull << 63);
      else {
        // an occupied slot of another key: on along the probe sequence (rare: the table is sparse)
        for (uint64_t s = (fs[r] + 1) & fmask;; s = (s + 1) & fmask) {
          if (!((fbits[s >> 5] >> (s & 31)) & 1u)) break;
          const uint32_t c2 = ftable[s];
          if (key_eq(fkey[c2], mine[r])) {
            atomicMax(&fbest[c2], 1ull << 63);
            break;
          }
        }
      }
    }
    bool own = false, want_tab = false;
    uint64_t key = 0;
    if (cd[r] != MC_NONE && fixed == 2) {
      code[i] = (uint8_t)(cd[r] | MC_TABBED);
    } else if (cd[r] != MC_NONE) {  // a candidate that is a true pair
      const bool tab = cd[r] == MC_TABLE || ((bl[r] >> (hi[r] & 31u)) & 1u);
      if (!tab) {  // nobody else announced this key: the two neighbours are the pair
        mate[i] = cd[r] == MC_LEAD ? (uint32_t)i + 1 : (uint32_t)i - 1;
        if (cd[r] == MC_FOLLOW) {
          if (fixed == 1) {
            own = true;
            key = ((uint64_t)(uint32_t)sc[r] << 32) | (uint32_t)pair_hash(pair_key(mine[r], prev[r]));
          } else {
            code[i] = (uint8_t)(cd[r] | MC_TABBED);
          }
        }
      } else {
        code[i] = (uint8_t)(cd[r] | MC_TABBED);
        if (tab_list) want_tab = true;  // (deferred: k_mate_table)
        else mate_table_insert(m, fkey, (uint32_t)i, cd[r] == MC_TABLE ? ((uint64_t)hash32[i] << 32) | hl[r] : qname_hash(m, (uint32_t)i), table, mask, mate, rep_of, err);
      }
    }
    // Round 5: in aligner order the few records that need the table (the sr-tagged copies of an sfm context: 2 % of the records, one or two
    // lanes of EVERY wave) are listed for a dense pass of their own (k_mate_table) instead of walking the table here: a wave waited for
    // its one lane's name hash and compare-and-swap in HBM - 63 lanes idle - and the kernel took 1.7 instead of 0.8 ms.  64 lists, a wave
    // appends to list (workgroup % 64) with one atomic on that list's counter (a cache line of its own).
    if (tab_list) {
      const unsigned long long tm = __ballot(want_tab);
      if (tm) {
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)tm) - 1;
        const uint32_t li = blockIdx.x & 63u;
        uint32_t at = 0;
        if (lane == leader) at = atomicAdd(&tab_cnt[li * 16u], (uint32_t)__popcll(tm));
        at = __shfl(at, leader, 64) + (uint32_t)__popcll(tm & ((1ull << lane) - 1ull));
        if (want_tab) tab_list[(size_t)li * tab_cap + at] = (uint32_t)i;
      }
    }
    if (fixed == 1) {
      const int next_owns = __shfl_down((int)own, 1, 64);
      if (own) {
        pk[i >> 1] = key;
        pv[i >> 1] = (uint32_t)i;
      } else if (!(threadIdx.x & 1u) && !next_owns && i < m.n) {
        pk[i >> 1] = 0xFFFFFFFF00000000ull | (uint32_t)mix64(i);
        pv[i >> 1] = EMPTY;
      }
    }
  }
}

// ---- mates that are NOT neighbours (coordinate-ordered or shuffled input: every candidate would go through the table in HBM, one
// random compare-and-swap each).  The same remedy as for the pairs: the candidates' {32 hash bits of the mate key, record} entries are
// partitioned by hash bits (the sort's scatter passes), one workgroup per bucket matches its entries in an LDS table.  The table pairs
// records up exactly as the global one does (first at the slot = representative, the second claims it, a third marks the group BIG for
// the arrival-order pairing) - DeleteOrStore toggling, :336-340.
constexpr int ML_TILES = 8;
__global__ __launch_bounds__(256) void k_mate_list(uint64_t n, const uint8_t *__restrict__ code, const uint32_t *__restrict__ hash32, uint64_t *__restrict__ mk,
                                                   uint32_t *__restrict__ mv, uint32_t *ne) {
  __shared__ uint64_t lk[ML_TILES * 256];
  __shared__ uint32_t lv[ML_TILES * 256];
  __shared__ uint32_t lcount, gbase;
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
#pragma unroll 2
  for (int tile = 0; tile < ML_TILES; tile++) {
    const uint64_t i = ((uint64_t)blockIdx.x * ML_TILES + (uint64_t)tile) * 256 + threadIdx.x;
    const uint8_t cd = i < n ? (uint8_t)(code[i] & MC_KIND) : (uint8_t)MC_NONE;
    const bool cand = cd != MC_NONE;
    const uint32_t h = cand ? hash32[cd == MC_FOLLOW ? i - 1 : i] : 0u;
    const unsigned long long mask = __ballot(cand);
    if (mask) {
      const int lane = threadIdx.x & 63, leader = __ffsll((long long)mask) - 1;
      uint32_t at = 0;
      if (lane == leader) at = atomicAdd(&lcount, (uint32_t)__popcll(mask));
      at = __shfl(at, leader, 64);
      if (cand) {
        at += (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        lk[at] = (uint64_t)h;
        lv[at] = (uint32_t)i;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) gbase = lcount ? atomicAdd(ne, lcount) : 0u;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < lcount; k += 256) {
    mk[gbase + k] = lk[k];
    mv[gbase + k] = lv[k];
  }
}
constexpr int MB_TARGET = 640, MB_CAP = 2048;
__global__ __launch_bounds__(256) void k_mate_bucket(MdCols m, const uint4 *__restrict__ fkey, const uint64_t *__restrict__ ks, const uint32_t *__restrict__ vs,
                                                     const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ bend, uint32_t *mate, uint32_t *rep_of, uint32_t *err) {
  __shared__ unsigned long long s_key[MB_CAP];  // hash bits << 32 | representative (the first record of its key to arrive at the slot)
  __shared__ uint32_t s_mate[MB_CAP];           // the second one
  const uint32_t start = bstart[blockIdx.x], cnt = bend[blockIdx.x] - start;
  if (cnt == 0) return;
  const uint32_t t = threadIdx.x;
  uint32_t T = 2;
  while (T < 2 * cnt && T < (uint32_t)MB_CAP) T <<= 1;
  const uint32_t tmask = T - 1;
  for (uint32_t k = t; k < T; k += 256) { s_key[k] = PB_EMPTY; s_mate[k] = EMPTY; }
  __syncthreads();
  for (uint32_t e = t; e < cnt; e += 256) {
    const uint32_t h32 = (uint32_t)ks[start + e], i = vs[start + e];
    const unsigned long long mine = ((unsigned long long)h32 << 32) | i;
    uint32_t idx = ((h32 * 0x9E3779B1u) >> 16) & tmask;
    uint32_t rep = EMPTY, slot = 0;
    for (uint32_t probes = 0; probes <= tmask; probes++, idx = (idx + 1) & tmask) {
      const unsigned long long cur = atomicCAS(&s_key[idx], PB_EMPTY, mine);
      if (cur == PB_EMPTY) { rep = i; slot = idx; break; }
      if ((uint32_t)(cur >> 32) == h32 && mate_key_eq(m, fkey, (uint32_t)cur, fkey[i], i)) { rep = (uint32_t)cur; slot = idx; break; }
    }
    if (rep == EMPTY) { atomicOr(&err[1], 4u); continue; }  // the table is full (keys crafted to share hash bits): the host takes the table in HBM
    if (rep == i) continue;  // first of its key at the slot: waits for the second
    rep_of[i] = rep;
    const uint32_t old = atomicCAS(&s_mate[slot], EMPTY, i);
    if (old == EMPTY) { mate[i] = rep; mate[rep] = i; }
    else {
      rep_of[rep] = MATE_BIG;  // more than two records share {split, library, QNAME}
      atomicOr(&err[1], 1u);
    }
  }
}

// members of big groups -> list of (representative << 32 | record); their mate entries are reset
__global__ __launch_bounds__(256) void k_big_collect(uint64_t n, const uint32_t *__restrict__ rep_of, uint32_t *mate, uint64_t *__restrict__ keys,
                                                     uint32_t *__restrict__ vals, uint32_t *count) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t r = rep_of[i];
  if (r == EMPTY) return;
  uint32_t rep;
  if (r == MATE_BIG) rep = (uint32_t)i;
  else if (rep_of[r] == MATE_BIG) rep = r;
  else return;
  const uint32_t at = atomicAdd(count, 1u);
  keys[at] = ((uint64_t)rep << 32) | (uint64_t)i;
  vals[at] = (uint32_t)i;
  mate[i] = EMPTY;
}
// list sorted by (representative, record): the head of every group pairs its members up in arrival order
__global__ __launch_bounds__(256) void k_big_pair(uint32_t cnt, const uint64_t *__restrict__ keys, uint32_t *mate) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= cnt) return;
  const uint32_t rep = (uint32_t)(keys[j] >> 32);
  if (j > 0 && (uint32_t)(keys[j - 1] >> 32) == rep) return;
  for (uint32_t k = j; k + 1 < cnt && (uint32_t)(keys[k + 1] >> 32) == rep; k += 2) {
    const uint32_t a = (uint32_t)keys[k], b = (uint32_t)keys[k + 1];
    mate[a] = b;
    mate[b] = a;
  }
}

// ---------------- pairs
// The pairs are grouped by key WITHOUT a table in HBM (rounds 1-3 ran one compare-and-swap per pair on a 256 MB table: the rate of
// random device-scope atomics, not bandwidth, bounded it).  Round 4:
//   k_pair_list    every pair's owner (its later-arriving record, :336-340) writes ONE 12-byte entry - {score sum | 32 hash bits of the
//                  pair key, owner} - to a compact list (one global atomic per workgroup);
//   radix passes   the list is partitioned by `bbits` hash bits into buckets of <= ~384 pairs (the sort's own scatter kernel; the
//                  list's length stays on the device: over-launched tiles leave at once);
//   k_pair_bounds  first / last entry of every bucket;
//   k_pair_bucket  one workgroup per bucket: find-or-insert into a table in LDS (64-bit compare-and-swap on {hash bits | first
//                  owner}; a hash match is confirmed on the two ends' packed keys, so there are no false merges), best score by LDS
//                  atomic max, the (QNAME asc, later arrival) tournament among the score-tied pairs by LDS compare-and-swap, and the
//                  flags of both reads of every losing pair - insert, tie and flag of :329-396 in one kernel.
// pair_win[owner] = the owner of the winning pair of its key, for the owners of LOSING pairs only (EMPTY everywhere else): all the
// metrics pass needs (metrics.hip).
__device__ __forceinline__ uint64_t pair_entry(const MdCols &m, const uint4 &later, const uint4 &earlier, uint32_t i, uint32_t mt) {
  return ((uint64_t)(uint32_t)(m.score[i] + m.score[mt]) << 32) | (uint32_t)pair_hash(pair_key(later, earlier));
}
// the pairs that formed in the mate table (or in the big groups' arrival-order pairing): their owners carry MC_TABBED in the code
// column (one byte per record to scan; in aligner order next to none of them is marked)
constexpr int PL_TILES = 8;
__global__ __launch_bounds__(256) void k_pair_list_table(MdCols m, const uint4 *__restrict__ fkey, const uint32_t *__restrict__ mate,
                                                         const uint8_t *__restrict__ code, uint64_t *__restrict__ pk, uint32_t *__restrict__ pv,
                                                         uint32_t *np) {
  __shared__ uint64_t lk[PL_TILES * 256];
  __shared__ uint32_t lv[PL_TILES * 256];
  __shared__ uint32_t lcount, gbase;
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
#pragma unroll 2
  for (int tile = 0; tile < PL_TILES; tile++) {
    const uint64_t i64 = ((uint64_t)blockIdx.x * PL_TILES + (uint64_t)tile) * 256 + threadIdx.x;
    const uint32_t i = (uint32_t)i64;
    const uint32_t mt = (i64 < m.n && (code[i64] & MC_TABBED)) ? mate[i] : EMPTY;
    const bool own = mt != EMPTY && mt < i;  // the later-arriving mate owns the pair (:336-340)
    uint64_t key = 0;
    if (own) key = pair_entry(m, fkey[i], fkey[mt], i, mt);
    const unsigned long long mask = __ballot(own);
    if (mask) {
      const int lane = threadIdx.x & 63, leader = __ffsll((long long)mask) - 1;
      uint32_t at = 0;
      if (lane == leader) at = atomicAdd(&lcount, (uint32_t)__popcll(mask));
      at = __shfl(at, leader, 64);
      if (own) {
        at += (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        lk[at] = key;
        lv[at] = i;
      }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) gbase = lcount ? atomicAdd(np, lcount) : 0u;
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < lcount; k += 256) {
    pk[gbase + k] = lk[k];
    pv[gbase + k] = lv[k];
  }
}

// the same over the lists of k_mate_pairs' deferred table inserts (aligner order with a few records on the table path: every owner that
// carries MC_TABBED is in the lists, so the code column of ALL records need not be scanned: 0.31 -> 0.03 ms per 49 M records)
__global__ __launch_bounds__(256) void k_pair_list_lists(MdCols m, const uint4 *__restrict__ fkey, const uint32_t *__restrict__ mate, const uint32_t *__restrict__ tab_list,
                                                         const uint32_t *__restrict__ tab_cnt, uint32_t cap, uint64_t *__restrict__ pk, uint32_t *__restrict__ pv, uint32_t *np) {
  __shared__ uint32_t first[65];
  if (threadIdx.x < 64) {
    uint32_t incl = tab_cnt[threadIdx.x * 16u];
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(incl, d, 64);
      if ((int)threadIdx.x >= d) incl += up;
    }
    first[threadIdx.x + 1] = incl;
    if (threadIdx.x == 0) first[0] = 0;
  }
  __syncthreads();
  __shared__ uint64_t lk[PL_TILES * 256];
  __shared__ uint32_t lv[PL_TILES * 256];
  __shared__ uint32_t lcount, gbase;
  const uint32_t total = first[64], chunk = PL_TILES * 256u;
  // a workgroup collects the entries of PL_TILES * 256 list members in LDS and appends them with ONE global atomic (a returning atomic
  // per wave on the one counter serialises at ~12 ns each: 0.66 ms for 1.3 M members, measured)
  for (uint32_t c0 = blockIdx.x * chunk; c0 < total; c0 += gridDim.x * chunk) {  // (uniform per workgroup)
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
#pragma unroll 2
    for (int tile = 0; tile < PL_TILES; tile++) {
      const uint32_t g = c0 + (uint32_t)tile * 256u + threadIdx.x;
      uint32_t i = 0, mt = EMPTY;
      if (g < total) {
        uint32_t lo = 0, hi = 64;
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (first[mid] <= g) lo = mid; else hi = mid;
        }
        i = tab_list[(size_t)lo * cap + (g - first[lo])];
        mt = mate[i];
      }
      const bool own = mt != EMPTY && mt < i;  // the later-arriving mate owns the pair (:336-340)
      uint64_t key = 0;
      if (own) key = pair_entry(m, fkey[i], fkey[mt], i, mt);
      const unsigned long long mask = __ballot(own);
      if (mask) {
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)mask) - 1;
        uint32_t at = 0;
        if (lane == leader) at = atomicAdd(&lcount, (uint32_t)__popcll(mask));
        at = __shfl(at, leader, 64);
        if (own) {
          at += (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
          lk[at] = key;
          lv[at] = i;
        }
      }
    }
    __syncthreads();
    if (threadIdx.x == 0) gbase = lcount ? atomicAdd(np, lcount) : 0u;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < lcount; k += 256) {
      pk[gbase + k] = lk[k];
      pv[gbase + k] = lv[k];
    }
    __syncthreads();
  }
}

// bucket of an entry: the top `bbits` of the `sbits` low hash bits the list was sorted by
__device__ __forceinline__ uint32_t pair_bucket(uint64_t key, int sbits, int bbits) {
  return bbits ? ((uint32_t)key & ((1u << sbits) - 1u)) >> (sbits - bbits) : 0u;
}
__global__ __launch_bounds__(256) void k_pair_bounds(const uint64_t *__restrict__ ks, const uint32_t *__restrict__ np, int sbits, int bbits,
                                                     uint32_t *__restrict__ bstart, uint32_t *__restrict__ bend) {
  const uint32_t n = *np;
  const uint32_t j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const uint32_t b = pair_bucket(ks[j], sbits, bbits);
  if (j == 0 || pair_bucket(ks[j - 1], sbits, bbits) != b) bstart[b] = j;
  if (j + 1 == n || pair_bucket(ks[j + 1], sbits, bbits) != b) bend[b] = j + 1;
}

__device__ __forceinline__ PairKey pair_key_of(const uint4 *__restrict__ fkey, const uint32_t *__restrict__ mate, uint32_t owner) {
  return pair_key(fkey[owner], fkey[mate[owner]]);
}
// (QNAME asc, later arrival wins) among the best-scoring pairs of a key, on a word in LDS
__device__ __forceinline__ void tournament_lds(const MdCols &m, uint32_t *winner, uint32_t me) {
  uint32_t w = *(volatile uint32_t *)winner;
  for (;;) {
    if (w != EMPTY) {
      const int cq = qname_cmp(m.qname, m.qname_off, me, w);
      if (!(cq < 0 || (cq == 0 && me > w))) return;  // :383-389: on equal QNAME the later completion replaces the holder
    }
    const uint32_t old = atomicCAS(winner, w, me);
    if (old == w) return;
    w = old;
  }
}
__device__ __forceinline__ void pair_lost(const MdCols &m, const uint32_t *__restrict__ mate, uint32_t o, uint32_t w, uint32_t *__restrict__ pair_win,
                                          uint16_t *__restrict__ flag_out) {
  const uint32_t mt = mate[o];
  pair_win[o] = w;
  flag_out[o] = (uint16_t)(flag_out[o] | F_DUPLICATE);
  flag_out[mt] = (uint16_t)(flag_out[mt] | F_DUPLICATE);
}

constexpr int PB_THREADS = 256;
constexpr int PB_TARGET = 384;     // the list is partitioned until a bucket holds at most this many entries on average
constexpr int PB_CAP = 1024;       // table slots in LDS
constexpr int PB_ECAP = 1024;      // entries whose table slot is remembered in LDS between the phases (the others look theirs up again)

// table slot of entry {h32, o}: find-or-insert.  Returns -1 if the table is full.
__device__ __forceinline__ int pb_slot(const MdCols &m, const uint4 *__restrict__ fkey, const uint32_t *__restrict__ mate,
                                       unsigned long long *s_key, uint32_t tmask, uint32_t h32, uint32_t o, bool insert) {
  uint32_t idx = (h32 * 0x9E3779B1u) >> 16 & tmask;
  const unsigned long long mine = ((unsigned long long)h32 << 32) | o;
  for (uint32_t probes = 0; probes <= tmask; probes++, idx = (idx + 1) & tmask) {
    unsigned long long cur = insert ? atomicCAS(&s_key[idx], PB_EMPTY, mine) : s_key[idx];
    if (cur == PB_EMPTY) { if (insert) return (int)idx; else continue; }
    if (cur == mine) return (int)idx;
    if ((uint32_t)(cur >> 32) == h32 && pair_key_eq(pair_key_of(fkey, mate, (uint32_t)cur), pair_key_of(fkey, mate, o))) return (int)idx;
  }
  return -1;
}

__global__ __launch_bounds__(PB_THREADS) void k_pair_bucket(MdCols m, const uint4 *__restrict__ fkey, const uint32_t *__restrict__ mate,
                                                     const uint64_t *__restrict__ ks, uint32_t *__restrict__ vs,
                                                     const uint32_t *__restrict__ bstart, const uint32_t *__restrict__ bend, int cap_slots,
                                                     uint32_t *__restrict__ pair_win, uint16_t *__restrict__ flag_out, int inv_end /* bend holds ~end (RadixBounds) */) {
  __shared__ unsigned long long s_key[PB_CAP];
  __shared__ uint32_t s_best[PB_CAP], s_win[PB_CAP];
  __shared__ uint16_t s_slot[PB_ECAP];
  __shared__ uint32_t s_full, s_min, s_h;
  const uint32_t start = bstart[blockIdx.x], end = inv_end ? ~bend[blockIdx.x] : bend[blockIdx.x];
  if (end <= start) return;
  const uint32_t cnt = end - start;
  const uint32_t t = threadIdx.x;
  if (cnt == 1) return;  // the only pair of its bucket wins
  uint32_t T = 2;
  while (T < 2 * cnt && T < (uint32_t)cap_slots) T <<= 1;
  const uint32_t tmask = T - 1;
  for (uint32_t k = t; k < T; k += PB_THREADS) { s_key[k] = PB_EMPTY; s_best[k] = 0; s_win[k] = EMPTY; }
  if (t == 0) s_full = 0;
  __syncthreads();
  ks += start;
  vs += start;
  // insert + best score (:342-378)
  for (uint32_t e = t; e < cnt; e += PB_THREADS) {
    const uint64_t key = ks[e];
    const uint32_t o = vs[e];
    if (o == EMPTY) continue;  // a hole of the list (k_mate_pairs)
    const int slot = pb_slot(m, fkey, mate, s_key, tmask, (uint32_t)key, o, true);
    if (slot < 0) { s_full = 1; break; }
    atomicMax(&s_best[slot], (uint32_t)(key >> 32));
    if (e < PB_ECAP) s_slot[e] = (uint16_t)slot;
  }
  __syncthreads();
  if (!s_full) {
    // tournament among the pairs with the best score (:379-391)
    for (uint32_t e = t; e < cnt; e += PB_THREADS) {
      const uint64_t key = ks[e];
      const uint32_t o = vs[e];
      if (o == EMPTY) continue;
      const int slot = e < PB_ECAP ? (int)s_slot[e] : pb_slot(m, fkey, mate, s_key, tmask, (uint32_t)key, o, false);
      if ((uint32_t)(key >> 32) == s_best[slot]) tournament_lds(m, &s_win[slot], o);
    }
    __syncthreads();
    for (uint32_t e = t; e < cnt; e += PB_THREADS) {
      const uint32_t o = vs[e];
      if (o == EMPTY) continue;
      const int slot = e < PB_ECAP ? (int)s_slot[e] : pb_slot(m, fkey, mate, s_key, tmask, (uint32_t)ks[e], o, false);
      const uint32_t w = s_win[slot];
      if (w != o) pair_lost(m, mate, o, w, pair_win, flag_out);
    }
    return;
  }
  // More distinct keys than the table holds: not a natural case (the hash spreads the keys evenly over the buckets; only keys crafted
  // to share their hash bits get here).  Exact and slow: the groups are peeled off one by one - the unresolved pair with the smallest
  // owner names the next key, every unresolved pair compares its key with that one; a resolved entry's owner is overwritten with EMPTY.
  for (;;) {
    __syncthreads();
    if (t == 0) { s_min = EMPTY; s_best[0] = 0; s_win[0] = EMPTY; }
    __syncthreads();
    for (uint32_t e = t; e < cnt; e += PB_THREADS) {
      const uint32_t o = vs[e];
      if (o != EMPTY) atomicMin(&s_min, o);
    }
    __syncthreads();
    const uint32_t rep = s_min;
    if (rep == EMPTY) return;
    for (uint32_t e = t; e < cnt; e += PB_THREADS)
      if (vs[e] == rep) s_h = (uint32_t)ks[e];
    __syncthreads();
    const uint32_t rh = s_h;
    const PairKey rk = pair_key_of(fkey, mate, rep);
    for (uint32_t e = t; e < cnt; e += PB_THREADS) {
      const uint32_t o = vs[e];
      if (o != EMPTY && (uint32_t)ks[e] == rh && pair_key_eq(rk, pair_key_of(fkey, mate, o))) atomicMax(&s_best[0], (uint32_t)(ks[e] >> 32));
    }
    __syncthreads();
    for (uint32_t e = t; e < cnt; e += PB_THREADS) {
      const uint32_t o = vs[e];
      if (o != EMPTY && (uint32_t)ks[e] == rh && (uint32_t)(ks[e] >> 32) == s_best[0] && pair_key_eq(rk, pair_key_of(fkey, mate, o)))
        tournament_lds(m, &s_win[0], o);
    }
    __syncthreads();
    for (uint32_t e = t; e < cnt; e += PB_THREADS) {
      const uint32_t o = vs[e];
      if (o != EMPTY && (uint32_t)ks[e] == rh && pair_key_eq(rk, pair_key_of(fkey, mate, o))) {
        if (s_win[0] != o) pair_lost(m, mate, o, s_win[0], pair_win, flag_out);
        vs[e] = EMPTY;
      }
    }
  }
}

// ---------------- the front pass (round 6; VERDICT r5 next #1a / #1b)
// Rounds 2-5 streamed the fixed fields four times before the pair phase: k_adapt_fixed (unclipped positions, sort keys), k_md_keys (packed
// fragment keys, list of true fragments), k_mate_scan (neighbour test on the names) and k_mate_pairs (fragment look-ups, neighbour pairs,
// pair entries) - 180 bytes per read where the data is ~100.  k_md_front makes ONE pass of them:
//   k_frag_list   the true fragments (~0.5 % of paired-end reads) are listed from the FLAG column alone, their keys (unclipped position
//                 from the CIGAR) and group payloads written; the host reads the list's length while the score kernel runs;
//   (k_frag_insert, k_frag_bits: the fragments' table is FINAL before any pair looks its key up)
//   k_md_front    a record per thread (+ three border threads per 256 records, as k_mate_scan): every load of the record is issued up front;
//                 unclipped position, sort key (when the adapt stage has not run yet), packed key, the neighbour test, code / name hash /
//                 Bloom announcement of the records that are not exactly-two-neighbours, the fragment look-up of every true pair, and -
//                 optimistically - the neighbour pairs' mates and entries at their fixed slots, as if nobody announced a key.  In
//                 aligner order nobody does (the count comes back with the call's one read-back) and the mate phase is over; otherwise the
//                 mates are cleared and k_mate_pairs redoes them with the filter, the table and the lists, as before.
// the value the NEXT lane of the wave holds (lane 63: unspecified): one DPP move - a ds_bpermute takes six times the issue time
// (profiles/r3m_isa_rate_probe.txt).  Every lane of the wave must be active.
__device__ __forceinline__ uint32_t next_lane(uint32_t x) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130 /* wave_shl:1 */, 0xf, 0xf, false); }
__device__ __forceinline__ uint64_t next_lane64(uint64_t x) { return (uint64_t)next_lane((uint32_t)x) | ((uint64_t)next_lane((uint32_t)(x >> 32)) << 32); }

struct FrontCols {
  uint64_t n;
  const int32_t *refid, *pos;
  const uint16_t *flag;   // as staged
  const uint8_t *state;   // the has_sr column: 0 live, 1 sr-tagged copy, 2 rejected by the fused predicates (never a candidate: k_md_flag_in)
  const uint16_t *rgid, *rg_lib, *split;
  const uint64_t *cigar_off;
  const uint32_t *cigar;
  const uint64_t *qname_off;
  const uint8_t *qname;
  uint32_t n_ref;
  int pos_bits;
};
// computeUnclippedPosition (:79-110) as k_adapt_fixed has it; o0 .. o3 = the record's first four CIGAR operations (loaded with everything
// else), further ones come from memory
__device__ __forceinline__ int32_t unclipped_pos(uint16_t f, int32_t p, uint64_t c0, uint64_t c1, const uint32_t *__restrict__ cigar, uint32_t o0, uint32_t o1,
                                                 uint32_t o2, uint32_t o3) {
  if ((f & (F_UNMAPPED | F_SECONDARY | F_SUPPLEMENTARY)) != 0) return 0;  // mark-duplicates.go:427,436
  int32_t up = p;
  const uint64_t nop = c1 - c0;
  auto at = [&](uint64_t k) __attribute__((always_inline)) -> uint32_t { return k == 0 ? o0 : (k == 1 ? o1 : (k == 2 ? o2 : (k == 3 ? o3 : cigar[c0 + k]))); };
  if (nop) {
    if (f & F_REVERSED) {  // :90-100
      int32_t clipped = 1;
      up--;
      for (uint64_t k = nop; k-- > 0;) {
        const uint32_t c = at(k), op = c & 0xF;
        const int32_t isclip = (op == OP_S || op == OP_H) ? 1 : 0;
        const int32_t isref = op_consumes_ref(op) ? 1 : 0;
        clipped *= isclip;
        up += (isref | clipped) * (int32_t)(c >> 4);
      }
    } else {  // :101-108
      for (uint64_t k = 0; k < nop; k++) {
        const uint32_t c = at(k), op = c & 0xF;
        if (!(op == OP_S || op == OP_H)) break;
        up -= (int32_t)(c >> 4);
      }
    }
  }
  return up;
}
__device__ __forceinline__ uint16_t flag_in_of(uint16_t f, uint8_t state) { return state == 2 ? (uint16_t)(f | F_SECONDARY) : f; }

// (a thread takes eight consecutive records per round: one 16-byte load of their flags, one 8-byte load of their states.  A few persistent
// workgroups, each appending with ONE global atomic per flush of its LDS list: with a workgroup per 4096 records, 12 K of them queued at
// the one counter for ~12 ns each - 0.15 of the kernel's 0.18 ms)
constexpr int FLI_CAP = 4096, FLI_CHUNK = 2048;
__global__ __launch_bounds__(256) void k_frag_list(FrontCols m, uint4 *__restrict__ fkey, uint32_t *__restrict__ flist, uint32_t *nf,
                                                   unsigned long long *__restrict__ fbest, uint32_t *__restrict__ fwinner) {
  __shared__ uint32_t lq[FLI_CAP];
  __shared__ uint32_t lcount, gbase;
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
  const uint64_t nchunks = (m.n + FLI_CHUNK - 1) / FLI_CHUNK;
  for (uint64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {  // (uniform per workgroup: barriers inside)
    const uint64_t i0 = (ch * 256 + threadIdx.x) * 8;
    uint4 fv = make_uint4(0, 0, 0, 0);
    uint2 sv = make_uint2(0, 0);
    if (i0 < m.n) {  // (a load may run past record n - 1 inside its aligned 16 / 8 bytes; the tests below do not)
      fv = *reinterpret_cast<const uint4 *>(m.flag + i0);
      sv = *reinterpret_cast<const uint2 *>(m.state + i0);
    }
    const uint32_t fw[4] = {fv.x, fv.y, fv.z, fv.w};
    const uint64_t st8 = (uint64_t)sv.x | ((uint64_t)sv.y << 32);
    uint32_t fm = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint16_t f = flag_in_of((uint16_t)(fw[k >> 1] >> (16 * (k & 1))), (uint8_t)(st8 >> (8 * k)));
      if (i0 + (uint64_t)k < m.n && is_candidate(f) && !is_true_pair(f)) fm |= 1u << k;
    }
    if (fm) {
      uint32_t at = atomicAdd(&lcount, (uint32_t)__popc(fm));
      for (; fm; fm &= fm - 1u) lq[at++] = (uint32_t)i0 + (uint32_t)(__ffs((int)fm) - 1);
    }
    __syncthreads();
    const uint32_t cnt = lcount;
    if (cnt + FLI_CHUNK <= (uint32_t)FLI_CAP && ch + gridDim.x < nchunks) continue;  // room for another chunk, and there is one
    // flush: the listed fragments' keys (unclipped position from the CIGAR) and group payloads, all lanes busy
    if (threadIdx.x == 0) gbase = cnt ? atomicAdd(nf, cnt) : 0u;
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < cnt; k += 256) {
      const uint32_t i = lq[k];
      const uint16_t f = m.flag[i], rg = m.rgid[i];  // (a fragment's state is 0 or 1: the flag is the staged one)
      const uint64_t c0 = m.cigar_off[i], c1 = m.cigar_off[i + 1];
      const uint64_t nop = c1 - c0;
      const uint32_t o0 = nop > 0 ? m.cigar[c0] : 0u, o1 = nop > 1 ? m.cigar[c0 + 1] : 0u, o2 = nop > 2 ? m.cigar[c0 + 2] : 0u, o3 = nop > 3 ? m.cigar[c0 + 3] : 0u;
      const uint16_t lib = rg == ELP_NIL16 ? (uint16_t)ELP_NIL16 : m.rg_lib[rg];
      const int32_t up = unclipped_pos(f, m.pos[i], c0, c1, m.cigar, o0, o1, o2, o3);
      fkey[i] = make_uint4((uint32_t)m.refid[i], (uint32_t)up, ((uint32_t)lib << 1) | ((f & F_REVERSED) ? 1u : 0u), (uint32_t)m.split[i]);
      fbest[i] = 0;  // any fragment can become its group's representative
      fwinner[i] = EMPTY;
      flist[gbase + k] = i;
    }
    __syncthreads();
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
  }
}

// Lanes and records: a workgroup of five waves covers MF_RECS = 312 records.  Wave w, lane l holds record base - 2 + 63 w + l: lane 63 holds
// the record lane 0 of the next wave holds as well, so that every neighbour test (lanes 0 .. 62 against the next lane) finds its right-hand
// record one DPP move away, whatever the wave - no border threads, no second fetch of a record's cache lines.  u = 63 w + l in [0, 315)
// numbers the workgroup's records; the tests' results and what a follower needs of its leader (key, score) go through LDS; records
// u = 2 .. 313 are the workgroup's own.
constexpr int MF_THREADS = 320, MF_RECS = 312;
template <bool ADAPT>
__global__ __launch_bounds__(MF_THREADS) void k_md_front(FrontCols m, const int32_t *__restrict__ score, int32_t *__restrict__ upos_out, uint64_t *__restrict__ key_out,
                                                         uint4 *fkey, uint8_t *__restrict__ code, uint32_t *__restrict__ hash32, uint32_t *__restrict__ hash_lo, uint32_t *bloom,
                                                         uint32_t bloom_mask, uint32_t *n_table, const uint32_t *__restrict__ ftable, const uint32_t *__restrict__ fbits,
                                                         uint64_t fmask /* 0: no fragments */, unsigned long long *fbest, uint32_t *__restrict__ mate,
                                                         uint32_t *__restrict__ pair_win, uint64_t *__restrict__ pk, uint32_t *__restrict__ pv, uint32_t *np,
                                                         uint32_t nfixed, int optimistic) {
  const uint64_t base = (uint64_t)blockIdx.x * MF_RECS;
  const uint32_t t = threadIdx.x, lane = t & 63u, u = 63u * (t >> 6) + lane;
  const int64_t a_s = (int64_t)base - 2 + (int64_t)u;
  const bool own_rec = a_s >= 0 && (uint64_t)a_s < m.n;             // the record exists
  const bool valid = a_s >= 0 && (uint64_t)a_s + 1 < m.n;           // ... and so does its right neighbour
  const uint64_t a = own_rec ? (uint64_t)a_s : 0;
  // round 1: the fixed columns and the offsets
  const uint16_t fa_st = m.flag[a], ra = m.rgid[a], sa = m.split[a];
  const uint8_t sta = m.state[a];
  const int32_t r = m.refid[a], p = m.pos[a], sc = score[a];
  const uint64_t oa = m.qname_off[a], oa1 = m.qname_off[a + 1];
  const uint64_t c0 = m.cigar_off[a], c1 = m.cigar_off[a + 1];
  const uint32_t la = (uint32_t)(oa1 - oa);
  const uint64_t nop = c1 - c0;
  // round 2: library id, the first 32 bytes of the name, the first four CIGAR operations
  const uint16_t lia = ra == ELP_NIL16 ? (uint16_t)ELP_NIL16 : m.rg_lib[ra];
  const uint8_t *pa = m.qname + oa;
  const uint64_t a0 = load8(pa), a1 = load8(pa + 8), a2 = load8(pa + 16), a3 = load8(pa + 24);
  const uint32_t o0 = nop > 0 ? m.cigar[c0] : 0u, o1 = nop > 1 ? m.cigar[c0 + 1] : 0u, o2 = nop > 2 ? m.cigar[c0 + 2] : 0u, o3 = nop > 3 ? m.cigar[c0 + 3] : 0u;
  const uint16_t fa = flag_in_of(fa_st, sta);
  const int32_t up = unclipped_pos(fa_st, p, c0, c1, m.cigar, o0, o1, o2, o3);
  const uint4 mine = make_uint4((uint32_t)r, (uint32_t)up, ((uint32_t)lia << 1) | ((fa & F_REVERSED) ? 1u : 0u), (uint32_t)sa);
  // first 32 bytes of the name, zero behind its end (the hash takes them as they are)
  const uint64_t w0 = la > 0 ? low_bytes(a0, la) : 0ull, w1 = la > 8 ? low_bytes(a1, la - 8) : 0ull, w2 = la > 16 ? low_bytes(a2, la - 16) : 0ull,
                 w3 = la > 24 ? low_bytes(a3, la - 24) : 0ull;
  // the right neighbour's side of the test: candidate? library, split, name length, name words (masked like ours: equal lengths are tested first)
  const uint32_t mcand = own_rec && is_mate_candidate(fa) ? 1u : 0u;
  const uint32_t nA = next_lane(mcand | ((uint32_t)lia << 1)), nL = next_lane(la);
  const uint32_t nS = next_lane((uint32_t)sa);
  const uint64_t b0 = next_lane64(w0), b1 = next_lane64(w1), b2 = next_lane64(w2), b3 = next_lane64(w3);
  const uint64_t n_oa = next_lane64(oa);
  bool join = valid && lane < 63u && mcand && (nA & 1u) && (uint16_t)(nA >> 1) == lia && (uint16_t)nS == sa && nL == la;
  if (join) {
    join = ((w0 ^ b0) | (w1 ^ b1) | (w2 ^ b2) | (w3 ^ b3)) == 0;
    const uint8_t *pb = m.qname + n_oa;
    for (uint32_t k = 32; join && k < la; k += 8) join = low_bytes(load8(pa + k), la - k) == low_bytes(load8(pb + k), la - k);
  }
  __shared__ uint8_t s_join[320];
  __shared__ uint4 s_fk[316];
  __shared__ int32_t s_sc[316];
  if (lane < 63u) {
    s_join[u] = join;
    s_fk[u] = mine;
    s_sc[u] = sc;
  }
  __syncthreads();
  const bool mine_rec = own_rec && lane < 63u && u >= 2u && u < 2u + (uint32_t)MF_RECS;  // this thread's record is one of the workgroup's own
  const uint64_t i = a;
  uint8_t cd = MC_NONE;
  if (mine_rec) {
    if (ADAPT) {
      // CoordinateLess primary key (k_adapt_fixed, sort.hip): REFID with unmapped behind the last contig and the records that are not sorted
      // at all behind those, POS, strand
      const uint64_t ru = sta ? (uint64_t)m.n_ref + 1 : (r < 0 ? (uint64_t)m.n_ref : (uint64_t)(uint32_t)r);
      key_out[i] = (ru << (m.pos_bits + 1)) | ((uint64_t)(uint32_t)p << 1) | ((fa_st & F_REVERSED) ? 1ull : 0ull);
      upos_out[i] = up;
    }
    if (!(is_candidate(fa) && !is_true_pair(fa))) fkey[i] = mine;  // (k_frag_list wrote the true fragments' keys: k_frag_insert read them)
    pair_win[i] = EMPTY;
  }
  const uint8_t *j = s_join + u;  // j[0] = joins(i)
  if (mine_rec && mcand) {
    const bool nx = j[0], pv_ = j[-1];
    cd = MC_TABLE;
    if (nx && !pv_ && !j[1]) cd = MC_LEAD;
    else if (pv_ && !nx && !j[-2]) cd = MC_FOLLOW;
    if (cd != MC_FOLLOW) {
      uint64_t h = 0x9e3779b97f4a7c15ull ^ la;  // qname_hash(m, i) from the words at hand
      const uint64_t K = 0xff51afd7ed558ccdull;
      if (la > 0) h = (h ^ w0) * K + (h >> 29);
      if (la > 8) h = (h ^ w1) * K + (h >> 29);
      if (la > 16) h = (h ^ w2) * K + (h >> 29);
      if (la > 24) h = (h ^ w3) * K + (h >> 29);
      for (uint32_t k = 32; k < la; k += 8) h = (h ^ low_bytes(load8(pa + k), la - k)) * K + (h >> 29);
      h = mix64(h ^ ((uint64_t)lia << 48) ^ ((uint64_t)sa << 24));
      const uint32_t hi = (uint32_t)(h >> 32);
      hash32[i] = hi;
      if (cd == MC_TABLE) hash_lo[i] = (uint32_t)h;  // (as k_mate_scan)
      if (cd != MC_LEAD) atomicOr(&bloom[(hi >> 5) & bloom_mask], 1u << (hi & 31u));
    }
  }
  if (mine_rec) code[i] = cd;
  const unsigned long long tb = __ballot(cd == MC_TABLE);
  if (lane == 0 && tb) atomicAdd(&n_table[((blockIdx.x * 5u + (t >> 6)) & 63u) * 16u], (uint32_t)__popcll(tb));  // (64 counters a cache line apart, none in aligner order)
  // every true pair looks its fragment key up (classifyFragment :210-251: a fragment group that holds a read of a true pair loses as a whole)
  if (cd != MC_NONE && fmask) {
    uint64_t s = frag_hash(mine) & fmask;
    for (;; s = (s + 1) & fmask) {
      if (!((fbits[s >> 5] >> (s & 31)) & 1u)) break;
      const uint32_t c2 = ftable[s];
      if (key_eq(fkey[c2], mine)) {
        atomicMax(&fbest[c2], 1ull << 63);
        break;
      }
    }
  }
  // the neighbour pairs, as if nobody announced a key (the caller knows whether anybody did when the count of MC_TABLE records is back)
  if (mine_rec) mate[i] = !optimistic ? EMPTY : (cd == MC_LEAD ? (uint32_t)i + 1 : (cd == MC_FOLLOW ? (uint32_t)i - 1 : EMPTY));
  if (optimistic && mine_rec) {
    if (cd == MC_FOLLOW) {  // the later arrival owns the pair (:336-340): its entry goes to slot i >> 1 (two neighbouring followers cannot both own one)
      pk[i >> 1] = ((uint64_t)(uint32_t)(sc + s_sc[u - 1]) << 32) | (uint32_t)pair_hash(pair_key(mine, s_fk[u - 1]));
      pv[i >> 1] = (uint32_t)i;
    } else if (!(i & 1ull) && !(j[0] && !j[1] && !j[-1])) {  // an even record that owns no pair and whose odd neighbour is no follower either: a hole
      pk[i >> 1] = 0xFFFFFFFF00000000ull | (uint32_t)mix64(i);
      pv[i >> 1] = EMPTY;
    }
    if (i == 0) *np = nfixed;
  }
}

static uint64_t table_size_for(uint64_t n) {
  uint64_t t = 1024;
  while (t < 2 * n + 16) t <<= 1;
  return t;
}

static int markdup_impl(elp_ctx *c) {
  const uint64_t n = c->n;
  ELP_TRY(ensure(c, c->mate, n + 1));
  ELP_TRY(ensure(c, c->pair_win, n + 1));
  if (n == 0) { ELP_TRY(ensure_adapted(c, true)); c->marked = true; return 0; }
  const bool fused = c->tune.md_fused != 1;
  // the front pass also does the adapt stage's fixed-field part when that has not run yet (what a host that marks duplicates first - the
  // reference's order, cmd/filter.go:142-211 - gets)
  const bool fuse_adapt = fused && !c->adapted;
  int pos_bits = 1;
  if (fuse_adapt) ELP_TRY(adapt_begin(c, &pos_bits));
  else ELP_TRY(ensure_adapted(c, false));  // (its quality-error word is read with this call's first read-back, below)
  const unsigned grid = blocks_for(n, 256);
  hipStream_t st = c->stream;
  // flag_in: tournaments see the flags as staged, with the records the fused predicates rejected made non-candidates.  The front pass
  // derives it from the state column as it goes; the separate passes read a patched copy
  uint16_t *flag_in = c->flag.p;
  if (!fused) {
    ELP_TRY(scratch(c, 4, n + 8, &flag_in));
    if (c->n_filtered)
      ELP_LAUNCH(c, "md_flag_in", k_md_flag_in, dim3(grid), dim3(256), 0, n, (const uint16_t *)c->flag.p, (const uint8_t *)c->has_sr.p, flag_in);
    else
      ELP_HIP(c, hipMemcpyAsync(flag_in, c->flag.p, n * sizeof(uint16_t), hipMemcpyDeviceToDevice, st));
  }
  MdCols m{n, c->refid.p, flag_in, c->rgid.p, c->rg_lib.p, c->split.p, c->upos.p, c->score.p, c->qname_off.p, c->qname.p};
  const uint64_t T = table_size_for(n);
  uint32_t *table;
  ELP_TRY(scratch(c, 0, T, &table));
  uint32_t *rep;
  unsigned long long *best;
  ELP_TRY(scratch(c, 2, n + 16, &best));  // (sized for the pair phase's list as well: no reallocation in mid-call)
  uint32_t *winner;
  ELP_TRY(scratch(c, 3, n + 16, &winner));

  uint4 *fkey;
  ELP_TRY(scratch(c, 5, n + 8, &fkey));
  // keys of all records + the list of true fragments (in `rep`'s scratch slot behind the n entries of frep)
  ELP_TRY(ensure(c, c->md_ctr, 16 + 64 * 16));
  uint32_t *nf_dev = c->md_ctr.p;
  ELP_TRY(scratch(c, 1, 2 * n + 16, &rep));
  uint32_t *flist = rep + n + 8;
  ELP_HIP(c, hipMemsetAsync(c->md_ctr.p, 0, (16 + 64 * 16) * sizeof(uint32_t), st));  // every counter of this call in one fill
  uint32_t nf = 0;
  FrontCols fc{n, c->refid.p, c->pos.p, c->flag.p, c->has_sr.p, c->rgid.p, c->rg_lib.p, c->split.p, c->cigar_off.p, c->cigar.p, c->qname_off.p, c->qname.p,
               (uint32_t)c->n_ref, pos_bits};
  if (fused) {
    // the list's length comes back through the mailbox while the stream runs on (the score kernel, when the adapt stage is this call's)
    ELP_TRY(mailbox(c));
    ELP_LAUNCH(c, "md_frag_list", k_frag_list, dim3(std::min<unsigned>(blocks_for(n, FLI_CHUNK), (unsigned)c->n_cu * 8)), dim3(256), 0, fc, fkey, flist, nf_dev, best, winner);
    ELP_HIP(c, hipMemcpyAsync(c->mail, nf_dev, 4, hipMemcpyDeviceToHost, st));
    ELP_HIP(c, hipEventRecord(c->mail_ev, st));
    if (fuse_adapt) ELP_TRY(adapt_scores(c));
  } else {
    ELP_LAUNCH(c, "md_keys", k_md_keys, dim3(blocks_for(n, 256 * MK_TILES)), dim3(256), 0, m, fkey, flist, nf_dev);  // (nf: read with the mate phase's table estimate below)
  }

  // the pair phase's list, partitioned by hash bits: at most n / 2 pairs
  int bbits = 0;
  while (bbits < 24 && ((n / 2 + 1) >> bbits) > (uint64_t)PB_TARGET) bbits++;
  const int ndig = (bbits + 7) / 8, sbits = 8 * ndig;
  const size_t nb = (size_t)1 << bbits;
  uint32_t *np_dev = c->md_ctr.p + 1;

  // ---- mates (+ the pairs' fragment look-ups and the entries of the neighbour pairs)
  uint64_t bw = 1024;  // Bloom filter words: ~2 bits per record, at most 4 MiB (what one XCD's L2 holds)
  while (bw < n / 16 && bw < (1u << 20)) bw <<= 1;
  uint32_t *bloom, *hash32;
  uint8_t *code;
  const uint32_t tab_cap = (uint32_t)((blocks_for(n, 256 * MP_R) + 63) / 64) * 256u * MP_R;  // a list's share of the workgroups x their records
  ELP_TRY(scratch(c, 6, bw + bw / 512 + 16 + n + 16 + (n + 16) / 4 + 64 * (size_t)tab_cap + 16 + n + 16, &bloom));
  uint32_t *coarse = bloom + bw;  // one bit per 16 words of the filter
  hash32 = coarse + bw / 512 + 16;
  code = reinterpret_cast<uint8_t *>(hash32 + n + 8);
  uint32_t *tab_list = hash32 + n + 16 + (n + 16) / 4;  // k_mate_pairs' deferred table inserts (aligner order): 64 lists
  uint32_t *hash_lo = tab_list + 64 * (size_t)tab_cap + 8;  // the low half of the key hashes (hash32: the high half)
  uint32_t *rep_of = c->pair_win.p;  // free until the pair phase fills it
  uint32_t *n_table_dev = c->md_ctr.p + 16;  // 64 counters, 16 words apart
  ELP_HIP(c, hipMemsetAsync(bloom, 0, bw * sizeof(uint32_t), st));
  uint32_t n_tab = 0, n_tab64[64 * 16];
  uint32_t adapt_word[ADAPT_WORDS] = {0, 0, 0, 0, 0, 0};
  uint64_t npmax = 0, nfixed = 0, Tf = 0;
  uint64_t *pk = nullptr;
  uint32_t *pv = nullptr, *ftable = nullptr, *fbits = nullptr;
  unsigned fgrid = 0;
  bool fixed = false, frag_done = false, fast = false;
  if (fused) {
    // the fragments' table first: it is final before any pair looks its key up
    ELP_HIP(c, elp::event_wait(c->mail_ev));
    nf = c->mail[0];
    fgrid = blocks_for(nf, 256);
    Tf = nf ? std::min<uint64_t>(T, table_size_for(4ull * nf)) : 0;  // sparse: most look-ups of the pairs end at an empty slot
    npmax = n + 2;  // (whatever the mates' order turns out to be)
    ELP_TRY(scratch(c, 7, 2 * npmax + (2 * npmax + Tf + Tf / 32 + 64) / 2 + 8, &pk));
    pv = reinterpret_cast<uint32_t *>(pk + 2 * npmax); ftable = pv + 2 * npmax; fbits = ftable + Tf;
    if (nf) {
      ELP_HIP(c, hipMemsetAsync(ftable, 0xFF, Tf * sizeof(uint32_t), st));
      ELP_LAUNCH(c, "md_frag_insert", k_frag_insert, dim3(fgrid), dim3(256), 0, m, (const uint4 *)fkey, (const uint32_t *)flist, nf, ftable, Tf - 1, rep, best);
      ELP_LAUNCH(c, "md_frag_bits", k_frag_bits, dim3(blocks_for(Tf / 32, 256)), dim3(256), 0, (const uint32_t *)ftable, Tf / 32, fbits);
    }
    const int optimistic = c->tune.mate_path == 0;
    const uint32_t nfx = (uint32_t)((n + 1) / 2);
    if (fuse_adapt)
      ELP_LAUNCH(c, "md_front", k_md_front<true>, dim3(blocks_for(n, MF_RECS)), dim3(MF_THREADS), 0, fc, (const int32_t *)c->score.p, c->upos.p, c->key.p, fkey, code, hash32, hash_lo, bloom,
                 (uint32_t)(bw - 1), n_table_dev, (const uint32_t *)ftable, (const uint32_t *)fbits, nf ? Tf - 1 : (uint64_t)0, best, c->mate.p, c->pair_win.p, pk, pv,
                 np_dev, nfx, optimistic);
    else
      ELP_LAUNCH(c, "md_front", k_md_front<false>, dim3(blocks_for(n, MF_RECS)), dim3(MF_THREADS), 0, fc, (const int32_t *)c->score.p, c->upos.p, c->key.p, fkey, code, hash32, hash_lo, bloom,
                 (uint32_t)(bw - 1), n_table_dev, (const uint32_t *)ftable, (const uint32_t *)fbits, nf ? Tf - 1 : (uint64_t)0, best, c->mate.p, c->pair_win.p, pk, pv,
                 np_dev, nfx, optimistic);
    if (fuse_adapt) c->adapted = true;
    // elp_sort_ahead: the keys exist - the coordinate sort's key passes go to the sort lane now and run under the pair phase
    ELP_TRY(sort_presort(c));
    // tournament among the fragments of pair-free groups (the pair bits are complete): queued in front of the read-back
    if (nf) {
      ELP_LAUNCH(c, "md_frag_tie", k_frag_tie, dim3(fgrid), dim3(256), 0, m, (const uint32_t *)flist, nf, (const uint32_t *)rep,
                 (const unsigned long long *)best, winner);
      ELP_LAUNCH(c, "md_frag_flag", k_frag_flag, dim3(fgrid), dim3(256), 0, m, (const uint32_t *)flist, nf, (const uint32_t *)rep,
                 (const unsigned long long *)best, (const uint32_t *)winner, c->flag.p);
    }
    frag_done = true;
  } else {
    ELP_HIP(c, hipMemsetAsync(c->mate.p, 0xFF, n * sizeof(uint32_t), st));
    ELP_HIP(c, hipMemsetAsync(rep_of, 0xFF, n * sizeof(uint32_t), st));
    ELP_LAUNCH(c, "md_mate_scan", k_mate_scan, dim3(grid), dim3(MS_THREADS), 0, m, code, hash32, hash_lo, bloom, (uint32_t)(bw - 1), n_table_dev);
    ELP_HIP(c, hipMemcpyAsync(&nf, nf_dev, 4, hipMemcpyDeviceToHost, st));
  }
  // the table only has to hold the records that are not exactly-two-neighbours (few in aligner order) plus the neighbour pairs a
  // Bloom-filter hit sends there (at most as many again, in practice a fraction): size it by their number, not by n
  ELP_HIP(c, hipMemcpyAsync(n_tab64, n_table_dev, sizeof n_tab64, hipMemcpyDeviceToHost, st));
  const bool adapt_read = c->adapt_pending;
  if (adapt_read) ELP_HIP(c, hipMemcpyAsync(adapt_word, c->adapt_err.p, sizeof adapt_word, hipMemcpyDeviceToHost, st));
  ELP_HIP(c, elp::stream_wait(st));
  if (adapt_read) adapt_note(c, adapt_word);
  if (c->adapt_bad_qual) return adapt_quality_error(c);  // computePhredScore panics on such a record (filters/mark-duplicates.go:64-66)
  for (int k = 0; k < 64; k++) n_tab += n_tab64[k * 16];

  // aligner order (few candidates need a table): the neighbour pairs' entries go to fixed slots, the table's pairs behind them.  Else
  // (coordinate-ordered, shuffled input) every candidate is matched by the partitioned pass (k_mate_list, k_mate_bucket)
  fixed = (uint64_t)n_tab < n / 8 && c->tune.mate_path == 0;
  // 1 neighbours + table in HBM for the rest, 2 partitioned, 0 table in HBM for all.  Measured (16 M reads staged in random order,
  // tools/prof/shuffled_md.py): partitioned 1.59 (buckets) + 0.32 (two scatter passes) + 0.15 (list, bounds) ms, table in HBM 1.84 ms - both
  // are bound by the ~10 random loads of the key comparison every pair needs once (two names, their offsets, read group -> library, split),
  // not by the insert; the table in HBM therefore stays the default for input whose mates are not neighbours
  int mate_mode = fixed ? 1 : (c->tune.mate_path == 1 ? 2 : 0);
  nfixed = fixed ? (n + 1) / 2 : 0;
  if (!fused) {
    npmax = std::max<uint64_t>(nfixed + n / 2 + 1, fixed ? 0 : n + 1);
    // pair list (two buffers each for the radix passes; the partitioned mate pass uses them first) | fragment table and its occupancy bits
    Tf = nf ? std::min<uint64_t>(T, table_size_for(4ull * nf)) : 0;  // sparse: most look-ups of the pairs end at an empty slot
    ELP_TRY(scratch(c, 7, 2 * npmax + (2 * npmax + Tf + Tf / 32 + 64) / 2 + 8, &pk));
    pv = reinterpret_cast<uint32_t *>(pk + 2 * npmax); ftable = pv + 2 * npmax; fbits = ftable + Tf;
    // fragments: group the true fragments (their table is final before the pairs look their keys up)
    fgrid = blocks_for(nf, 256);
    if (nf) {
      ELP_HIP(c, hipMemsetAsync(ftable, 0xFF, Tf * sizeof(uint32_t), st));
      ELP_LAUNCH(c, "md_frag_init", k_frag_init, dim3(fgrid), dim3(256), 0, (const uint32_t *)flist, nf, best, winner);
      ELP_LAUNCH(c, "md_frag_insert", k_frag_insert, dim3(fgrid), dim3(256), 0, m, (const uint4 *)fkey, (const uint32_t *)flist, nf, ftable, Tf - 1, rep, best);
      ELP_LAUNCH(c, "md_frag_bits", k_frag_bits, dim3(blocks_for(Tf / 32, 256)), dim3(256), 0, (const uint32_t *)ftable, Tf / 32, fbits);
    }
  } else {
    // nobody announced a key: the front pass's neighbour pairs stand (mates, entries at their fixed slots, pair_win all EMPTY)
    fast = fixed && n_tab == 0;
    if (!fast) ELP_HIP(c, hipMemsetAsync(c->mate.p, 0xFF, n * sizeof(uint32_t), st));  // (rep_of = pair_win is all EMPTY from the front pass)
  }
  if (n_tab && !fast) ELP_LAUNCH(c, "md_bloom_coarse", k_bloom_coarse, dim3(blocks_for(bw / 16, 256)), dim3(256), 0, (const uint32_t *)bloom, (uint32_t)(bw / 16), coarse);
  // (the front pass made the pairs' fragment look-ups: k_mate_pairs skips them)
  const uint64_t fmask_pairs = (nf && !fused) ? Tf - 1 : (uint64_t)0;

  uint64_t Tm = mate_mode == 0 ? T : std::min<uint64_t>(T, table_size_for(std::min<uint64_t>(n, 4ull * n_tab + 1024)));
  uint32_t e[4] = {0, 0, 0, 0};
  bool listed = false;  // listed: the last pass of k_mate_pairs listed its table inserts (tab_list)
  while (!fast) {
    if (mate_mode != 2) ELP_HIP(c, hipMemsetAsync(table, 0xFF, Tm * sizeof(uint32_t), st));
    ELP_HIP(c, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(np_dev), (int)(uint32_t)nfixed, 1, st));
    // aligner order with a few records on the table path: their inserts are listed and made by a dense pass (the counters of k_mate_scan,
    // read back above, serve as the lists' lengths)
    const bool defer_tab = mate_mode == 1 && n_tab != 0;
    listed = defer_tab;
    if (defer_tab) ELP_HIP(c, hipMemsetAsync(n_table_dev, 0, 64 * 16 * sizeof(uint32_t), st));
    ELP_LAUNCH(c, "md_mate_pairs", k_mate_pairs, dim3(blocks_for(n, 256 * MP_R)), dim3(256), 0, m, (const uint4 *)fkey, code,
               (const uint32_t *)hash32, (const uint32_t *)hash_lo, (const uint32_t *)bloom, (uint32_t)(bw - 1), (const uint32_t *)(n_tab ? coarse : nullptr), table, Tm - 1, c->mate.p,
               rep_of, c->err_flag.p,
               (const uint32_t *)ftable, (const uint32_t *)fbits, fmask_pairs, best, mate_mode == 1 ? 1 : (mate_mode == 2 ? 2 : 0), pk, pv,
               defer_tab ? tab_list : (uint32_t *)nullptr, n_table_dev, tab_cap);
    if (defer_tab)  // (the lists hold the records k_mate_scan counted plus the neighbour pairs a filter hit sent along: in practice a fraction as many again)
      ELP_LAUNCH(c, "md_mate_table", k_mate_table, dim3(blocks_for(std::min<uint64_t>(n, 2ull * n_tab + 4096), 256)), dim3(256), 0, m, (const uint4 *)fkey, (const uint8_t *)code,
                 (const uint32_t *)hash32, (const uint32_t *)hash_lo, (const uint32_t *)tab_list, (const uint32_t *)n_table_dev, tab_cap, table, Tm - 1, c->mate.p, rep_of,
                 c->err_flag.p);
    if (mate_mode == 2) {
      int mbits = 0;
      while (mbits < 24 && ((n + 1) >> mbits) > (uint64_t)MB_TARGET) mbits++;
      const int mdig = (mbits + 7) / 8, msbits = 8 * mdig;
      const size_t mnb = (size_t)1 << mbits;
      uint32_t *ne_dev = c->md_ctr.p + 2, *mbounds;
      ELP_TRY(scratch(c, 0, 2 * mnb + 8, &mbounds));  // (the table in HBM is not used in this mode)
      ELP_HIP(c, hipMemsetAsync(ne_dev, 0, 4, st));
      ELP_HIP(c, hipMemsetAsync(mbounds, 0, 2 * mnb * sizeof(uint32_t), st));
      ELP_LAUNCH(c, "md_mate_list", k_mate_list, dim3(blocks_for(n, 256 * ML_TILES)), dim3(256), 0, n, (const uint8_t *)code, (const uint32_t *)hash32, pk, pv, ne_dev);
      uint64_t *mks = pk;
      uint32_t *mvs = pv;
      if (mdig) {
        ProfScope ps(c, "md_mate_");
        ELP_TRY(radix_sort_pairs_low(c, pk, pv, pk + npmax, pv + npmax, n + 1, mdig, &mks, &mvs, nullptr, false, ne_dev));
      }
      ELP_LAUNCH(c, "md_mate_bounds", k_pair_bounds, dim3(blocks_for(n + 1, 256)), dim3(256), 0, (const uint64_t *)mks, (const uint32_t *)ne_dev, msbits, mbits, mbounds,
                 mbounds + mnb);
      ELP_LAUNCH(c, "md_mate_bucket", k_mate_bucket, dim3((unsigned)mnb), dim3(256), 0, m, (const uint4 *)fkey, (const uint64_t *)mks, (const uint32_t *)mvs, (const uint32_t *)mbounds,
                 (const uint32_t *)(mbounds + mnb), c->mate.p, rep_of, c->err_flag.p);
      // (`table` may have been re-pointed by the scratch call above: take it again for a fall-back pass)
      ELP_TRY(scratch(c, 0, T, &table));
    }
    // tournament among the fragments of pair-free groups (the marks k_mate_pairs left in `best` are complete whatever the error words
    // will say): queued in front of the read-back, so that the device works on it while the host waits
    if (nf && !frag_done) {
      ELP_LAUNCH(c, "md_frag_tie", k_frag_tie, dim3(fgrid), dim3(256), 0, m, (const uint32_t *)flist, nf, (const uint32_t *)rep,
                 (const unsigned long long *)best, winner);
      ELP_LAUNCH(c, "md_frag_flag", k_frag_flag, dim3(fgrid), dim3(256), 0, m, (const uint32_t *)flist, nf, (const uint32_t *)rep,
                 (const unsigned long long *)best, (const uint32_t *)winner, c->flag.p);
      frag_done = true;
    }
    ELP_TRY(fetch_err(c, e));
    if (mate_mode == 2 && (e[1] & 4u)) {
      // a bucket's keys did not fit its LDS table (keys crafted to share hash bits): the table in HBM takes all candidates
      mate_mode = 0;
      Tm = T;
    } else if (!(e[1] & 2u)) {
      break;
    } else {
      if (Tm == T) return set_error(c, ELP_ERR_HIP, "mark duplicates: mate table overflow");
      Tm = T;  // more Bloom-filter hits than estimated: once more with the full-size table (the look-ups and entries are simply made again)
    }
    ELP_HIP(c, hipMemsetAsync(c->err_flag.p + 1, 0, 4, st));
    ELP_HIP(c, hipMemsetAsync(c->mate.p, 0xFF, n * sizeof(uint32_t), st));
    ELP_HIP(c, hipMemsetAsync(rep_of, 0xFF, n * sizeof(uint32_t), st));
  }
  if (e[1]) {
    // keys with more than two records: pair their members up in arrival order
    ELP_HIP(c, hipMemsetAsync(c->err_flag.p + 1, 0, 4, st));
    uint64_t *bk;
    uint32_t *bv, *cnt_dev = c->err_flag.p + 3;  // the scan-total mailbox doubles as the list counter
    ELP_TRY(scratch(c, 2, 2 * n + 8, &bk));  // `best` and `winner` of the fragment phase are free
    ELP_TRY(scratch(c, 3, 2 * n + 8, &bv));
    ELP_HIP(c, hipMemsetAsync(cnt_dev, 0, 4, st));
    ELP_LAUNCH(c, "md_big_collect", k_big_collect, dim3(grid), dim3(256), 0, n, (const uint32_t *)rep_of, c->mate.p, bk, bv, cnt_dev);
    uint32_t cnt = 0;
    ELP_HIP(c, hipMemcpyAsync(&cnt, cnt_dev, 4, hipMemcpyDeviceToHost, st));
    ELP_HIP(c, elp::stream_wait(st));
    ELP_HIP(c, hipMemsetAsync(cnt_dev, 0, 4, st));
    uint64_t *ks;
    uint32_t *vs;
    ProfScope ps(c, "md_big_");
    ELP_TRY(radix_sort_pairs(c, bk, bv, bk + n, bv + n, cnt, &ks, &vs));
    ELP_LAUNCH(c, "md_big_pair", k_big_pair, dim3(blocks_for(cnt, 256)), dim3(256), 0, cnt, (const uint64_t *)ks, c->mate.p);
  }

  // ---- pairs: the entries of the pairs that formed in the table, partition by hash bits, one LDS table per bucket
  {
    uint32_t *bounds;
    ELP_TRY(scratch(c, 1, 2 * nb + 8, &bounds));  // `frep` and the fragment list are free again
    // with radix passes in front, the last of them reports the buckets' bounds (RadixBounds: both arrays start as 0xFF..); a list that is
    // one bucket (tiny inputs) takes its bounds from k_pair_bounds
    const bool folded = ndig > 0;
    ELP_HIP(c, hipMemsetAsync(bounds, folded ? 0xFF : 0, 2 * nb * sizeof(uint32_t), st));
    // aligner order without stragglers: no record announced its key, so nothing went through the mate table - `rep_of` (pair_win's
    // buffer) is still all EMPTY from its fill in front of the scan and no owner carries MC_TABBED: no second fill, no scan of the codes
    const bool no_table = fixed && n_tab == 0 && !e[1];
    if (!no_table) {
      ELP_HIP(c, hipMemsetAsync(c->pair_win.p, 0xFF, n * sizeof(uint32_t), st));
      if (listed)
        ELP_LAUNCH(c, "md_pair_list", k_pair_list_lists, dim3(blocks_for(std::min<uint64_t>(n, 2ull * n_tab + 4096), 256 * PL_TILES)), dim3(256), 0, m, (const uint4 *)fkey,
                   (const uint32_t *)c->mate.p, (const uint32_t *)tab_list, (const uint32_t *)n_table_dev, tab_cap, pk, pv, np_dev);
      else
        ELP_LAUNCH(c, "md_pair_list", k_pair_list_table, dim3(blocks_for(n, 256 * PL_TILES)), dim3(256), 0, m, (const uint4 *)fkey, (const uint32_t *)c->mate.p,
                   (const uint8_t *)code, pk, pv, np_dev);
    }
    uint64_t *ks = pk;
    uint32_t *vs = pv;
    if (ndig) {
      ProfScope ps(c, "md_pair_");
      ELP_TRY(radix_sort_pairs_low(c, pk, pv, pk + npmax, pv + npmax, npmax, ndig, &ks, &vs, nullptr, false, np_dev,
                                   RadixBounds{bounds, bounds + nb, (uint32_t)((1ull << sbits) - 1ull), sbits - bbits}));
    }
    if (!folded)
      ELP_LAUNCH(c, "md_pair_bounds", k_pair_bounds, dim3(blocks_for(npmax, 256)), dim3(256), 0, (const uint64_t *)ks, (const uint32_t *)np_dev, sbits, bbits,
                 bounds, bounds + nb);
    ELP_LAUNCH(c, "md_pair_bucket", k_pair_bucket, dim3((unsigned)nb), dim3(PB_THREADS), 0, m, (const uint4 *)fkey, (const uint32_t *)c->mate.p,
               (const uint64_t *)ks, vs, (const uint32_t *)bounds, (const uint32_t *)(bounds + nb), std::min(c->tune.pair_table_slots, PB_CAP), c->pair_win.p, c->flag.p,
               folded ? 1 : 0);
  }
  c->radix_check_pending = true;
  c->marked = true;
  return 0;
}

}  // namespace elp

extern "C" int elp_mark_duplicates(elp_ctx *c, int also_opticals) {
  (void)also_opticals;  // LIBID is derived from rgid on demand for every read
  if (!c) return ELP_ERR_ARG;
  ELP_HIP(c, hipSetDevice(c->device));
  c->marked = false;
  return elp::markdup_impl(c);
}
