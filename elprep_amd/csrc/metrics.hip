// metrics.hip — DuplicationMetrics counters and optical-duplicate counting.
//
// Reference: filters.MarkOpticalDuplicates (filters/mark-optical-duplicates.go:469-525), markOpticalDuplicatesPair (:182-224),
// countOpticalDuplicates (:275-325), countOpticalDuplicatesFromSlice (:327-368), countOpticalDuplicatesWithGraph (:232-273),
// computeTileInfo (:50-71), isOpticalDuplicateShort (filters/unpedantic.go:32-34), graph.cluster (filters/graph.go:72-85).
//
// For every pair group the "origin" is the best pair; each losing (duplicate) pair contributes its First-flag read.  Origin and
// duplicates are split by the strand of the listed read; the optical count of a list is  n - #connected components  under the
// relation {same RG, same tile != -1, |dx| <= d, |dy| <= d}, which is what both the n <= 3 special cases and the union-find of
// the reference compute.  The reference caps each strand list at 300001 entries and counts 0 optical duplicates for a list longer
// than 300000 (:289-299, :328-330); the same is done here.  Records with the sr tag never reach MarkOpticalDuplicates
// (RemoveOptionalReads, cmd/filter.go:803): they are not counted, and a duplicate pair with a tagged read is never listed.
// Sets of up to OPT_SMALL listed reads are evaluated by one thread each; larger ones cooperatively: their members are sorted by a hash
// of (set, RG, strand, tile), every member compares itself with the members behind it in its bucket (the all-pairs test of
// fillGraphFromAGroup, :232-243, per (RG, tile) bucket) and joins them in a lock-free union-find; optical = members - components.
#include "common.hpp"

namespace elp {

constexpr uint32_t EMPTY = 0xFFFFFFFFu;

struct MxCols {
  uint64_t n;
  const int32_t *refid;
  const uint16_t *flag;   // flags after elp_mark_duplicates
  const uint16_t *rgid;
  const uint16_t *rg_lib;
  const int32_t *upos;
  const uint64_t *qname_off;
  const uint8_t *qname;
  const uint32_t *mate, *pair_win;  // markdup.hip: pair_win[owner of a losing pair] = owner of the winning pair of the key
  int32_t n_lib;
  const uint8_t *has_sr;
  const uint16_t *split;
  int32_t n_split;  // 1 + largest staged split id
};
constexpr uint32_t OPT_SMALL = 32;        // sets up to this size: one thread, all pairs
constexpr uint32_t OPT_LIST_CAP = 300000; // :289-299

__device__ __forceinline__ uint32_t lib_row(const MxCols &m, uint32_t i) {
  const uint16_t rg = m.rgid[i];
  const uint16_t lb = rg == ELP_NIL16 ? (uint16_t)ELP_NIL16 : m.rg_lib[rg];
  return lb == ELP_NIL16 ? (uint32_t)m.n_lib : (uint32_t)lb;  // "Unknown Library" row (:435-447)
}
__device__ __forceinline__ bool true_pair(uint16_t f) { return (f & (F_MULTIPLE | F_NEXT_UNMAPPED)) == F_MULTIPLE; }

// ends ordered as in filters/mark-duplicates.go:347-353 (`second` arrived after `first`)
__device__ __forceinline__ void order_ends(const MxCols &m, uint32_t second, uint32_t first, uint32_t &a1, uint32_t &a2) {
  a1 = second; a2 = first;
  const int32_t r1 = m.refid[a1], r2 = m.refid[a2];
  const int32_t p1 = m.upos[a1], p2 = m.upos[a2];
  const bool v1 = m.flag[a1] & F_REVERSED, v2 = m.flag[a2] & F_REVERSED;
  if (r1 > r2 || (r1 == r2 && (p1 > p2 || (p1 == p2 && v1 && !v2)))) { uint32_t t = a1; a1 = a2; a2 = t; }
}
__device__ __forceinline__ uint32_t listed_read(const MxCols &m, uint32_t owner) {  // :216-221 / :278-283
  uint32_t a1, a2;
  order_ends(m, owner, m.mate[owner], a1, a2);
  return (m.flag[a1] & F_FIRST) ? a1 : a2;
}

// :473-502 — one pass over all records (order does not matter for sums)
// ReadPairsExamined counts reads and is halved at the end of a filter run (:504-506), i.e. once per split file: the reads of
// true pairs are counted per (split, library) in pair_reads[n_split][n_lib + 1] and halved by the host, split by split.
// The same pass lists the LOSING pairs (owner record, pair group) for the optical-duplicate sets: a few per cent of the records,
// so everything behind this kernel works on a compact list instead of record-indexed arrays.  A workgroup owns a contiguous range
// of records, collects its losers in LDS and appends them with one global atomic per ~DC_CAP entries (a global atomic per wave
// on the one list counter would serialise at ~12 ns each).
constexpr int DC_CAP = 2048, DC_STEP = 1024;
// one LDS atomic per distinct counter of the wave (nearly all records of a wave add to the same one or two cells: 64 lanes on one
// LDS address would serialise)
__device__ __forceinline__ void wave_count(unsigned int *lds, int cell) {
  unsigned long long todo = __ballot(cell >= 0);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const int c0 = __builtin_amdgcn_readlane(cell, leader);  // (the leader is wave-uniform: no trip through the LDS crossbar)
    const unsigned long long peers = __ballot(cell == c0);
    if ((int)(threadIdx.x & 63) == leader) atomicAdd(&lds[c0], (unsigned int)__popcll(peers));
    todo &= ~peers;
  }
}
__global__ __launch_bounds__(256) void k_dup_counters(MxCols m, unsigned long long *__restrict__ ctr, unsigned long long *__restrict__ pair_reads,
                                                      uint64_t chunk, uint64_t *__restrict__ lkeys, uint32_t *__restrict__ lvals, uint32_t *list_n) {
  extern __shared__ unsigned int lds_ctr[];  // [(n_lib+1)*7], then [n_split][n_lib+1]
  __shared__ uint2 lq[DC_CAP];               // (owner record, owner of its key's winning pair = the pair group) of the losers collected so far
  __shared__ uint32_t lcount, gbase;
  const int nrow = (m.n_lib + 1) * ELP_NCTR, ncell = nrow + m.n_split * (m.n_lib + 1);
  for (int k = threadIdx.x; k < ncell; k += blockDim.x) lds_ctr[k] = 0;
  if (threadIdx.x == 0) lcount = 0;
  __syncthreads();
  const uint64_t lo = (uint64_t)blockIdx.x * chunk, hi = lo + chunk < m.n ? lo + chunk : m.n;
  for (uint64_t base = lo; base < hi; base += DC_STEP) {
    // three rounds over the step's records, each unrolled: the loads of a round are independent of each other and in flight together
    // (the kernel waits on memory, not on arithmetic)
    constexpr int R = DC_STEP / 256;
    uint8_t sr[R];
    uint16_t fl[R], rg[R], sp[R];
    uint32_t pw[R], mt[R];
#pragma unroll
    for (int t = 0; t < R; t++) {
      const uint64_t i = base + (uint64_t)t * 256 + threadIdx.x;
      const bool in = i < hi;
      sr[t] = in ? m.has_sr[i] : (uint8_t)1;
      fl[t] = in ? m.flag[i] : (uint16_t)0;
      rg[t] = in ? m.rgid[i] : (uint16_t)ELP_NIL16;
      sp[t] = in ? m.split[i] : (uint16_t)0;
      pw[t] = in ? m.pair_win[i] : EMPTY;
      mt[t] = in ? m.mate[i] : EMPTY;
    }
    uint16_t lb[R], fm[R];
    uint8_t srm[R];
#pragma unroll
    for (int t = 0; t < R; t++) {
      lb[t] = rg[t] == ELP_NIL16 ? (uint16_t)ELP_NIL16 : m.rg_lib[rg[t]];
      // the mate's state and flags only matter for a record that is flagged itself (the second of two flagged mates counts the pair;
      // a losing pair's owner is flagged): one record in eleven instead of two random loads for every paired record
      const bool need = mt[t] != EMPTY && ((fl[t] & F_DUPLICATE) || pw[t] != EMPTY);
      srm[t] = need ? m.has_sr[mt[t]] : (uint8_t)0;
      fm[t] = need ? m.flag[mt[t]] : (uint16_t)0;
    }
#pragma unroll
    for (int t = 0; t < R; t++) {
      const uint32_t i = (uint32_t)(base + (uint64_t)t * 256 + threadIdx.x);
      bool loser = false;
      int cell1 = -1, cell2 = -1;  // the (at most two) counters the record adds one to
      if (!sr[t]) {  // sr (and the padding past the range): dropped by RemoveOptionalReads before the metrics pass
        const uint16_t f = fl[t];
        const int lib = lb[t] == ELP_NIL16 ? m.n_lib : (int)lb[t], row = lib * ELP_NCTR;  // "Unknown Library" row (:435-447)
        if (f & F_UNMAPPED) cell1 = row + 3;
        else if (f & (F_SECONDARY | F_SUPPLEMENTARY)) cell1 = row + 2;
        else {
          const bool tp = true_pair(f);
          cell1 = tp ? nrow + (int)sp[t] * (m.n_lib + 1) + lib : row;
          if (f & F_DUPLICATE) {
            if (!tp) cell2 = row + 4;
            // counted once per pair, when the second of two duplicate-flagged mates is met (:186-192)
            else if (mt[t] != EMPTY && mt[t] < i && (fm[t] & F_DUPLICATE) && !srm[t]) cell2 = row + 5;
          }
        }
        // owner of a pair that lost its group; a pair with a tagged read is never completed by the pass over the reads (:186-190)
        loser = pw[t] != EMPTY && !srm[t];
      }
      wave_count(lds_ctr, cell1);
      wave_count(lds_ctr, cell2);
      const unsigned long long mask = __ballot(loser);
      if (mask) {
        const int lane = threadIdx.x & 63, leader = __ffsll((long long)mask) - 1;
        uint32_t at = 0;
        if (lane == leader) at = atomicAdd(&lcount, (uint32_t)__popcll(mask));
        at = __shfl(at, leader, 64);
        if (loser) lq[at + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = make_uint2(i, pw[t]);
      }
    }
    __syncthreads();
    const uint32_t have = lcount;
    if (have > DC_CAP - DC_STEP || base + DC_STEP >= hi) {  // uniform: the next step might not fit, or there is none
      if (threadIdx.x == 0) gbase = have ? atomicAdd(list_n, have) : 0u;
      __syncthreads();
      for (uint32_t k = threadIdx.x; k < have; k += 256) {
        const uint2 e = lq[k];
        lkeys[gbase + k] = (uint64_t)e.y;  // sorted by pair group below
        lvals[gbase + k] = e.x;
      }
      __syncthreads();
      if (threadIdx.x == 0) lcount = 0;
      __syncthreads();
    }
  }
  for (int k = threadIdx.x; k < nrow; k += blockDim.x)
    if (lds_ctr[k]) atomicAdd(&ctr[k], (unsigned long long)lds_ctr[k]);
  for (int k = threadIdx.x + nrow; k < ncell; k += blockDim.x)
    if (lds_ctr[k]) atomicAdd(&pair_reads[k - nrow], (unsigned long long)lds_ctr[k]);
}

// the list sorted by pair group: a group's losers are a run.  Slots of the member table: per group its origin, then its losers.
__global__ __launch_bounds__(256) void k_opt_heads(uint32_t L, const uint64_t *__restrict__ ks, uint32_t *__restrict__ head) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < L) head[j] = (j == 0 || ks[j] != ks[j - 1]) ? 1u : 0u;
}
// gidx = exclusive scan of head: a head's group number is gidx[j], a follower's gidx[j] - 1
__global__ __launch_bounds__(256) void k_opt_starts(uint32_t L, const uint32_t *__restrict__ head, const uint32_t *__restrict__ gidx,
                                                    uint32_t *__restrict__ gstart, uint32_t G) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= L) return;
  if (head[j]) gstart[gidx[j]] = j;
  if (j == L - 1) gstart[G] = L;
}

struct Tile { long long t, x, y; };

// computeTileInfo :50-71; *bad is set where internal.ParseInt would panic.  One pass over the name, eight bytes per load: the
// integer value (and parse status) of fields 2..6 is kept in scalars, the field count decides at the end which three are used
// (7 fields -> 4,5,6; 5 fields -> 2,3,4).
// `word(k)` = bytes 8 k .. 8 k + 7 of the name
template <class W>
__device__ __forceinline__ Tile tile_parse(uint32_t len, bool *bad, W word) {
  long long v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0;
  uint32_t badmask = 0;       // bit k: field k does not parse
  int col = 0;
  long long x = 0;
  uint32_t ndig = 0;          // digits seen in the current field
  bool neg = false, fbad = false, first = true;
  uint64_t w = 0;
  for (uint32_t i = 0;; i++) {
    const bool end = i == len;
    if (!end && (i & 7u) == 0) w = word(i >> 3);
    const uint32_t ch = end ? (uint32_t)':' : (uint32_t)(w >> (8 * (i & 7u))) & 0xFFu;
    if (ch == ':') {
      const bool fb = fbad || ndig == 0 || ndig > 18;
      const long long val = neg ? -x : x;
      if (col == 2) v2 = val; else if (col == 3) v3 = val; else if (col == 4) v4 = val; else if (col == 5) v5 = val; else if (col == 6) v6 = val;
      if (col >= 2 && col <= 6 && fb) badmask |= 1u << col;
      col++;
      x = 0; ndig = 0; neg = false; fbad = false; first = true;
    } else {
      if (first && (ch == '+' || ch == '-')) {
        neg = ch == '-';
      } else {
        const uint32_t d = ch - (uint32_t)'0';
        if (d > 9) fbad = true;
        else if (ndig < 19) x = x * 10 + (long long)d;
        ndig++;
      }
      first = false;
    }
    if (end) break;
  }
  int a;
  if (col == 7) a = 4;
  else if (col == 5) a = 2;
  else return Tile{-1, -1, -1};
  if (badmask & (7u << a)) { *bad = true; return Tile{-1, -1, -1}; }
  return a == 4 ? Tile{v4, v5, v6} : Tile{v2, v3, v4};
}
// Names of up to 48 bytes (all of a sequencer's) are fetched with six loads issued together, in front of the parse: the members of the
// duplicate sets are scattered over the name pool, and a load per eight parsed bytes brought the name's cache line in again and again
// (850 B of HBM traffic per member, measured; a resident wave per name holds more lines than the caches do).
__device__ inline Tile tile_info(const uint8_t *__restrict__ q, uint32_t len, bool *bad) {
  if (len <= 48) {
    const uint64_t w0 = len > 0 ? load8(q) : 0ull, w1 = len > 8 ? load8(q + 8) : 0ull, w2 = len > 16 ? load8(q + 16) : 0ull,
                   w3 = len > 24 ? load8(q + 24) : 0ull, w4 = len > 32 ? load8(q + 32) : 0ull, w5 = len > 40 ? load8(q + 40) : 0ull;
    return tile_parse(len, bad, [&](uint32_t k) { return k == 0 ? w0 : k == 1 ? w1 : k == 2 ? w2 : k == 3 ? w3 : k == 4 ? w4 : w5; });
  }
  return tile_parse(len, bad, [&](uint32_t k) { return load8(q + 8 * k); });
}

struct Member { long long t, x, y; uint32_t rg_rev; };  // rg_rev = rgid << 1 | reversed

__device__ inline Member make_member(const MxCols &m, uint32_t read, uint32_t *err) {
  bool bad = false;
  const uint64_t o = m.qname_off[read];
  Tile tl = tile_info(m.qname + o, (uint32_t)(m.qname_off[read + 1] - o), &bad);
  if (bad) atomicOr(&err[2], 1u);
  return Member{tl.t, tl.x, tl.y, ((uint32_t)m.rgid[read] << 1) | ((m.flag[read] & F_REVERSED) ? 1u : 0u)};
}

// Members of the duplicate sets are laid out group by group; this pass over the sorted list decides the slot of each member and
// notes which read is listed there; the origin's slot (filled by the group's first loser) also gets {set size, group id}.
__global__ __launch_bounds__(256) void k_opt_slots(MxCols m, uint32_t L, const uint64_t *__restrict__ ks, const uint32_t *__restrict__ vs,
                                                   const uint32_t *__restrict__ head, const uint32_t *__restrict__ gidx,
                                                   const uint32_t *__restrict__ gstart, uint32_t *__restrict__ mread, uint2 *__restrict__ ginfo,
                                                   uint32_t *__restrict__ mset) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= L) return;
  const bool is_head = head[j] != 0;
  const uint32_t g = gidx[j] - (is_head ? 0u : 1u);
  const uint32_t j0 = gstart[g], oslot = j0 + g;  // g origins sit in front of the group's own
  const uint32_t slot = j + g + 1;
  mread[slot] = listed_read(m, vs[j]);
  mset[slot] = oslot;  // the set a member belongs to = the slot of its origin
  if (is_head) {
    const uint32_t origin = (uint32_t)ks[j];  // the list is keyed by the owner of the winning pair
    mread[oslot] = listed_read(m, origin);
    mset[oslot] = oslot;
    ginfo[oslot] = make_uint2(gstart[g + 1] - j0 + 1, origin);
  }
}
// dense pass over the member slots: tile / x / y from the QNAME (every lane busy, unlike a pass over all records)
__global__ __launch_bounds__(256) void k_opt_fill(MxCols m, uint32_t total, const uint32_t *__restrict__ mread, Member *__restrict__ members, uint32_t *err) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  members[s] = make_member(m, mread[s], err);
}

__device__ inline uint32_t uf_find(uint32_t *p, uint32_t x) {
  uint32_t r = x;
  while (p[r] != r) r = p[r];
  while (p[x] != r) { uint32_t nx = p[x]; p[x] = r; x = nx; }
  return r;
}
__device__ __forceinline__ bool optical_close(const Member &a, const Member &b, long long dist) {  // isOpticalDuplicate + same RG / strand list / tile
  if (a.t == -1 || b.rg_rev != a.rg_rev || b.t != a.t) return false;
  long long dx = a.x - b.x, dy = a.y - b.y;
  if (dx < 0) dx = -dx;
  if (dy < 0) dy = -dy;
  return dx <= dist && dy <= dist;
}

// one thread per member slot; the origin's slot evaluates its set: optical count = n - #components under the closeness relation
// (sets of two and three members, the bulk, are evaluated in registers; larger ones with a union-find in `parent`)
// hist (may be null): the three set-size histograms per library, [(n_lib + 1)][3][hist_len] (incrementDuplicatesCountsHistograms
// :150-174: sets of n listed reads, of them `optical` optical duplicates - bin n of the first, bin n - optical (if > 0) of the
// second, bin optical + 1 (if optical > 0) of the third; indices beyond the last bin count into the last bin); the first lds_bins
// bins of every histogram are collected in LDS like the optical counts
__global__ __launch_bounds__(128) void k_opt_eval(MxCols m, uint32_t total, const uint2 *__restrict__ ginfo, const Member *__restrict__ members,
                                                  uint32_t *__restrict__ parent, long long dist, unsigned long long *__restrict__ ctr,
                                                  uint32_t *err, unsigned long long *__restrict__ hist, int hist_len, int lds_bins,
                                                  uint32_t *__restrict__ linfo, uint32_t *__restrict__ lcount, uint32_t *n_large) {
  // per-library optical counts are collected in LDS first: the global counters are a handful of addresses, and a global atomic
  // on one address serialises at ~12 ns
  extern __shared__ unsigned int lds_opt[];  // [n_lib + 1], then [(n_lib + 1)][3][lds_bins]
  unsigned int *lds_h = lds_opt + (m.n_lib + 1);
  const int n_h = hist ? (m.n_lib + 1) * 3 * lds_bins : 0;
  for (int k = threadIdx.x; k <= m.n_lib + n_h; k += blockDim.x) lds_opt[k] = 0;
  __syncthreads();
  // (a few thousand workgroups stride over the slots: every workgroup ends with atomics on the same few global counters, and 31 K of
  // them - one per 128 slots - queued up there for 0.37 ms: 12 ns each)
  for (uint32_t b = blockIdx.x * blockDim.x + threadIdx.x; b < total; b += gridDim.x * blockDim.x) {
  const uint2 gi = ginfo[b];
  const uint32_t cnt = gi.x;
  uint32_t optical = 0;
  bool large = false;
  if (cnt > OPT_SMALL) {
    // left to the cooperative kernels: note which strand lists are over the reference's cap (they count 0 and are cut to 300001
    // entries in the set-size histograms) and the capped size of the set
    const Member *g = members + b;
    uint32_t nr = 0;
    for (uint32_t k = 0; k < cnt; k++) nr += g[k].rg_rev & 1u;
    const uint32_t nf = cnt - nr;
    linfo[b] = 1u | (nf > OPT_LIST_CAP ? 2u : 0u) | (nr > OPT_LIST_CAP ? 4u : 0u);
    lcount[b] = (nf > OPT_LIST_CAP ? OPT_LIST_CAP + 1 : nf) + (nr > OPT_LIST_CAP ? OPT_LIST_CAP + 1 : nr);
    atomicAdd(n_large, cnt);
    large = true;
  } else if (cnt >= 2) {
  const Member *g = members + b;
  if (cnt == 2) {
    optical = optical_close(g[0], g[1], dist) ? 1u : 0u;
  } else if (cnt == 3) {
    const Member m0 = g[0], m1 = g[1], m2 = g[2];
    const uint32_t e01 = optical_close(m0, m1, dist), e02 = optical_close(m0, m2, dist), e12 = optical_close(m1, m2, dist);
    const uint32_t edges = e01 + e02 + e12;
    optical = edges >= 2 ? 2u : edges;  // components = 3 - min(edges, 2)
  } else {
    uint32_t *p = parent + b;
    for (uint32_t k = 0; k < cnt; k++) p[k] = k;
    uint32_t comps = cnt;
    for (uint32_t a = 0; a < cnt; a++) {
      const Member ma = g[a];
      if (ma.t == -1) continue;
      for (uint32_t c2 = a + 1; c2 < cnt; c2++) {
        if (optical_close(ma, g[c2], dist)) {
          const uint32_t ra = uf_find(p, a), rb = uf_find(p, c2);
          if (ra != rb) { p[rb] = ra; comps--; }
        }
      }
    }
    optical = cnt - comps;  // sum over both strand lists of (n - components): lists never connect (rg_rev differs)
  }
  }
  if (!large && (optical || (hist && cnt >= 2))) {
    const uint32_t owner = gi.y;
    uint32_t a1, a2;
    order_ends(m, owner, m.mate[owner], a1, a2);
    const uint32_t lib = lib_row(m, a1);  // origin.aln1.LIBID() :381
    if (optical) atomicAdd(&lds_opt[lib], optical);
    if (hist && cnt >= 2) {
      const int idx[3] = {(int)cnt, (int)(cnt - optical), optical ? (int)optical + 1 : 0};  // cnt - optical >= 1: a set has a component
#pragma unroll
      for (int k = 0; k < 3; k++) {
        if (idx[k] <= 0) continue;
        const int bin = idx[k] < hist_len ? idx[k] : hist_len - 1;
        if (bin < lds_bins) atomicAdd(&lds_h[((int)lib * 3 + k) * lds_bins + bin], 1u);
        else atomicAdd(&hist[((size_t)lib * 3 + k) * hist_len + bin], 1ull);
      }
    }
  }
  }
  __syncthreads();
  for (int k = threadIdx.x; k <= m.n_lib; k += blockDim.x)
    if (lds_opt[k]) atomicAdd(&ctr[k * ELP_NCTR + 6], (unsigned long long)lds_opt[k]);
  for (int k = threadIdx.x; k < n_h; k += blockDim.x)
    if (lds_h[k]) atomicAdd(&hist[(size_t)(k / lds_bins) * hist_len + (k % lds_bins)], (unsigned long long)lds_h[k]);
}

// ---- large sets (more than OPT_SMALL listed reads)
// members that can have an optical partner at all (tile parsed, strand list not over the cap) -> (bucket hash, member slot)
__global__ __launch_bounds__(256) void k_large_list(uint32_t total, const uint32_t *__restrict__ mset, const uint32_t *__restrict__ linfo,
                                                    const Member *__restrict__ members, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                    uint32_t *count) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= total) return;
  const uint32_t b = mset[s], li = linfo[b];
  if (!(li & 1u)) return;
  const Member mem = members[s];
  if (mem.t == -1 || (li & ((mem.rg_rev & 1u) ? 4u : 2u))) return;
  const uint32_t at = atomicAdd(count, 1u);
  keys[at] = mix64(mix64(((uint64_t)b << 32) | mem.rg_rev) ^ (uint64_t)mem.t);
  vals[at] = s;
}
__device__ __forceinline__ uint32_t puf_find(uint32_t *parent, uint32_t x) {
  for (;;) {
    const uint32_t p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p == x) return x;
    x = p;
  }
}
// roots only ever get a smaller parent, so there are no cycles and every chain ends
__device__ inline void puf_union(uint32_t *parent, uint32_t a, uint32_t b) {
  for (;;) {
    a = puf_find(parent, a);
    b = puf_find(parent, b);
    if (a == b) return;
    const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
    if (atomicCAS(&parent[hi], hi, lo) == hi) return;
  }
}
// sorted by bucket hash: member j looks at the members behind it with the same hash (same bucket, or a colliding one: the exact
// test is repeated) and joins the ones that are optical duplicates of it
__global__ __launch_bounds__(256) void k_large_union(uint32_t cnt, const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                     const uint32_t *__restrict__ mset, const Member *__restrict__ members, long long dist,
                                                     uint32_t *parent) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cnt) return;
  const uint64_t kj = keys[j];
  const uint32_t sj = vals[j], bj = mset[sj];
  const Member mj = members[sj];
  for (uint32_t k = j + 1; k < cnt && keys[k] == kj; k++) {
    const uint32_t sk = vals[k];
    if (mset[sk] == bj && optical_close(mj, members[sk], dist)) puf_union(parent, j, k);
  }
}
// per set: members listed (low word) and roots among them (high word)
__global__ __launch_bounds__(256) void k_large_roots(uint32_t cnt, const uint32_t *__restrict__ vals, const uint32_t *__restrict__ mset,
                                                     const uint32_t *__restrict__ parent, unsigned long long *setcnt) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cnt) return;
  atomicAdd(&setcnt[mset[vals[j]]], 1ull | (parent[j] == j ? (1ull << 32) : 0ull));
}
// one thread per member slot; origins of large sets account for their set
__global__ __launch_bounds__(256) void k_large_final(MxCols m, uint32_t total, const uint2 *__restrict__ ginfo, const uint32_t *__restrict__ linfo,
                                                     const uint32_t *__restrict__ lcount, const unsigned long long *__restrict__ setcnt,
                                                     unsigned long long *__restrict__ ctr, unsigned long long *__restrict__ hist, int hist_len) {
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= total || !(linfo[b] & 1u) || ginfo[b].x == 0) return;
  const unsigned long long sc = setcnt[b];
  const uint32_t optical = (uint32_t)sc - (uint32_t)(sc >> 32);
  const uint32_t owner = ginfo[b].y;
  uint32_t a1, a2;
  order_ends(m, owner, m.mate[owner], a1, a2);
  const uint32_t lib = lib_row(m, a1);
  if (optical) atomicAdd(&ctr[lib * ELP_NCTR + 6], (unsigned long long)optical);
  if (hist) {
    const uint32_t cnt = lcount[b];
    const int idx[3] = {(int)cnt, (int)(cnt - optical), optical ? (int)optical + 1 : 0};
    for (int k = 0; k < 3; k++) {
      if (idx[k] <= 0) continue;
      const int bin = idx[k] < hist_len ? idx[k] : hist_len - 1;
      atomicAdd(&hist[((size_t)lib * 3 + k) * hist_len + bin], 1ull);
    }
  }
}
__global__ __launch_bounds__(256) void k_iota32(uint32_t *v, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}

// number of pair groups (= origins, the pairs that stayed best of their group) per library
__global__ __launch_bounds__(256) void k_origin_count(MxCols m, unsigned long long *__restrict__ origins) {
  extern __shared__ unsigned int lds_org[];  // [n_lib + 1]
  for (int k = threadIdx.x; k <= m.n_lib; k += blockDim.x) lds_org[k] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m.n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t mt = m.mate[i];
    if (mt == EMPTY || mt > (uint32_t)i || m.pair_win[i] != EMPTY) continue;  // not the owner of a pair, or of a pair that lost
    uint32_t a1, a2;
    order_ends(m, (uint32_t)i, mt, a1, a2);
    atomicAdd(&lds_org[lib_row(m, a1)], 1u);
  }
  __syncthreads();
  for (int k = threadIdx.x; k <= m.n_lib; k += blockDim.x)
    if (lds_org[k]) atomicAdd(&origins[k], (unsigned long long)lds_org[k]);
}

static int metrics_impl(elp_ctx *c, int dist, int64_t *counters_host, int64_t *hist_host, int hist_len) {
  const uint64_t n = c->n;
  const int ncell = (c->n_lib + 1) * ELP_NCTR;
  if (!c->marked) return set_error(c, ELP_ERR_ARG, "elp_dup_metrics: call elp_mark_duplicates first");
  // device block: the counters, then (if asked for) the origins per library and the three histograms per library
  const size_t nhist = hist_host ? (size_t)(c->n_lib + 1) * 3 * (size_t)hist_len : 0, norg = hist_host ? (size_t)(c->n_lib + 1) : 0;
  const int n_split = (int)c->max_split + 1;
  const size_t npr = (size_t)n_split * (size_t)(c->n_lib + 1);
  if (((size_t)ncell + npr) * sizeof(unsigned int) > 46000)  // + 16 KB of static LDS in k_dup_counters
    return set_error(c, ELP_ERR_UNSUPPORTED, "elp_dup_metrics: %d split ids x %d libraries in one context", n_split, c->n_lib + 1);
  unsigned long long *ctr;
  ELP_TRY(scratch(c, 0, (size_t)ncell + norg + nhist + npr + 8, &ctr));
  unsigned long long *origins = ctr + ncell, *hist = hist_host ? origins + norg : nullptr, *pair_reads = ctr + ncell + norg + nhist;
  hipStream_t st = c->stream;
  ELP_HIP(c, hipMemsetAsync(ctr, 0, ((size_t)ncell + norg + nhist + npr) * sizeof(unsigned long long), st));
  int lds_bins = hist_host ? std::min(hist_len, 32) : 0;
  if ((size_t)(c->n_lib + 1) * (1 + 3 * (size_t)lds_bins) * sizeof(unsigned int) > 32768) lds_bins = 0;
  if (n) {
    MxCols m{n, c->refid.p, c->flag.p, c->rgid.p, c->rg_lib.p, c->upos.p, c->qname_off.p, c->qname.p, c->mate.p, c->pair_win.p, c->n_lib,
             c->has_sr.p, c->split.p, n_split};
    const unsigned grid = blocks_for(n, 256);
    // losing pairs: at most one per two records.  Slot 1: keys (u64) x 2 and values (u32) x 2 of the list and its sort buffers
    const size_t lcap = n / 2 + 8;
    uint32_t *gs;
    ELP_TRY(scratch(c, 1, 6 * lcap + 16, &gs));
    uint64_t *pk = reinterpret_cast<uint64_t *>(gs);
    uint32_t *pv = gs + 4 * lcap;
    uint32_t *mailbox = c->err_flag.p + 3;  // the scan-total word: length of the list, then members of large sets, then their candidates
    ELP_HIP(c, hipMemsetAsync(mailbox, 0, 4, st));
    const unsigned cgrid = (unsigned)std::min<uint64_t>(2048, (n + DC_STEP - 1) / DC_STEP);
    const uint64_t chunk = (((n + cgrid - 1) / cgrid) + DC_STEP - 1) / DC_STEP * DC_STEP;
    ELP_LAUNCH(c, "mx_counters", k_dup_counters, dim3(cgrid), dim3(256), ((size_t)ncell + npr) * sizeof(unsigned int), m, ctr, pair_reads, chunk, pk, pv,
               mailbox);
    uint32_t L = 0;
    ELP_HIP(c, hipMemcpyAsync(&L, mailbox, 4, hipMemcpyDeviceToHost, st));
    ELP_HIP(c, elp::stream_wait(st));
    ELP_HIP(c, hipMemsetAsync(mailbox, 0, 4, st));
    if ((size_t)L > lcap) return set_error(c, ELP_ERR_HIP, "elp_dup_metrics: more losing pairs than pairs");
    uint32_t total = 0, G = 0;
    uint64_t *ks = pk;
    uint32_t *vs = pv, *head = nullptr, *gidx = nullptr, *gstart = nullptr;
    if (L) {
      int ndig = 1;
      while (ndig < 4 && (n >> (8 * ndig)) != 0) ndig++;  // pair groups are record indices < n < 2^32
      {
        ProfScope ps(c, "mx_");
        ELP_TRY(radix_sort_pairs_low(c, pk, pv, pk + lcap, pv + lcap, L, ndig, &ks, &vs));
      }
      // the halves the sort left free hold the head flags, the group numbers and the group starts
      head = reinterpret_cast<uint32_t *>(ks == pk ? pk + lcap : pk);
      gidx = head + lcap;
      gstart = vs == pv ? pv + lcap : pv;
      ELP_LAUNCH(c, "mx_opt_heads", k_opt_heads, dim3(blocks_for(L, 256)), dim3(256), 0, L, (const uint64_t *)ks, head);
      {
        ProfScope ps(c, "mx_");
        ELP_TRY(exclusive_scan_u32(c, head, gidx, L, &G));
      }
      ELP_LAUNCH(c, "mx_opt_starts", k_opt_starts, dim3(blocks_for(L, 256)), dim3(256), 0, L, (const uint32_t *)head, (const uint32_t *)gidx, gstart, G);
      total = L + G;
    }
    if (total) {
      Member *members;
      uint32_t *wk, *mread;
      uint2 *ginfo;
      const size_t tp = ((size_t)total + 7) & ~(size_t)7;
      ELP_TRY(scratch(c, 2, (size_t)total + 4, &members));
      ELP_TRY(scratch(c, 3, 12 * tp + 16, &wk));  // parent | mset | linfo | lcount | setcnt (u64) | vals x 2 | keys (u64) x 2
      ELP_TRY(scratch(c, 4, (size_t)total + 4, &mread));
      ELP_TRY(scratch(c, 5, (size_t)total + 4, &ginfo));
      uint32_t *parent = wk, *mset = wk + tp, *linfo = wk + 2 * tp, *lcount = wk + 3 * tp, *lvals = wk + 6 * tp;
      unsigned long long *setcnt = reinterpret_cast<unsigned long long *>(wk + 4 * tp);
      uint64_t *lkeys = reinterpret_cast<uint64_t *>(wk + 8 * tp);
      ELP_HIP(c, hipMemsetAsync(ginfo, 0, (size_t)total * sizeof(uint2), st));
      ELP_HIP(c, hipMemsetAsync(linfo, 0, 4 * tp * sizeof(uint32_t), st));  // linfo, lcount, setcnt
      ELP_LAUNCH(c, "mx_opt_slots", k_opt_slots, dim3(blocks_for(L, 256)), dim3(256), 0, m, L, (const uint64_t *)ks, (const uint32_t *)vs,
                 (const uint32_t *)head, (const uint32_t *)gidx, (const uint32_t *)gstart, mread, ginfo, mset);
      ELP_LAUNCH(c, "mx_opt_fill", k_opt_fill, dim3(blocks_for(total, 256)), dim3(256), 0, m, total, (const uint32_t *)mread, members, c->err_flag.p);
      ELP_LAUNCH(c, "mx_opt_eval", k_opt_eval, dim3(std::min(blocks_for(total, 128), 2048u)), dim3(128),
                 (size_t)(c->n_lib + 1) * (1 + (hist ? 3 * (size_t)lds_bins : 0)) * sizeof(unsigned int), m, total, (const uint2 *)ginfo,
                 (const Member *)members, parent, (long long)dist, ctr, c->err_flag.p, hist, hist_len, lds_bins, linfo, lcount, mailbox);
      uint32_t n_large = 0;
      ELP_HIP(c, hipMemcpyAsync(&n_large, mailbox, 4, hipMemcpyDeviceToHost, st));
      ELP_HIP(c, elp::stream_wait(st));
      ELP_HIP(c, hipMemsetAsync(mailbox, 0, 4, st));
      if (n_large) {
        ELP_LAUNCH(c, "mx_large_list", k_large_list, dim3(blocks_for(total, 256)), dim3(256), 0, total, (const uint32_t *)mset, (const uint32_t *)linfo,
                   (const Member *)members, lkeys, lvals, mailbox);
        uint32_t cl = 0;
        ELP_HIP(c, hipMemcpyAsync(&cl, mailbox, 4, hipMemcpyDeviceToHost, st));
        ELP_HIP(c, elp::stream_wait(st));
        ELP_HIP(c, hipMemsetAsync(mailbox, 0, 4, st));
        if (cl) {
          uint64_t *ks;
          uint32_t *vs;
          ProfScope ps(c, "mx_large_");
          ELP_TRY(radix_sort_pairs(c, lkeys, lvals, lkeys + tp, lvals + tp, cl, &ks, &vs));
          ELP_LAUNCH(c, "mx_large_iota", k_iota32, dim3(blocks_for(cl, 256)), dim3(256), 0, parent, cl);
          ELP_LAUNCH(c, "mx_large_union", k_large_union, dim3(blocks_for(cl, 256)), dim3(256), 0, cl, (const uint64_t *)ks, (const uint32_t *)vs,
                     (const uint32_t *)mset, (const Member *)members, (long long)dist, parent);
          ELP_LAUNCH(c, "mx_large_roots", k_large_roots, dim3(blocks_for(cl, 256)), dim3(256), 0, cl, (const uint32_t *)vs, (const uint32_t *)mset,
                     (const uint32_t *)parent, setcnt);
        }
        ELP_LAUNCH(c, "mx_large_final", k_large_final, dim3(blocks_for(total, 256)), dim3(256), 0, m, total, (const uint2 *)ginfo, (const uint32_t *)linfo,
                   (const uint32_t *)lcount, (const unsigned long long *)setcnt, ctr, hist, hist_len);
      }
    }
    if (hist)
      ELP_LAUNCH(c, "mx_origin_count", k_origin_count, dim3(std::min(grid, 2048u)), dim3(256), (c->n_lib + 1) * sizeof(unsigned int), m, origins);
  }
  std::vector<unsigned long long> h((size_t)ncell + norg + nhist + npr);
  ELP_HIP(c, hipMemcpyAsync(h.data(), ctr, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  uint32_t e[4];
  ELP_TRY(fetch_err(c, e));
  if (e[2]) {
    ELP_HIP(c, hipMemsetAsync(c->err_flag.p + 2, 0, 4, st));
    return set_error(c, ELP_ERR_DATA, "QNAME tile/x/y field is not an integer (reference: internal.ParseInt panics, filters/mark-optical-duplicates.go:55-61)");
  }
  for (int l = 0; l <= c->n_lib; l++)
    for (int k = 0; k < ELP_NCTR; k++) counters_host[l * ELP_NCTR + k] = (int64_t)h[l * ELP_NCTR + k];
  // ReadPairsExamined counts reads, then halves (:504-506) - per filter run, i.e. per split file
  for (int l = 0; l <= c->n_lib; l++) {
    int64_t pairs = 0;
    for (int sp = 0; sp < n_split; sp++) pairs += (int64_t)(h[(size_t)ncell + norg + nhist + (size_t)sp * (c->n_lib + 1) + l] / 2);
    counters_host[l * ELP_NCTR + 1] = pairs;
  }
  if (hist_host) {
    // the device counted the sets that have duplicates; an origin without duplicates is a set of one read, none of them optical
    const unsigned long long *ho = h.data() + ncell, *hh = ho + norg;
    for (int l = 0; l <= c->n_lib; l++) {
      int64_t *out = hist_host + (size_t)l * 3 * hist_len;
      unsigned long long with_dups = 0;
      for (size_t k = 0; k < (size_t)3 * hist_len; k++) out[k] = (int64_t)hh[(size_t)l * 3 * hist_len + k];
      for (int b = 0; b < hist_len; b++) with_dups += hh[(size_t)l * 3 * hist_len + b];
      const int64_t alone = (int64_t)(ho[l] - with_dups);
      const int one = 1 < hist_len ? 1 : hist_len - 1;
      out[one] += alone;             // duplicatesCountHistogram[1]
      out[hist_len + one] += alone;  // nonOpticalDuplicatesCountHistogram[1]
    }
  }
  return 0;
}

}  // namespace elp

namespace elp {
// The pass on the context's side lane (common.hpp): the shadow context sees the columns the pass reads as VIEWS for the duration of the
// call (they are taken back whatever happens: the shadow owns nothing but its scratch), runs on its own stream behind what the context's
// stream holds now, and returns when its own stream is done - the context's stream never waits for it.
static int metrics_on_side(elp_ctx *c, int dist, int64_t *counters_host, int64_t *hist_host, int hist_len) {
  if (!c->marked) return set_error(c, ELP_ERR_ARG, "elp_dup_metrics: call elp_mark_duplicates first");
  // (the time-out word of mark duplicates' radix passes is the context's: whoever reads its error words next - the sort's last read-back,
  // elp_sync, elp_get_flags - reports it; reading it here would wait for the context's whole stream)
  elp_ctx *s = nullptr;
  ELP_TRY(side_lane(c, 0, &s));
  s->n = c->n; s->n_lib = c->n_lib; s->n_rg = c->n_rg; s->n_ref = c->n_ref; s->max_split = c->max_split; s->marked = c->marked;
  s->n_sr = c->n_sr; s->n_filtered = c->n_filtered;
  s->flag.p = c->flag.p; s->rgid.p = c->rgid.p; s->rg_lib.p = c->rg_lib.p; s->refid.p = c->refid.p; s->qname_off.p = c->qname_off.p; s->qname.p = c->qname.p;
  s->pair_win.p = c->pair_win.p; s->mate.p = c->mate.p; s->has_sr.p = c->has_sr.p; s->upos.p = c->upos.p; s->split.p = c->split.p;
  const int rc = metrics_impl(s, dist, counters_host, hist_host, hist_len);
  s->flag.p = nullptr; s->rgid.p = nullptr; s->rg_lib.p = nullptr; s->refid.p = nullptr; s->qname_off.p = nullptr; s->qname.p = nullptr;
  s->pair_win.p = nullptr; s->mate.p = nullptr; s->has_sr.p = nullptr; s->upos.p = nullptr; s->split.p = nullptr;
  if (rc != 0) {
    (void)elp::stream_wait(s->stream);
    c->err = s->err;
  }
  return rc;
}
}  // namespace elp

extern "C" int elp_dup_metrics(elp_ctx *c, int optical_pixel_distance, int64_t *counters) {
  if (!c || !counters) return ELP_ERR_ARG;
  ELP_HIP(c, hipSetDevice(c->device));
  return elp::metrics_on_side(c, optical_pixel_distance, counters, nullptr, 0);
}

extern "C" int elp_dup_metrics_hist(elp_ctx *c, int optical_pixel_distance, int64_t *counters, int64_t *hist, int hist_len) {
  if (!c || !counters || !hist || hist_len < 2) return elp::set_error(c, ELP_ERR_ARG, "elp_dup_metrics_hist: bad arguments (hist_len >= 2)");
  ELP_HIP(c, hipSetDevice(c->device));
  return elp::metrics_on_side(c, optical_pixel_distance, counters, hist, hist_len);
}
