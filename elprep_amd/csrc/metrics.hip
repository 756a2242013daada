// metrics.hip — DuplicationMetrics counters and optical-duplicate counting.
//
// Reference: filters.MarkOpticalDuplicates (filters/mark-optical-duplicates.go:469-525), markOpticalDuplicatesPair (:182-224),
// countOpticalDuplicates (:275-325), countOpticalDuplicatesFromSlice (:327-368), countOpticalDuplicatesWithGraph (:232-273),
// computeTileInfo (:50-71), isOpticalDuplicateShort (filters/unpedantic.go:32-34), graph.cluster (filters/graph.go:72-85).
//
// For every pair group the "origin" is the best pair; each losing (duplicate) pair contributes its First-flag read.  Origin and
// duplicates are split by the strand of the listed read; the optical count of a list is  n - #connected components  under the
// relation {same RG, same tile != -1, |dx| <= d, |dy| <= d}, which is what both the n <= 3 special cases and the union-find of
// the reference compute.  Lists longer than 300000 (the reference's cap, :289-299) are rejected as unsupported.
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
  const uint32_t *mate, *prep, *pwinner;
  int32_t n_lib;
};

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
__global__ __launch_bounds__(256) void k_dup_counters(MxCols m, unsigned long long *__restrict__ ctr) {
  extern __shared__ unsigned int lds_ctr[];  // [(n_lib+1)*7]
  const int ncell = (m.n_lib + 1) * ELP_NCTR;
  for (int k = threadIdx.x; k < ncell; k += blockDim.x) lds_ctr[k] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m.n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint16_t f = m.flag[i];
    unsigned int *row = lds_ctr + lib_row(m, (uint32_t)i) * ELP_NCTR;
    if (f & F_UNMAPPED) { atomicAdd(&row[3], 1u); continue; }
    if (f & (F_SECONDARY | F_SUPPLEMENTARY)) { atomicAdd(&row[2], 1u); continue; }
    const bool tp = true_pair(f);
    atomicAdd(&row[tp ? 1 : 0], 1u);
    if (f & F_DUPLICATE) {
      if (!tp) atomicAdd(&row[4], 1u);
      else {
        const uint32_t mt = m.mate[i];
        // counted once per pair, when the second of two duplicate-flagged mates is met (:186-192)
        if (mt != EMPTY && mt < (uint32_t)i && (m.flag[mt] & F_DUPLICATE)) atomicAdd(&row[5], 1u);
      }
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < ncell; k += blockDim.x)
    if (lds_ctr[k]) atomicAdd(&ctr[k], (unsigned long long)lds_ctr[k]);
}

// losing pairs per group
__global__ __launch_bounds__(256) void k_opt_count(MxCols m, uint32_t *gsize) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.n) return;
  const uint32_t rep = m.prep[i];
  if (rep == EMPTY || m.pwinner[rep] == (uint32_t)i) return;
  atomicAdd(&gsize[rep], 1u);
}
__global__ __launch_bounds__(256) void k_opt_plus_origin(uint64_t n, uint32_t *gsize) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && gsize[i] > 0) gsize[i] += 1;
}

struct Tile { long long t, x, y; };

// computeTileInfo :50-71; *bad is set where internal.ParseInt would panic.  One pass over the name, eight bytes per load: the
// integer value (and parse status) of fields 2..6 is kept in scalars, the field count decides at the end which three are used
// (7 fields -> 4,5,6; 5 fields -> 2,3,4).
__device__ inline Tile tile_info(const uint8_t *__restrict__ q, uint32_t len, bool *bad) {
  long long v2 = 0, v3 = 0, v4 = 0, v5 = 0, v6 = 0;
  uint32_t badmask = 0;       // bit k: field k does not parse
  int col = 0;
  long long x = 0;
  uint32_t ndig = 0;          // digits seen in the current field
  bool neg = false, fbad = false, first = true;
  uint64_t w = 0;
  for (uint32_t i = 0;; i++) {
    const bool end = i == len;
    if (!end && (i & 7u) == 0) w = load8(q + i);
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

struct Member { long long t, x, y; uint32_t rg_rev; };  // rg_rev = rgid << 1 | reversed

__device__ inline Member make_member(const MxCols &m, uint32_t read, uint32_t *err) {
  bool bad = false;
  const uint64_t o = m.qname_off[read];
  Tile tl = tile_info(m.qname + o, (uint32_t)(m.qname_off[read + 1] - o), &bad);
  if (bad) atomicOr(&err[2], 1u);
  return Member{tl.t, tl.x, tl.y, ((uint32_t)m.rgid[read] << 1) | ((m.flag[read] & F_REVERSED) ? 1u : 0u)};
}

// Members of the duplicate sets are laid out group by group (goff); this pass (one thread per record, most exit at once) only
// decides the slot of each member and notes which read is listed there; the origin's slot also gets {set size, group id}.
__global__ __launch_bounds__(256) void k_opt_slots(MxCols m, const uint32_t *__restrict__ goff, uint32_t *gfill, uint32_t *__restrict__ mread,
                                                   uint2 *__restrict__ ginfo) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m.n) return;
  const uint32_t rep = m.prep[i];
  if (rep == EMPTY) return;
  const bool is_origin = m.pwinner[rep] == (uint32_t)i;
  const uint32_t g0 = goff[rep], g1 = goff[rep + 1];
  if (is_origin && g1 == g0) return;  // group without duplicates: count is 0
  const uint32_t slot = g0 + (is_origin ? 0u : 1u + atomicAdd(&gfill[rep], 1u));
  mread[slot] = listed_read(m, (uint32_t)i);
  if (is_origin) ginfo[slot] = make_uint2(g1 - g0, rep);
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
                                                  uint32_t *err, unsigned long long *__restrict__ hist, int hist_len, int lds_bins) {
  // per-library optical counts are collected in LDS first: the global counters are a handful of addresses, and a global atomic
  // on one address serialises at ~12 ns
  extern __shared__ unsigned int lds_opt[];  // [n_lib + 1], then [(n_lib + 1)][3][lds_bins]
  unsigned int *lds_h = lds_opt + (m.n_lib + 1);
  const int n_h = hist ? (m.n_lib + 1) * 3 * lds_bins : 0;
  for (int k = threadIdx.x; k <= m.n_lib + n_h; k += blockDim.x) lds_opt[k] = 0;
  __syncthreads();
  const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
  const uint2 gi = b < total ? ginfo[b] : make_uint2(0u, 0u);
  const uint32_t cnt = gi.x;
  uint32_t optical = 0;
  if (cnt > 300000u) {
    atomicOr(&err[2], 2u);
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
  if (optical || (hist && cnt >= 2 && cnt <= 300000u)) {
    const uint32_t owner = m.pwinner[gi.y];
    uint32_t a1, a2;
    order_ends(m, owner, m.mate[owner], a1, a2);
    const uint32_t lib = lib_row(m, a1);  // origin.aln1.LIBID() :381
    if (optical) atomicAdd(&lds_opt[lib], optical);
    if (hist && cnt >= 2 && cnt <= 300000u) {
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
  __syncthreads();
  for (int k = threadIdx.x; k <= m.n_lib; k += blockDim.x)
    if (lds_opt[k]) atomicAdd(&ctr[k * ELP_NCTR + 6], (unsigned long long)lds_opt[k]);
  for (int k = threadIdx.x; k < n_h; k += blockDim.x)
    if (lds_h[k]) atomicAdd(&hist[(size_t)(k / lds_bins) * hist_len + (k % lds_bins)], (unsigned long long)lds_h[k]);
}

// number of pair groups (= origins, the pairs that stayed best of their group) per library
__global__ __launch_bounds__(256) void k_origin_count(MxCols m, unsigned long long *__restrict__ origins) {
  extern __shared__ unsigned int lds_org[];  // [n_lib + 1]
  for (int k = threadIdx.x; k <= m.n_lib; k += blockDim.x) lds_org[k] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < m.n; i += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t rep = m.prep[i];
    if (rep == EMPTY || m.pwinner[rep] != (uint32_t)i) continue;
    uint32_t a1, a2;
    order_ends(m, (uint32_t)i, m.mate[i], a1, a2);
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
  unsigned long long *ctr;
  ELP_TRY(scratch(c, 0, (size_t)ncell + norg + nhist + 8, &ctr));
  unsigned long long *origins = ctr + ncell, *hist = hist_host ? origins + norg : nullptr;
  hipStream_t st = c->stream;
  ELP_HIP(c, hipMemsetAsync(ctr, 0, ((size_t)ncell + norg + nhist) * sizeof(unsigned long long), st));
  int lds_bins = hist_host ? std::min(hist_len, 32) : 0;
  if ((size_t)(c->n_lib + 1) * (1 + 3 * (size_t)lds_bins) * sizeof(unsigned int) > 32768) lds_bins = 0;
  if (n) {
    MxCols m{n, c->refid.p, c->flag.p, c->rgid.p, c->rg_lib.p, c->upos.p, c->qname_off.p, c->qname.p, c->mate.p, c->pair_slot.p, c->pair_winner.p, c->n_lib};
    const unsigned grid = blocks_for(n, 256);
    ELP_LAUNCH(c, "mx_counters", k_dup_counters, dim3(std::min(grid, 2048u)), dim3(256), ncell * sizeof(unsigned int), m, ctr);
    uint32_t *gs;
    ELP_TRY(scratch(c, 1, 3 * n + 16, &gs));
    uint32_t *gsize = gs, *goff = gs + n + 1, *gfill = gs + 2 * n + 2;
    ELP_HIP(c, hipMemsetAsync(gsize, 0, (n + 1) * sizeof(uint32_t), st));
    ELP_HIP(c, hipMemsetAsync(gfill, 0, n * sizeof(uint32_t), st));
    ELP_LAUNCH(c, "mx_opt_count", k_opt_count, dim3(grid), dim3(256), 0, m, gsize);
    ELP_LAUNCH(c, "mx_opt_plus_origin", k_opt_plus_origin, dim3(grid), dim3(256), 0, n, gsize);
    uint32_t total = 0;
    ELP_TRY(exclusive_scan_u32(c, gsize, goff, n + 1, &total));  // goff[n] = total
    if (total) {
      Member *members;
      uint32_t *parent, *mread;
      uint2 *ginfo;
      ELP_TRY(scratch(c, 2, (size_t)total + 4, &members));
      ELP_TRY(scratch(c, 3, (size_t)total + 4, &parent));
      ELP_TRY(scratch(c, 4, (size_t)total + 4, &mread));
      ELP_TRY(scratch(c, 5, (size_t)total + 4, &ginfo));
      ELP_HIP(c, hipMemsetAsync(ginfo, 0, (size_t)total * sizeof(uint2), st));
      ELP_LAUNCH(c, "mx_opt_slots", k_opt_slots, dim3(grid), dim3(256), 0, m, (const uint32_t *)goff, gfill, mread, ginfo);
      ELP_LAUNCH(c, "mx_opt_fill", k_opt_fill, dim3(blocks_for(total, 256)), dim3(256), 0, m, total, (const uint32_t *)mread, members, c->err_flag.p);
      ELP_LAUNCH(c, "mx_opt_eval", k_opt_eval, dim3(blocks_for(total, 128)), dim3(128),
                 (size_t)(c->n_lib + 1) * (1 + (hist ? 3 * (size_t)lds_bins : 0)) * sizeof(unsigned int), m, total, (const uint2 *)ginfo,
                 (const Member *)members, parent, (long long)dist, ctr, c->err_flag.p, hist, hist_len, lds_bins);
    }
    if (hist)
      ELP_LAUNCH(c, "mx_origin_count", k_origin_count, dim3(std::min(grid, 2048u)), dim3(256), (c->n_lib + 1) * sizeof(unsigned int), m, origins);
  }
  std::vector<unsigned long long> h((size_t)ncell + norg + nhist);
  ELP_HIP(c, hipMemcpyAsync(h.data(), ctr, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
  uint32_t e[4];
  ELP_TRY(fetch_err(c, e));
  if (e[2]) {
    ELP_HIP(c, hipMemsetAsync(c->err_flag.p + 2, 0, 4, st));
    if (e[2] & 1u) return set_error(c, ELP_ERR_DATA, "QNAME tile/x/y field is not an integer (reference: internal.ParseInt panics, filters/mark-optical-duplicates.go:55-61)");
    return set_error(c, ELP_ERR_UNSUPPORTED, "a duplicate set exceeds 300000 pairs (the reference truncates such lists, filters/mark-optical-duplicates.go:289-299)");
  }
  for (int l = 0; l <= c->n_lib; l++)
    for (int k = 0; k < ELP_NCTR; k++) counters_host[l * ELP_NCTR + k] = (int64_t)h[l * ELP_NCTR + k];
  for (int l = 0; l <= c->n_lib; l++) counters_host[l * ELP_NCTR + 1] /= 2;  // ReadPairsExamined counts reads, then halves (:504-506)
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

extern "C" int elp_dup_metrics(elp_ctx *c, int optical_pixel_distance, int64_t *counters) {
  if (!c || !counters) return ELP_ERR_ARG;
  ELP_HIP(c, hipSetDevice(c->device));
  return elp::metrics_impl(c, optical_pixel_distance, counters, nullptr, 0);
}

extern "C" int elp_dup_metrics_hist(elp_ctx *c, int optical_pixel_distance, int64_t *counters, int64_t *hist, int hist_len) {
  if (!c || !counters || !hist || hist_len < 2) return elp::set_error(c, ELP_ERR_ARG, "elp_dup_metrics_hist: bad arguments (hist_len >= 2)");
  ELP_HIP(c, hipSetDevice(c->device));
  return elp::metrics_impl(c, optical_pixel_distance, counters, hist, hist_len);
}
