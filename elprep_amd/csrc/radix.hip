// radix.hip — device-wide exclusive scan and stable LSD radix sort of (key64, val32) pairs for gfx950.
//
// This is the mechanism behind the coordinate sort (By.ParallelStableSort, sam/sam-types.go:639-641; pargo
// sort.StableSort in the reference) and the tie-break / grouping passes.  8-bit digits; digit positions whose
// histogram has a single non-empty bin are skipped (the 64-bit coordinate key of a human genome has ~5 live bytes).
//
// Per live pass ONE kernel, k_radix_scatter: wave64 ballot multisplit of a 4096-key tile (stable local ranks), the tile's
// global bin offsets by decoupled look-back over the tiles in front of it (tiles take tickets, so a tile only ever waits
// for tiles that are already running), digit-ordered staging in LDS, scatter.  The global digit histograms of all eight
// digit positions come from one sweep up front (k_radix_hist_all), which also decides the live passes.
// HBM traffic per key per pass: 12 (read) + 12 (write) = 24 B, + 8 B once for the histogram sweep.
#include "common.hpp"

namespace elp {

// ------------------------------------------------------------------ block scan helpers
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// exclusive scan across a 256-thread block; returns exclusive prefix of `v`, total in *total (all threads)
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *total, uint32_t *lds /* >= 5 words */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = wave_incl_scan(v);
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    uint32_t s = lds[i];
    if (i < w) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = 256 * SCAN_ITEMS;

__global__ __launch_bounds__(256) void k_scan_reduce(const uint32_t *__restrict__ in, uint32_t *__restrict__ sums, uint64_t n) {
  __shared__ uint32_t lds[8];
  uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++)
    if (base + i < n) s += in[base + i];
  uint32_t tot;
  (void)block_excl_scan_256(s, &tot, lds);
  if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_scan_down(const uint32_t *__restrict__ in, uint32_t *__restrict__ out,
                                                   const uint32_t *__restrict__ block_base, uint64_t n) {
  __shared__ uint32_t lds[8];
  uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
  uint32_t v[SCAN_ITEMS];
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    v[i] = (base + i < n) ? in[base + i] : 0u;
    s += v[i];
  }
  uint32_t tot;
  uint32_t ex = block_excl_scan_256(s, &tot, lds) + (block_base ? block_base[blockIdx.x] : 0u);
#pragma unroll
  for (int i = 0; i < SCAN_ITEMS; i++) {
    if (base + i < n) out[base + i] = ex;
    ex += v[i];
  }
}

static int scan_levels(elp_ctx *c, const uint32_t *in, uint32_t *out, uint64_t n, uint32_t *total_dev) {
  if (n == 0) {
    if (total_dev) ELP_HIP(c, hipMemsetAsync(total_dev, 0, 4, c->stream));
    return 0;
  }
  // sizes of each level
  std::vector<uint64_t> cnt;  // cnt[0] = n, cnt[k+1] = blocks of level k
  cnt.push_back(n);
  while (true) {
    uint64_t nb = (cnt.back() + SCAN_TILE - 1) / SCAN_TILE;
    cnt.push_back(nb);
    if (nb <= 1) break;
  }
  size_t need = 0;
  for (size_t k = 1; k < cnt.size(); k++) need += cnt[k] + 4;
  uint32_t *pool;
  ELP_TRY(ensure(c, c->scan_pool, need + 16));  // (its own buffer, as radix_hist)
  pool = c->scan_pool.p;
  std::vector<uint32_t *> sums(cnt.size(), nullptr);
  size_t off = 0;
  for (size_t k = 1; k < cnt.size(); k++) { sums[k] = pool + off; off += cnt[k] + 4; }
  // up-sweep
  const uint32_t *src = in;
  for (size_t k = 0; k + 1 < cnt.size(); k++) {
    ELP_LAUNCH(c, "scan_reduce", k_scan_reduce, dim3((unsigned)cnt[k + 1]), dim3(256), 0, src, sums[k + 1], cnt[k]);
    src = sums[k + 1];
  }
  // top level has exactly one element = grand total
  size_t top = cnt.size() - 1;
  if (total_dev) ELP_HIP(c, hipMemcpyAsync(total_dev, sums[top], 4, hipMemcpyDeviceToDevice, c->stream));
  ELP_HIP(c, hipMemsetAsync(sums[top], 0, 4, c->stream));
  // down-sweep: level k's sums become exclusive prefixes using level k+1
  for (size_t k = top - 1; k >= 1; k--) {
    ELP_LAUNCH(c, "scan_down", k_scan_down, dim3((unsigned)cnt[k + 1]), dim3(256), 0, (const uint32_t *)sums[k], sums[k],
               (const uint32_t *)sums[k + 1], cnt[k]);
  }
  ELP_LAUNCH(c, "scan_down", k_scan_down, dim3((unsigned)cnt[1]), dim3(256), 0, in, out, (const uint32_t *)sums[1], n);
  return 0;
}

int exclusive_scan_u32(elp_ctx *c, const uint32_t *in, uint32_t *out, uint64_t n, uint32_t *total_host) {
  uint32_t *tot_dev = nullptr;
  if (total_host) tot_dev = c->err_flag.p + 3;  // word 3 of the error block doubles as the scan-total mailbox
  ELP_TRY(scan_levels(c, in, out, n, tot_dev));
  if (total_host) {
    ELP_HIP(c, hipMemcpyAsync(total_host, tot_dev, 4, hipMemcpyDeviceToHost, c->stream));
    ELP_HIP(c, elp::stream_wait(c->stream));
    ELP_HIP(c, hipMemsetAsync(tot_dev, 0, 4, c->stream));
  }
  return 0;
}

// ------------------------------------------------------------------ radix sort
constexpr int RS_THREADS = 256;
constexpr int RS_ITEMS = 16;
constexpr int RS_TILE = RS_THREADS * RS_ITEMS;  // 4096 keys per workgroup

// one sweep: histograms of all 8 digit positions (decides which passes are live)
template <int ND>
__global__ __launch_bounds__(256) void k_radix_hist_all(const uint64_t *__restrict__ keys, uint64_t n, unsigned long long *__restrict__ ghist /* [8][256] */,
                                                        const uint32_t *__restrict__ n_dev) {
  __shared__ uint32_t h[8][256];
  if (n_dev) n = *n_dev;  // the length is a device-side count (no read-back): `n` was its upper bound
  for (int i = threadIdx.x; i < 8 * 256; i += 256) (&h[0][0])[i] = 0;
  __syncthreads();
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    uint64_t k = keys[i];
    const unsigned long long act = __ballot(1);
    const int first = __ffsll((long long)act) - 1;
#pragma unroll
    for (int d = 0; d < ND; d++) {
      // constant digit positions (most of the high bytes) would be a 64-way same-address LDS conflict: aggregate per wave
      uint32_t dg = (uint32_t)(k >> (8 * d)) & 0xFF;
      uint32_t d0 = __shfl(dg, first, 64);
      if (__all(dg == d0)) {
        if ((int)(threadIdx.x & 63) == first) atomicAdd(&h[d][d0], (uint32_t)__popcll(act));
      } else {
        atomicAdd(&h[d][dg], 1u);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < ND * 256; i += 256) {
    uint32_t v = (&h[0][0])[i];
    if (v) atomicAdd(&ghist[i], (unsigned long long)v);
  }
}

// stable scatter.  Element order inside a tile: wave-major, then round, then lane (== index order, because each wave
// owns a contiguous 1024-key sub-tile and reads it 64 keys per round).  Keys (then values) are first reordered by digit inside
// the tile through LDS, so that the global stores of a workgroup are runs of consecutive addresses per digit (on average 16
// keys = 128 bytes with 4096-key tiles and 256 bins) instead of 64 scattered 8-byte stores per wave instruction.
//
// Look-back word of (tile, digit): [63:34] pass epoch, [33:32] 1 = the tile's own count, 2 = inclusive count over tiles 0..tile,
// [31:0] the count.  Words of earlier passes carry an older epoch and read as "not there yet", so the array is never cleared.
constexpr unsigned long long RS_AGG = 1ull << 32, RS_INCL = 2ull << 32;

// PAIRS = false: no values - the sort's elements are single words, e.g. key << b | index (`fuse` = b > 0: this pass reads bare keys and
// appends the element's index: the first pass of such a sort); 16 instead of 24 bytes per element and pass.
template <int THREADS, bool PAIRS>
__global__ __launch_bounds__(THREADS, THREADS >= 512 ? 4 : 1) void k_radix_scatter_t(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                              uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint64_t n,
                                                              int shift, const unsigned long long *__restrict__ ghist /* [256] of this digit */,
                                                              unsigned long long *state, uint32_t epoch, uint32_t *ticket,
                                                              uint32_t *err, const uint32_t *__restrict__ n_dev, int fuse, RadixBounds bnd) {
  constexpr int WAVES = THREADS / 64, TILE = THREADS * RS_ITEMS;  // 4096 keys (256 threads) or 8192 (512: runs of twice the length per digit)
  __shared__ uint32_t cnt[WAVES][256];
  __shared__ uint32_t gbase[256];   // global position of the tile's first key with digit d, minus its position inside the sorted tile
  __shared__ uint32_t scan_lds[16];
  // the tile in digit order: keys first, then (as uint32) the values - 32 KB static, or 64 KB of dynamic LDS
  extern __shared__ __attribute__((aligned(16))) uint64_t sbuf_dyn[];
  __shared__ uint64_t sbuf_static[THREADS == 256 ? RS_TILE : 1];
  uint64_t *const sbuf = THREADS == 256 ? sbuf_static : sbuf_dyn;
  __shared__ uint32_t my_tile;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (threadIdx.x == 0) my_tile = atomicAdd(ticket, 1u);  // tiles are numbered in the order they start (the host zeroes the counter)
  for (int i = threadIdx.x; i < WAVES * 256; i += THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const uint32_t tile = my_tile;
  const uint64_t tbase = (uint64_t)tile * TILE;
  if (n_dev) {  // device-side length: the launch covers its upper bound, tiles behind the end leave at once (nobody looks back at them)
    n = *n_dev;
    if (tbase >= n) return;
  }
  const uint64_t wbase = tbase + (uint64_t)w * (64 * RS_ITEMS);
  const uint32_t tile_n = (uint32_t)((n - tbase) < (uint64_t)TILE ? (n - tbase) : (uint64_t)TILE);
  uint64_t k[RS_ITEMS];
  uint32_t v[PAIRS ? RS_ITEMS : 1];
  uint32_t pos[RS_ITEMS];
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    uint64_t i = wbase + (uint64_t)r * 64 + lane;
    bool valid = i < n;
    k[r] = valid ? keys[i] : ~0ull;
    if (!PAIRS && fuse && valid) k[r] = (k[r] << fuse) | i;
    if (PAIRS) v[r] = valid ? (vals ? vals[i] : (uint32_t)i) : 0u;  // no value array: the values are the indices (first pass of a sort)
    uint32_t d = (uint32_t)(k[r] >> shift) & 0xFF;
    // peers = lanes of this wave holding the same digit this round (invalid lanes form their own class)
    unsigned long long peers = __ballot(valid);
    peers = valid ? peers : ~peers;
#pragma unroll
    for (int b = 0; b < 8; b++) {
      unsigned long long m = __ballot((d >> b) & 1);
      peers &= ((d >> b) & 1) ? m : ~m;
    }
    uint32_t before = (uint32_t)__popcll(peers & lt_mask);
    int leader = __ffsll((long long)peers) - 1;
    uint32_t old = 0;
    if (valid && lane == leader) {
      old = cnt[w][d];
      cnt[w][d] = old + (uint32_t)__popcll(peers);
    }
    old = __shfl(old, leader, 64);
    pos[r] = old + before;  // rank among the wave's keys with this digit
  }
  __syncthreads();
  // thread t owns digit t: position of the digit inside the sorted tile (exclusive scan over digits), then per-wave starts
  uint32_t tot = 0, tstart;
  const unsigned long long tag = (unsigned long long)epoch << 34;
  const bool owner = THREADS == 256 || threadIdx.x < 256;  // the first 256 threads own a digit each
  const uint32_t dg = THREADS == 256 ? threadIdx.x : (threadIdx.x & 255u);
  unsigned long long *mine = state + (size_t)tile * 256 + dg;
  {
    if (owner) {
#pragma unroll
      for (int i = 0; i < WAVES; i++) tot += cnt[i][dg];
      // the tile's own count goes out as early as possible, the look-back comes as late as possible (behind the LDS reorder of the
      // keys): the tiles in front get that time to publish
      __hip_atomic_store(mine, tag | (tile == 0 ? RS_INCL : RS_AGG) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t all;
    tstart = block_excl_scan_256(tot, &all, scan_lds);  // (waves 4 .. 7 of a 512-thread tile add zeros behind the 256 digits)
    if (owner) {
      uint32_t run = tstart;
#pragma unroll
      for (int i = 0; i < WAVES; i++) {
        uint32_t t = cnt[i][dg];
        cnt[i][dg] = run;
        run += t;
      }
    }
  }
  __syncthreads();
  // keys into digit order
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    uint64_t i = wbase + (uint64_t)r * 64 + lane;
    if (i < n) {
      uint32_t d = (uint32_t)(k[r] >> shift) & 0xFF;
      pos[r] += cnt[w][d];
      sbuf[pos[r]] = k[r];
    }
  }
  {
    // first output position of digit t: keys with a smaller digit (global histogram) + keys with digit t in the tiles in front
    uint32_t dall;
    const uint32_t dbase = block_excl_scan_256(owner ? (uint32_t)ghist[dg] : 0u, &dall, scan_lds);
    uint32_t prefix = 0;
    if (tile != 0 && owner) {
      // walk back over the tiles in front, eight words per round trip (device-scope loads cross the XCDs: ~1 us each)
      constexpr int LB = 8;
      uint32_t back = 1, spins = 0;  // next tile to look at: tile - back
      bool done = false;
      while (!done) {
        unsigned long long v[LB];
#pragma unroll
        for (int i = 0; i < LB; i++) {
          const uint32_t b = back + (uint32_t)i;
          v[i] = b <= tile ? __hip_atomic_load(mine - (size_t)b * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (tag | RS_INCL);
        }
#pragma unroll
        for (int i = 0; i < LB; i++) {
          if (done || (v[i] >> 34) != epoch) break;  // not published yet: poll again from here
          prefix += (uint32_t)v[i];
          back++;
          done = (v[i] & RS_INCL) != 0;
        }
        if (!done) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins == (1u << 22)) { atomicOr(err, 256u); break; }  // seconds: something is broken, do not hang the device
        }
      }
      __hip_atomic_store(mine, tag | RS_INCL | (prefix + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (owner) gbase[dg] = dbase + prefix - tstart;
  }
  __syncthreads();
  uint32_t dst[RS_ITEMS];
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const uint32_t j = (uint32_t)r * THREADS + threadIdx.x;
    if (j < tile_n) {
      const uint64_t key = sbuf[j];
      dst[r] = gbase[(uint32_t)(key >> shift) & 0xFF] + j;
      keys_out[dst[r]] = key;
      if (PAIRS && bnd.start) {
        // the LAST pass of a sort whose caller wants the bounds of the buckets its (sorted) low key bits define: inside the tile the
        // elements of a bucket are neighbours (the tile is in digit order, and the passes before sorted what is left of the bucket's
        // bits); the first / last one of such a run reports its output position - a few hundred atomics per tile instead of a pass
        // over the sorted array (k_pair_bounds)
        const uint32_t b = ((uint32_t)key & bnd.mask) >> bnd.shift;
        const bool first = j == 0 || (((uint32_t)sbuf[j - 1] & bnd.mask) >> bnd.shift) != b;
        const bool last = j + 1 == tile_n || (((uint32_t)sbuf[j + 1] & bnd.mask) >> bnd.shift) != b;
        if (first) atomicMin(&bnd.start[b], dst[r]);
        if (last) atomicMin(&bnd.nend[b], ~(dst[r] + 1u));
      }
    }
  }
  if (!PAIRS) return;
  __syncthreads();
  // values through the same buffer
  uint32_t *sval = reinterpret_cast<uint32_t *>(sbuf);
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    uint64_t i = wbase + (uint64_t)r * 64 + lane;
    if (i < n) sval[pos[r]] = v[PAIRS ? r : 0];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < RS_ITEMS; r++) {
    const uint32_t j = (uint32_t)r * THREADS + threadIdx.x;
    if (j < tile_n) vals_out[dst[r]] = sval[j];
  }
}

// state words, ticket counters and the epoch of one more pass over `ntiles` tiles
static int radix_pass_setup(elp_ctx *c, uint32_t ntiles) {
  if ((size_t)ntiles * 256 > c->radix_state.cap) {
    ELP_TRY(ensure(c, c->radix_state, (size_t)ntiles * 256));
    ELP_HIP(c, hipMemsetAsync(c->radix_state.p, 0, c->radix_state.cap * sizeof(unsigned long long), c->stream));
    c->radix_epoch = 0;
  }
  return 0;  // (the ticket counters, one per digit position, live behind the caller's histograms: one fill clears both)
}
static int radix_next_epoch(elp_ctx *c) {
  if (++c->radix_epoch >= (1u << 30)) {  // the tag wrapped: forget every word
    ELP_HIP(c, hipMemsetAsync(c->radix_state.p, 0, c->radix_state.cap * sizeof(unsigned long long), c->stream));
    c->radix_epoch = 1;
  }
  return 0;
}

// one pass.  Tiles of 16384 keys (1024 threads, 128 KB of LDS: one workgroup per CU) for arrays of 16 M elements and more, of 8192 from 4 M
// on - a digit's keys leave a tile as runs of four times / twice the length: 50 M pairs, four passes: 2.14 ms (4096), 2.01 (8192), 1.81
// (16384) -, of 4096 for the short ones (more tiles than CUs matter more there); elp_set_tuning "radix_tile": 1 / 2 / 3 = 4096 / 8192 /
// 16384 always
static int radix_tile_shift(const elp_ctx *c, uint64_t n) {  // tile = 4096 keys << shift
  if (c->tune.radix_tile >= 1 && c->tune.radix_tile <= 3) return c->tune.radix_tile - 1;
  return n >= (16ull << 20) ? 2 : n >= (4ull << 20) ? 1 : 0;
}
static bool radix_big_tiles(const elp_ctx *c, uint64_t n) { return radix_tile_shift(c, n) == 1; }
static uint32_t radix_tiles(const elp_ctx *c, uint64_t n) {
  const uint64_t tile = (uint64_t)RS_TILE << radix_tile_shift(c, n);
  return (uint32_t)((n + tile - 1) / tile);
}
template <bool PAIRS>
static int radix_scatter_launch_t(elp_ctx *c, uint64_t n, const uint64_t *kin, const uint32_t *vin, uint64_t *kdst, uint32_t *vdst, int shift,
                                  const unsigned long long *ghist, uint32_t *ticket, const uint32_t *n_dev, int fuse, RadixBounds bnd = RadixBounds{}) {
  const uint32_t ntiles = radix_tiles(c, n);
  const int ts = radix_tile_shift(c, n);
  if (ts == 2) {  // one workgroup of 16 waves per CU
    const size_t dyn = (size_t)4 * RS_TILE * sizeof(uint64_t);
    ELP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_radix_scatter_t<1024, PAIRS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    ELP_LAUNCH(c, "radix_scatter", (k_radix_scatter_t<1024, PAIRS>), dim3(ntiles), dim3(1024), dyn, kin, vin, kdst, vdst, n, shift, ghist, c->radix_state.p,
               c->radix_epoch, ticket, c->err_flag.p, n_dev, fuse, bnd);
  } else if (ts == 1) {
    const size_t dyn = (size_t)2 * RS_TILE * sizeof(uint64_t);
    ELP_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void *>(&k_radix_scatter_t<512, PAIRS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    ELP_LAUNCH(c, "radix_scatter", (k_radix_scatter_t<512, PAIRS>), dim3(ntiles), dim3(512), dyn, kin, vin, kdst, vdst, n, shift, ghist, c->radix_state.p,
               c->radix_epoch, ticket, c->err_flag.p, n_dev, fuse, bnd);
  } else {
    ELP_LAUNCH(c, "radix_scatter", (k_radix_scatter_t<256, PAIRS>), dim3(ntiles), dim3(256), 0, kin, vin, kdst, vdst, n, shift, ghist, c->radix_state.p,
               c->radix_epoch, ticket, c->err_flag.p, n_dev, fuse, bnd);
  }
  return 0;
}
static int radix_scatter_launch(elp_ctx *c, uint64_t n, const uint64_t *kin, const uint32_t *vin, uint64_t *kdst, uint32_t *vdst, int shift,
                                const unsigned long long *ghist, uint32_t *ticket, const uint32_t *n_dev, RadixBounds bnd = RadixBounds{}) {
  return radix_scatter_launch_t<true>(c, n, kin, vin, kdst, vdst, shift, ghist, ticket, n_dev, 0, bnd);
}

// The coordinate sort's own form: elements key << idx_bits | index in ONE word (the sort's key has ~31 live bits, an index 26), sorted on the
// key's digits by stable passes - 16 bytes of traffic per element and pass instead of 24, and a tile's runs per digit are twice as
// long in bytes.  `keycol` holds the bare keys (key_bits live bits); the result (n words) ends in *out (buf0 or buf1).
int radix_sort_fused(elp_ctx *c, const uint64_t *keycol, uint64_t n, int key_bits, int idx_bits, uint64_t *buf0, uint64_t *buf1, uint64_t **out) {
  if (n >= 0xFFFFFFFFull || key_bits + idx_bits > 64 || key_bits < 1 || idx_bits < 1) return set_error(c, ELP_ERR_UNSUPPORTED, "radix sort: bad size");
  const int ndigits = (key_bits + 7) / 8;
  unsigned long long *ghist;
  ELP_TRY(ensure(c, c->radix_hist, 8 * 256 + 4));  // (its own buffer: callers keep live data in the scratch slots across a sort)
  ghist = c->radix_hist.p;
  ELP_HIP(c, hipMemsetAsync(ghist, 0, (8 * 256 + 4) * sizeof(unsigned long long), c->stream));
  uint32_t *ticket = reinterpret_cast<uint32_t *>(ghist + 8 * 256);
  const unsigned hb = (unsigned)std::min<uint64_t>((n + 255) / 256, 2048);
  if (ndigits <= 2) ELP_LAUNCH(c, "radix_hist_all", k_radix_hist_all<2>, dim3(hb), dim3(256), 0, keycol, n, ghist, (const uint32_t *)nullptr);
  else if (ndigits <= 4) ELP_LAUNCH(c, "radix_hist_all", k_radix_hist_all<4>, dim3(hb), dim3(256), 0, keycol, n, ghist, (const uint32_t *)nullptr);
  else ELP_LAUNCH(c, "radix_hist_all", k_radix_hist_all<8>, dim3(hb), dim3(256), 0, keycol, n, ghist, (const uint32_t *)nullptr);
  ELP_TRY(radix_pass_setup(c, radix_tiles(c, n)));
  uint64_t *src = buf1, *dst = buf0;  // (the first pass reads the key column)
  for (int d = 0; d < ndigits; d++) {
    ELP_TRY(radix_next_epoch(c));
    ELP_TRY(radix_scatter_launch_t<false>(c, n, d == 0 ? keycol : src, nullptr, dst, nullptr, idx_bits + 8 * d, (const unsigned long long *)(ghist + d * 256),
                                          ticket + d, nullptr, d == 0 ? idx_bits : 0));
    std::swap(src, dst);
  }
  *out = src;
  return 0;
}

// first_src (optional): the keys are read from there by the first pass (and `keys` is only written); identity_vals: the values are
// 0 .. n-1 and `vals` is only written
int radix_sort_pairs_low(elp_ctx *c, uint64_t *keys, uint32_t *vals, uint64_t *keys_tmp, uint32_t *vals_tmp, uint64_t n, int ndigits,
                         uint64_t **keys_out, uint32_t **vals_out, const uint64_t *first_src, bool identity_vals, const uint32_t *n_dev, RadixBounds bnd) {
  *keys_out = keys;
  *vals_out = vals;
  if ((ndigits <= 0 || (n < 2 && !n_dev)) && !first_src && !identity_vals) return 0;
  if (ndigits <= 0) ndigits = 1;  // a pass is needed to materialise keys / values
  if (n >= 0xFFFFFFFFull || ndigits > 8) return set_error(c, ELP_ERR_UNSUPPORTED, "radix sort: bad size");
  unsigned long long *ghist;
  ELP_TRY(ensure(c, c->radix_hist, 8 * 256 + 4));  // (its own buffer: callers keep live data in the scratch slots across a sort)
  ghist = c->radix_hist.p;
  ELP_HIP(c, hipMemsetAsync(ghist, 0, (8 * 256 + 4) * sizeof(unsigned long long), c->stream));
  uint32_t *ticket = reinterpret_cast<uint32_t *>(ghist + 8 * 256);
  const unsigned hb = (unsigned)std::min<uint64_t>((n + 255) / 256, 2048);
  // histograms of the digit positions that are sorted, not of all eight
  if (ndigits <= 2) ELP_LAUNCH(c, "radix_hist_all", k_radix_hist_all<2>, dim3(hb), dim3(256), 0, (const uint64_t *)(first_src ? first_src : keys), n, ghist, n_dev);
  else if (ndigits <= 4) ELP_LAUNCH(c, "radix_hist_all", k_radix_hist_all<4>, dim3(hb), dim3(256), 0, (const uint64_t *)(first_src ? first_src : keys), n, ghist, n_dev);
  else ELP_LAUNCH(c, "radix_hist_all", k_radix_hist_all<8>, dim3(hb), dim3(256), 0, (const uint64_t *)(first_src ? first_src : keys), n, ghist, n_dev);
  ELP_TRY(radix_pass_setup(c, radix_tiles(c, n)));
  uint64_t *ksrc = keys, *kdst = keys_tmp;
  uint32_t *vsrc = vals, *vdst = vals_tmp;
  for (int d = 0; d < ndigits; d++) {
    ELP_TRY(radix_next_epoch(c));
    const uint64_t *kin = (d == 0 && first_src) ? first_src : ksrc;
    const uint32_t *vin = (d == 0 && identity_vals) ? nullptr : vsrc;
    ELP_TRY(radix_scatter_launch(c, n, kin, vin, kdst, vdst, 8 * d, (const unsigned long long *)(ghist + d * 256), ticket + d, n_dev,
                                 d == ndigits - 1 ? bnd : RadixBounds{}));
    std::swap(ksrc, kdst);
    std::swap(vsrc, vdst);
  }
  *keys_out = ksrc;
  *vals_out = vsrc;
  return 0;
}

int radix_sort_pairs(elp_ctx *c, uint64_t *keys, uint32_t *vals, uint64_t *keys_tmp, uint32_t *vals_tmp, uint64_t n, uint64_t **keys_out,
                     uint32_t **vals_out) {
  *keys_out = keys;
  *vals_out = vals;
  if (n < 2) return 0;
  if (n >= 0xFFFFFFFFull) return set_error(c, ELP_ERR_UNSUPPORTED, "radix sort: more than 2^32-1 elements");
  unsigned long long *ghist;
  ELP_TRY(ensure(c, c->radix_hist, 8 * 256 + 4));  // (its own buffer: callers keep live data in the scratch slots across a sort)
  ghist = c->radix_hist.p;
  ELP_HIP(c, hipMemsetAsync(ghist, 0, (8 * 256 + 4) * sizeof(unsigned long long), c->stream));
  uint32_t *ticket = reinterpret_cast<uint32_t *>(ghist + 8 * 256);
  unsigned hb = (unsigned)std::min<uint64_t>((n + 255) / 256, 2048);
  ELP_LAUNCH(c, "radix_hist_all", k_radix_hist_all<8>, dim3(hb), dim3(256), 0, (const uint64_t *)keys, n, ghist, (const uint32_t *)nullptr);
  unsigned long long hh[8 * 256];
  ELP_HIP(c, hipMemcpyAsync(hh, ghist, sizeof hh, hipMemcpyDeviceToHost, c->stream));
  ELP_HIP(c, elp::stream_wait(c->stream));  // (a look-back timeout of any pass is reported by the callers, behind their last pass)
  ELP_TRY(radix_pass_setup(c, radix_tiles(c, n)));
  uint64_t *ksrc = keys, *kdst = keys_tmp;
  uint32_t *vsrc = vals, *vdst = vals_tmp;
  for (int d = 0; d < 8; d++) {
    bool live = true;
    for (int b = 0; b < 256; b++)
      if (hh[d * 256 + b] == n) { live = false; break; }
    if (!live) continue;
    ELP_TRY(radix_next_epoch(c));
    ELP_TRY(radix_scatter_launch(c, n, ksrc, vsrc, kdst, vdst, 8 * d, (const unsigned long long *)(ghist + d * 256), ticket + d, nullptr));
    std::swap(ksrc, kdst);
    std::swap(vsrc, vdst);
  }
  *keys_out = ksrc;
  *vals_out = vsrc;
  return 0;
}

}  // namespace elp
