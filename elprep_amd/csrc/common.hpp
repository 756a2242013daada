// common.hpp — context, device vectors, launch/profiling helpers shared by the HIP translation units of
// libelprep_hip.so (gfx950 only; wave64).
#pragma once

#include <hip/hip_runtime.h>

#include <sched.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/elprep_hip.h"
#include "../../include/elprep_hip_debug.h"

namespace elp {

constexpr int WAVE = 64;

// SAM FLAG bits (sam/sam-types.go:484-520)
enum : uint16_t {
  F_MULTIPLE = 0x1, F_PROPER = 0x2, F_UNMAPPED = 0x4, F_NEXT_UNMAPPED = 0x8, F_REVERSED = 0x10, F_NEXT_REVERSED = 0x20,
  F_FIRST = 0x40, F_LAST = 0x80, F_SECONDARY = 0x100, F_QCFAILED = 0x200, F_DUPLICATE = 0x400, F_SUPPLEMENTARY = 0x800
};

// BAM CIGAR op codes: index into "MIDNSHP=X"
enum : uint32_t { OP_M = 0, OP_I = 1, OP_D = 2, OP_N = 3, OP_S = 4, OP_H = 5, OP_P = 6, OP_EQ = 7, OP_X = 8 };

__host__ __device__ inline bool op_consumes_read(uint32_t op) { return op == OP_M || op == OP_I || op == OP_S || op == OP_EQ || op == OP_X; }
__host__ __device__ inline bool op_consumes_ref(uint32_t op) { return op == OP_M || op == OP_D || op == OP_N || op == OP_EQ || op == OP_X; }

// ELP_DEBUG_GUARD=1 (tests): every device buffer of a context gets 4 KB behind its end filled with a pattern; the pattern is checked when
// the buffer is released or regrown (and by elp_debug_check_guards): a kernel that writes past the end of its buffer aborts the process
// with the buffer's size on stderr instead of corrupting a neighbour.  ctx.hip.
size_t debug_guard_bytes();
void debug_guard_arm(void *p, size_t payload_bytes);
void debug_guard_release(void *p);
int debug_guard_check_all();

template <class T>
struct DVec {
  T *p = nullptr;
  size_t cap = 0;  // elements
  DVec() = default;
  DVec(const DVec &) = delete;
  DVec &operator=(const DVec &) = delete;
  ~DVec() { release(); }
  void release() {
    if (p) { debug_guard_release(p); (void)hipFree(p); }
    p = nullptr;
    cap = 0;
  }
  size_t bytes() const { return cap * sizeof(T); }
};

struct ProfPending {
  int name_id;
  hipEvent_t a, b;
};

}  // namespace elp

struct elp_ctx {
  int device = 0;
  int n_cu = 256;  // compute units of the device (grid sizing)
  hipStream_t stream = nullptr;
  std::string err;
  std::mutex stage_mu;

  // header
  bool have_header = false;
  int32_t n_ref = 0, n_rg = 0, n_lib = 0, n_cov = 0;
  std::vector<int32_t> h_ref_len;
  std::vector<uint16_t> h_rg_lib, h_rg_cov;
  elp::DVec<int32_t> ref_len;
  elp::DVec<uint16_t> rg_lib, rg_cov;

  // staged columns
  static constexpr uint64_t SEQ_FRONT = 16;  // the SEQ column starts this many bytes into its allocation: the per-base kernels load the
                                              // window of a block from one byte in front of it (flat2.hpp bodies), also for the first read
  uint64_t n = 0, qname_bytes = 0, cigar_ops = 0, seq_bytes = SEQ_FRONT, qual_bytes = 0;
  uint64_t n_sr = 0;  // staged records that carry the sr tag (dropped by RemoveOptionalReads) or were rejected by elp_filter_records:
                      // behind everything else in the permutation
  uint64_t n_filtered = 0;  // of those, rejected by elp_filter_records (state 2 in the has_sr column: no duplicate marking either)
  elp::DVec<int32_t> refid, pos, next_refid, pnext, tlen;
  elp::DVec<uint16_t> flag, rgid, split;
  elp::DVec<uint8_t> mapq, has_sr;
  elp::DVec<uint32_t> l_seq;
  elp::DVec<uint64_t> qname_off, cigar_off, seq_off, qual_off;  // n+1
  elp::DVec<uint8_t> qname, seq4, qual;
  elp::DVec<uint32_t> cigar;
  elp::DVec<uint64_t> stage_tmp;  // offsets of the batch being staged
  uint32_t max_qname_len = 0, max_l_seq = 0;
  uint32_t max_split = 0;  // largest staged split id
  uint32_t max_pos = 0;  // largest staged POS (as uint32): width of the POS field of the coordinate-sort key
  int key_bits = 64;     // live low bits of the coordinate-sort key column (set with it, ensure_adapted)

  // derived state
  bool adapted = false, sorted = false, marked = false;
  bool radix_check_pending = false;  // radix passes were queued whose look-back timeout bit nobody has read yet (fetch_err)
  bool adapt_bad_qual = false;  // adapt_score met a quality > 93 in a duplicate-marking candidate
  bool adapt_pending = false;   // ... or may have: its error word (adapt_err) has not been read yet
  elp::DVec<uint32_t> adapt_err;  // [0] the score kernel's error word, [2 .. 5] the quality values k_score_uniform's sampled groups saw
  bool adapt_sampled = false;      // the score kernel of this adapt stage sampled the quality values ...
  bool adapt_qmask_valid = false;  // ... and they have been read (adapt_note)
  unsigned long long adapt_qmask[2] = {0, 0};
  elp::DVec<uint2> apply_recs;     // ApplyBQSR's per-read records (apply_rec.hpp), written by k_score_uniform ...
  bool apply_recs_valid = false;   // ... of this adapt stage, for reads of apply_recs_lmax bases
  int apply_recs_lmax = 0;
  bool have_qual_present = false;
  unsigned long long qual_present[2] = {0, 0};  // bit q set if quality value q was seen in a sample of the QUAL column (sizing hint for the BQSR tables)
  elp::DVec<int32_t> upos, score;
  elp::DVec<uint64_t> qbounds;  // per record: 1 + index of the last quality > 2 (low 32 bits; 0 = none) and index of the first one (high): the
                                // low-quality-tail bounds of computeStrandedClippedSeq (bqsr.go:316-332) on the full read; valid when adapted
  elp::DVec<uint64_t> key;      // coordinate sort keys, staging order
  elp::DVec<uint32_t> perm;     // sorted position -> staging index
  elp::DVec<uint32_t> err_flag; // device-side error word(s)
  elp::DVec<uint32_t> tile_first;  // flat.hpp tile index over the QUAL column
  void *h_pinned = nullptr;        // pinned host staging (BQSR tables come back in one copy)
  size_t h_pinned_cap = 0;
  elp::DVec<unsigned long long> radix_state;  // radix.hip: per (tile, digit) look-back words, tagged with the pass epoch
  elp::DVec<uint32_t> radix_ticket;           // radix.hip: tile ticket counters
  // radix.hip's own workspaces.  Until round 5 the digit histograms lived in scratch slot 6 and the scan's partial sums in slot 7 - the
  // slots in which mark duplicates keeps its candidate codes and its pair list ACROSS its radix passes, and the record exchange its
  // headers across the gather's scans: a context whose slot was larger than the asking call needed (a small read set behind a large one)
  // had those overwritten; a smaller slot was reallocated under the live pointers.
  elp::DVec<unsigned long long> radix_hist;   // digit histograms of a sort + its ticket word
  elp::DVec<uint32_t> scan_pool;              // partial sums of exclusive_scan_u32's levels
  elp::DVec<unsigned long long> xchg_hdr;     // exchange.hip: piece headers and verdicts
  elp::DVec<uint32_t> tie_live;               // sort.hip: bit j = byte j of the comparator string differs among the members of large runs
  elp::DVec<uint32_t> md_ctr;                 // markdup.hip: device counters read back with one copy (0: true fragments listed)
  static constexpr uint32_t MAX_QNAME = 1000; // staged QNAME length limit; the comparator string (QNAME + 15 bytes) fits TIE_LIVE_WORDS * 32 bits
  static constexpr uint32_t TIE_LIVE_WORDS = 32;
  uint32_t radix_epoch = 0;
  uint64_t flat_index_n = 0, flat_index_bytes = 0;
  uint32_t uniform_len = 0;  // > 0: every staged read has this many bases and the offset columns are arithmetic (ensure_uniform_len)
  uint64_t uniform_n = ~0ull, uniform_bytes = ~0ull;

  // mark-duplicates results kept for the metrics pass
  elp::DVec<uint32_t> mate;        // per record: staging index of its mate if the two form a pair (classifyPair), else 0xFFFFFFFF
  elp::DVec<uint32_t> pair_win;    // per record: for the owner (the later-arriving mate) of a pair that LOST its key's tournament, the owner of
                                   // the winning pair; 0xFFFFFFFF for every other record

  // BQSR inputs
  std::vector<uint8_t *> h_ref_seq;  // device pointers per refid
  std::vector<int64_t> h_ref_seq_len;
  std::vector<int32_t *> h_sites;    // device pointers per refid, [n][2]
  std::vector<int64_t> h_n_sites;
  std::vector<uint8_t> ref_flags_dirty;  // per refid: the known-site flags inside the packed reference (bit 2 of a base's nibble) are stale
  std::vector<uint32_t *> h_site_idx;  // device pointers per refid: per 64-bp bucket b the first site whose end is >= 64 b
  elp::DVec<uint8_t *> d_ref_seq;
  elp::DVec<int64_t> d_ref_seq_len;
  elp::DVec<int32_t *> d_sites;
  elp::DVec<uint32_t *> d_site_idx;
  elp::DVec<int64_t> d_n_sites;
  bool bqsr_ptrs_dirty = true;

  // BQSR count tables in HBM (elp_bqsr_gather_device): [n_cov][94][2] | [n_cov][94][2 max_cycle + 1][2] | [n_cov][94][16][2], then
  // TABLES_TAIL spare values (the duplication counters ride behind the tables through the all-reduce, group.hip)
  static constexpr size_t TABLES_TAIL = 4096;
  elp::DVec<unsigned long long> dev_tables;
  size_t tables_n = 0;
  int tables_max_cycle = 0;
  unsigned long long tables_quals[2] = {0, 0};  // the qualities that had table slots in the gather that made dev_tables (no other row holds anything)
  elp::DVec<unsigned long long> tables_pack;    // elp_bqsr_tables_fetch_rows: the rows of the asked qualities, packed on the device
  elp::DVec<uint8_t> lut_rows_dev;              // elp_bqsr_lut_upload_rows: the LUT in rows form before its expansion into lut_dev
  // device group (group.hip)
  void *comm = nullptr;  // ncclComm_t
  bool comm_borrowed = false;  // elp_group_share: another context of this process owns (and destroys) it
  int (*xport)(void *, int64_t *, size_t) = nullptr;  // caller's transport instead of RCCL (elp_group_init_transport)
  void *xport_user = nullptr;
  int (*p2p)(void *, int, const void *, size_t, int, void *, size_t) = nullptr;  // caller's send-receive (elp_group_set_p2p)
  void *p2p_user = nullptr;
  int group_rank = 0, group_world = 1;
  elp_ctx *group_owner = nullptr;            // elp_group_share: the context whose group this one uses ...
  std::vector<elp_ctx *> group_borrowers;    // ... and, at the owner, the contexts that use its group (group_release takes it from them)

  // records staged from BAM bytes (bam.hip): the inflated records stay in HBM, elp_emit_sorted_bam reads bases and tags from them
  elp::DVec<uint8_t> raw;
  elp::DVec<uint64_t> raw_off;  // per staged record: offset of its block_size field in raw (n + 1)
  uint64_t raw_n = 0, raw_bytes = 0;
  uint64_t max_raw_rec = 0;     // largest staged record in bytes (block_size field included): bounds the chunks of elp_emit_sorted_bam
  elp::DVec<uint8_t> rg_ids;    // header read-group ids, concatenated (RG:Z -> rgid)
  elp::DVec<uint32_t> rg_ids_off;
  bool have_rg_ids = false;
  hipStream_t copy_stream = nullptr;
  elp::DVec<uint8_t> lut_dev;      // ApplyBQSR's dense LUT + covariate-present bytes uploaded ahead of elp_bqsr_apply (elp_bqsr_lut_upload)
  void *lut_pinned = nullptr;
  size_t lut_pinned_cap = 0;
  hipEvent_t lut_ev = nullptr;
  hipEvent_t apply_ev = nullptr;   // behind the last elp_bqsr_apply that read lut_dev / lut_wk: the next upload's copy waits for it
  int lut_uploaded_cycle = 0;
  // the row dictionary of the uploaded LUT (bqsr.hip: lut_dictionary), built on the copy stream behind the upload when the facts it needs
  // are known at that time; elp_bqsr_apply uses it if they still hold
  elp::DVec<uint32_t> lut_wk;
  bool dict_ready = false;
  bool dict_per_cov = false;  // the prebuilt dictionary is one per covariate (apply3's covariate split)
  int dict_qlo = 0, dict_nqi = 0, dict_lmax = 0, dict_cycle = 0, dict_ncov = 0;
  hipEvent_t tables_ev = nullptr;  // recorded on `stream` behind the last writer of dev_tables (gather, tables_add, all-reduce): elp_bqsr_tables_fetch
                                   // copies on copy_stream behind it, so the context's stream is free for the next stage meanwhile
  void *bounce[2] = {nullptr, nullptr};  // pinned double buffer for pageable sources
  hipEvent_t bounce_ev[2] = {nullptr, nullptr};

  // a page of page-locked host memory + an event: a few words come back while the stream runs on (mailbox(), ctx.hip)
  uint32_t *mail = nullptr;
  hipEvent_t mail_ev = nullptr;

  // the side lane (round 6): a shadow context with a stream, scratch pool, error words and radix workspaces of its OWN whose columns are
  // views of this context's.  The duplication-metrics pass runs there (metrics.hip): it reads what mark duplicates left and writes nothing
  // another stage reads, so a host may call elp_dup_metrics from a second thread while this context sorts - the pass's two dozen small
  // launches and five read-backs then hide under the sort's kernels instead of standing behind them.
  // Two of them: lane 0 = the metrics pass, lane 1 = the coordinate sort (sort.hip: it reads the key column and the comparator's columns
  // and writes the permutation - nothing the BQSR stages touch; a host may call elp_sort_coordinate from a thread of its own right behind
  // elp_mark_duplicates while the context gathers, finalises and applies: the reference runs these one after the other, cmd/filter.go:162-196).
  elp_ctx *side[2] = {nullptr, nullptr};
  hipEvent_t side_ev[2] = {nullptr, nullptr};   // in: the lane's stream waits for the context's
  hipEvent_t side_done[2] = {nullptr, nullptr}; // out: the context's stream waits for what the lane left queued
  // elp_sort_ahead (round 6): the coordinate sort's KEY passes need the key column only, not the duplicate bits - with the option on,
  // elp_mark_duplicates queues them on the sort lane the moment its front pass has written the keys (no host wait involved), and they run
  // under the pair phase; elp_sort_coordinate then finds the sorted words and only breaks the ties (which see the final FLAGs)
  bool sort_ahead = false;
  uint64_t adapt_epoch = 0;          // counts the writes of the key column (adapt_begin)
  uint64_t presort_epoch = ~0ull, presort_n = 0;
  uint64_t *presort_ks = nullptr;    // the sorted words, in the sort lane's scratch slot 0 (valid while presort_epoch == adapt_epoch)
  int presort_idx_bits = 0;

  // snapshot of the mutable columns
  elp::DVec<uint16_t> snap_flag;
  elp::DVec<uint8_t> snap_qual;
  uint64_t snap_n = 0, snap_qual_bytes = 0;
  bool have_snapshot = false;

  // elp_set_tuning: kernel choices a caller (tests, A/B measurements) can pin; 0 = the library decides
  struct Tuning {
    int count_kernel = 0;      // 1: the general count kernel even where the one-length kernel applies; 2: never split by covariate; 3: always
    int apply_kernel = 0;      // 1: the general apply kernel; 3: the one-length kernel split by covariate even where one table holds them all
    int count3_rlog = -1;      // >= 0: log2 of the context-cell replication of the one-length count kernel
    int qual_hint = 0;         // 1: no sampled quality hint (tables sized for every quality); 2: hint without the value qual_hint_drop
    int qual_hint_drop = -1;
    int pair_table_slots = 1 << 20;  // cap on the LDS table slots of a pair bucket (mark duplicates); tests shrink it to reach the overflow path
    long long bgzf_piece = 1ll << 30;    // elp_stage_bgzf: inflated bytes per device pass (tests: small pieces, records pending across them)
    int bgzf_weak_guess = 0;   // 1: a block guesses its first record start without looking at the bytes (tests: every guess wrong, all repaired)
    int score_kernel = 0;      // 1: the general (flat) score kernel even for read sets of one length
    int mate_path = 0;         // 1: every mate candidate goes through the table path (no neighbour shortcut)
    int radix_tile = 0;        // 1: radix passes in tiles of 4096 keys whatever the length; 2: of 8192 (default: 8192 from 8 M keys on)
    int sort_pairs = 0;        // 1: the coordinate sort moves (key, index) pairs even where key << b | index fits one word
    int tie_rounds = 0;        // 1: the sort's long runs by LSD rounds over every live position (no key-then-compare shortcut)
    int exchange_piece = 0;    // > 0: records per piece of elp_exchange_records (tests: several pieces on small inputs)
    long long bgzf_inflate_piece = 2ll << 30;  // elp_stage_bgzf: inflated bytes whose blocks are decoded by one launch (token scratch: 2.7x that)
    int bgzf_copy_chunk = 0;   // elp_stage_bgzf: blocks per H2D chunk / decoder launch (0: what fills the chip once)
    int bgzf_tok_lds = 0;      // (experiments) unused dynamic LDS bytes per decoder wave: lowers the waves per CU
    int bgzf_first_chunk_div = 4;  // the first H2D chunk is 1/div of the others
    int bgzf_fixed = 0;        // 1: elp_emit_sorted_bgzf writes fixed Huffman codes only (round 5's form) instead of dynamic codes
    int bgzf_tok_fail_above = 0;  // (tests) the token scratch "does not fit" for more than this many blocks: the halving path
    int bgzf_inflate = 0;      // 1: elp_stage_bgzf inflates with round 5's one-kernel decoder (window in LDS) instead of tokens + resolve
    int bgzf_stored = 0;       // 1: elp_emit_sorted_bgzf writes stored DEFLATE blocks (round 4's form) instead of compressing
    int apply_wgs = 0;         // 1 .. 3: workgroups per CU of the one-length ApplyBQSR kernel (default: as many as its LDS allows, at most 3) - A/B runs
                               // of a step whose sort runs at the same time and needs LDS of its own
    int presort_tile = 0;      // elp_sort_ahead: radix tile of the key passes made ahead (1 / 2 / 3 = 4096 / 8192 / 16384 keys; default 2: a tile of
                               // 16384 keys is a 1024-thread workgroup around 128 KB of LDS, which finds no CU while kernels of small workgroups
                               // keep every CU partly occupied)
    int side_priority = 0;     // 1: the side lanes' streams are made with the highest priority the device offers (default: the default priority;
                               // measured with the bench's step: no difference)
    int md_fused = 0;          // 1: mark duplicates by the separate passes of rounds 2-5 (adapt_fixed, md_keys, md_mate_scan, md_mate_pairs) instead of
                               // the fused front pass of round 6 (k_md_front); tests run both
  } tune;

  // generic scratch pool (grown on demand, reused between calls)
  elp::DVec<uint8_t> scratch[8];

  // profiling
  bool profiling = false;
  const char *prof_prefix = nullptr;  // prepended to the names of the launches made while it is set (ProfScope)
  std::vector<std::string> prof_names;
  std::map<std::string, int> prof_index;
  std::vector<uint64_t> prof_launches;
  std::vector<double> prof_ms;
  std::vector<elp::ProfPending> prof_pending;
};

namespace elp {

int set_error(elp_ctx *c, int code, const char *fmt, ...);
int prof_begin(elp_ctx *c, const char *name);  // returns pending index or -1
void prof_end(elp_ctx *c, int pending);
int prof_flush(elp_ctx *c);
struct ProfScope {  // launches inside the scope are booked as <prefix><name>
  elp_ctx *c;
  const char *saved;
  ProfScope(elp_ctx *ctx, const char *prefix) : c(ctx), saved(ctx->prof_prefix) { ctx->prof_prefix = prefix; }
  ~ProfScope() { c->prof_prefix = saved; }
};

#define ELP_HIP(ctx, call)                                                                             \
  do {                                                                                                 \
    hipError_t e__ = (call);                                                                           \
    if (e__ != hipSuccess)                                                                             \
      return elp::set_error((ctx), e__ == hipErrorOutOfMemory ? ELP_ERR_NOMEM : ELP_ERR_HIP, "%s failed: %s (%s:%d)", #call, \
                            hipGetErrorString(e__), __FILE__, __LINE__);                               \
  } while (0)

#define ELP_TRY(expr)            \
  do {                           \
    int rc__ = (expr);           \
    if (rc__ != 0) return rc__;  \
  } while (0)

// Waiting for a stream.  hipStreamSynchronize blocks on an interrupt after a short spin; on a box whose host is busy the wake-up takes
// 100-200 us, and the path waits ~15 times per step for a few words that decide the next launches' sizes (measured: identical kernel
// times, 0.9 against 2.3 ms per step without a kernel running).  ELP_SYNC_SPIN=1 (default) polls the stream instead: the waiting
// thread keeps its core.  0: hipStreamSynchronize.
bool sync_spin();  // ctx.hip
// (the pause instruction is x86's; after ~200 us of polling the waiting thread offers its core to whoever is runnable between
// two queries - several ranks per node, the sfm side threads - instead of holding it for a whole kernel)
inline void spin_relax(unsigned &spins) {
  if (++spins < 4096u) {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#endif
  } else {
    sched_yield();
  }
}
inline hipError_t stream_wait(hipStream_t st) {
  if (!sync_spin()) return hipStreamSynchronize(st);
  hipError_t e;
  unsigned spins = 0;
  while ((e = hipStreamQuery(st)) == hipErrorNotReady) spin_relax(spins);
  return e;
}
inline hipError_t event_wait(hipEvent_t ev) {
  if (!sync_spin()) return hipEventSynchronize(ev);
  hipError_t e;
  unsigned spins = 0;
  while ((e = hipEventQuery(ev)) == hipErrorNotReady) spin_relax(spins);
  return e;
}

int debug_poison();  // ctx.hip: the byte of ELP_DEBUG_POISON, or -1
bool debug_trace();  // ctx.hip: ELP_DEBUG_TRACE=1 - every launch is named on stderr and waited for (which kernel faulted)

// grow-only allocation; keep = copy old contents (device to device)
template <class T>
int ensure(elp_ctx *c, DVec<T> &v, size_t n, bool keep = false, size_t keep_elems = 0) {
  if (n <= v.cap) return 0;
  size_t ncap = keep ? (n + n / 2 + 16) : n;
  T *np = nullptr;
  ELP_HIP(c, hipMalloc((void **)&np, ncap * sizeof(T) + debug_guard_bytes()));
  if (debug_guard_bytes()) debug_guard_arm(np, ncap * sizeof(T));
  // ELP_DEBUG_POISON=<byte> (tests): every new device buffer starts filled with that byte - a kernel that reads memory nothing wrote
  // shows up as a parity failure instead of depending on what the allocator hands out
  if (const int pz = debug_poison(); pz >= 0) {  // (on the context's stream and waited for: a fill on the null stream could land behind later kernels)
    ELP_HIP(c, hipMemsetAsync(np, pz, ncap * sizeof(T), c->stream));
    ELP_HIP(c, elp::stream_wait(c->stream));
  }
  if (keep && v.p && keep_elems) {
    hipError_t e = hipMemcpyAsync(np, v.p, keep_elems * sizeof(T), hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = elp::stream_wait(c->stream);
    if (e != hipSuccess) {
      (void)hipFree(np);
      return set_error(c, ELP_ERR_HIP, "device copy failed while growing a column: %s", hipGetErrorString(e));
    }
  }
  if (v.p) {
    (void)elp::stream_wait(c->stream);
    debug_guard_release(v.p);
    // (ELP_DEBUG_POISON: the buffer that is given up is overwritten first - a pointer into it that somebody still holds reads 0xDD
    // instead of the old contents a freed block usually keeps)
    if (debug_poison() >= 0) { (void)hipMemsetAsync(v.p, 0xDD, v.cap * sizeof(T), c->stream); (void)elp::stream_wait(c->stream); }
    (void)hipFree(v.p);
  }
  v.p = np;
  v.cap = ncap;
  return 0;
}

// typed view into the scratch pool
template <class T>
int scratch(elp_ctx *c, int slot, size_t n, T **out) {
  ELP_TRY(ensure(c, c->scratch[slot], n * sizeof(T) + 256));
  *out = reinterpret_cast<T *>(c->scratch[slot].p);
  return 0;
}

// launch with optional event bracketing
#define ELP_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                              \
  do {                                                                                      \
    int pp__ = (ctx)->profiling ? elp::prof_begin((ctx), (name)) : -1;                      \
    if (elp::debug_trace()) fprintf(stderr, "[elp] %s\n", (name));                          \
    hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__);             \
    if (pp__ >= 0) elp::prof_end((ctx), pp__);                                              \
    ELP_HIP((ctx), hipGetLastError());                                                      \
    if (elp::debug_trace()) ELP_HIP((ctx), elp::stream_wait((ctx)->stream));            \
  } while (0)

inline unsigned blocks_for(uint64_t n, unsigned per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// ---- shared device-side helpers ----
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
  return x;
}

// QNAME access goes through unaligned 8-byte loads (the column is padded by 64 bytes, so reading up to 31 bytes past a name is safe)
__device__ __forceinline__ uint64_t load8(const uint8_t *__restrict__ p) {
  uint64_t w;
  __builtin_memcpy(&w, p, 8);
  return w;
}
__device__ __forceinline__ uint64_t low_bytes(uint64_t w, uint32_t nbytes) {  // keep the first nbytes (1..8) bytes in memory order
  return nbytes >= 8 ? w : (w & ((1ull << (8 * nbytes)) - 1ull));
}
// bytewise QNAME comparison with Go string semantics (shorter prefix is smaller), eight bytes per step
__device__ inline int qname_cmp(const uint8_t *__restrict__ q, const uint64_t *__restrict__ off, uint32_t a, uint32_t b) {
  const uint64_t oa = off[a], ob = off[b];
  const uint32_t la = (uint32_t)(off[a + 1] - oa), lb = (uint32_t)(off[b + 1] - ob);
  const uint32_t m = la < lb ? la : lb;
  for (uint32_t i = 0; i < m; i += 8) {
    const uint32_t nb = m - i;
    const uint64_t wa = low_bytes(load8(q + oa + i), nb), wb = low_bytes(load8(q + ob + i), nb);
    if (wa != wb) {
      const uint64_t ba = __builtin_bswap64(wa), bb = __builtin_bswap64(wb);  // first byte in memory = most significant
      return ba < bb ? -1 : 1;
    }
  }
  return la < lb ? -1 : (la > lb ? 1 : 0);
}
// 32 bytes per round trip: all eight loads of a round are issued before the first compare (names of up to 32 bytes - the usual
// case - cost one memory latency, not one per 8 bytes).  The column is padded by 64 bytes, so reading past a name's end is safe.
__device__ inline bool qname_eq(const uint8_t *__restrict__ q, const uint64_t *__restrict__ off, uint32_t a, uint32_t b) {
  const uint64_t oa = off[a], ob = off[b];
  const uint32_t la = (uint32_t)(off[a + 1] - oa), lb = (uint32_t)(off[b + 1] - ob);
  if (la != lb) return false;
  uint64_t diff = 0;
  for (uint32_t i = 0; i < la && diff == 0; i += 32) {
    const uint64_t a0 = load8(q + oa + i), a1 = load8(q + oa + i + 8), a2 = load8(q + oa + i + 16), a3 = load8(q + oa + i + 24);
    const uint64_t b0 = load8(q + ob + i), b1 = load8(q + ob + i + 8), b2 = load8(q + ob + i + 16), b3 = load8(q + ob + i + 24);
    const uint32_t nb = la - i;  // bytes left (>= 1)
    diff |= low_bytes(a0 ^ b0, nb);
    diff |= nb > 8 ? low_bytes(a1 ^ b1, nb - 8) : 0ull;
    diff |= nb > 16 ? low_bytes(a2 ^ b2, nb - 16) : 0ull;
    diff |= nb > 24 ? low_bytes(a3 ^ b3, nb - 24) : 0ull;
  }
  return diff == 0;
}

// sam/sam-types.go:408-421
__host__ __device__ inline uint16_t mod_flag(uint16_t flag) {
  if ((flag & F_MULTIPLE) == 0) flag &= (uint16_t) ~(F_NEXT_UNMAPPED | F_NEXT_REVERSED);
  if (flag & F_UNMAPPED) flag &= (uint16_t)~F_REVERSED;
  if (flag & F_NEXT_UNMAPPED) flag &= (uint16_t)~F_NEXT_REVERSED;
  return flag;
}

// ---- cross-TU entry points ----
int radix_sort_pairs(elp_ctx *c, uint64_t *keys, uint32_t *vals, uint64_t *keys_tmp, uint32_t *vals_tmp, uint64_t n,
                     uint64_t **keys_out, uint32_t **vals_out);
// the same over the low `ndigits` bytes of the keys only, every pass run (no histogram read-back, no host synchronisation)
int radix_sort_fused(elp_ctx *c, const uint64_t *keycol, uint64_t n, int key_bits, int idx_bits, uint64_t *buf0, uint64_t *buf1, uint64_t **out);
// bounds of the buckets the sorted low key bits define, reported by the LAST pass of radix_sort_pairs_low: bucket of a key = (key & mask) >> shift;
// start[b] = first element of bucket b, nend[b] = ~(one behind its last) - both arrays filled with 0xFF by the caller (an empty bucket keeps
// start 0xFFFFFFFF, end ~nend = 0)
struct RadixBounds {
  uint32_t *start = nullptr, *nend = nullptr;
  uint32_t mask = 0;
  int shift = 0;
};
int radix_sort_pairs_low(elp_ctx *c, uint64_t *keys, uint32_t *vals, uint64_t *keys_tmp, uint32_t *vals_tmp, uint64_t n, int ndigits,
                         uint64_t **keys_out, uint32_t **vals_out, const uint64_t *first_src = nullptr, bool identity_vals = false,
                         const uint32_t *n_dev = nullptr /* the length is *n_dev on the device and `n` its upper bound */, RadixBounds bnd = RadixBounds{});
int exclusive_scan_u32(elp_ctx *c, const uint32_t *in, uint32_t *out, uint64_t n, uint32_t *total_host /* may be null */);
int ensure_adapted(elp_ctx *c, bool check_quals = true);
// the two halves of ensure_adapted for a caller that computes the fixed-field part (unclipped positions, sort keys) itself - mark
// duplicates' front pass, markdup.hip: adapt_begin = buffers, key width (*pos_bits), the error word's fill; adapt_scores = the score kernel
// (every record's score and low-quality-tail bounds).  The caller sets c->adapted once its own pass is queued.
int adapt_begin(elp_ctx *c, int *pos_bits);
int adapt_scores(elp_ctx *c);
int mailbox(elp_ctx *c);  // ctx.hip: c->mail (1024 words, page-locked) and c->mail_ev exist
int side_lane(elp_ctx *c, int lane, elp_ctx **out);  // ctx.hip: c->side[lane] exists; its stream waits for what is queued on c->stream now
int side_join(elp_ctx *c, int lane);                 // ctx.hip: c->stream waits for what is queued on the lane's stream now
int sort_presort(elp_ctx *c);                        // sort.hip: the sort's key passes queued on lane 1 behind what c->stream holds now (elp_sort_ahead)
void prof_merge_side(elp_ctx *c);          // ctx.hip: the side lane's launch times join the context's
constexpr int ADAPT_WORDS = 6;
void adapt_note(elp_ctx *c, const uint32_t *words /* ADAPT_WORDS of adapt_err */);  // sort.hip: the score kernel's words were read (by whoever synchronised anyway)
int adapt_quality_error(elp_ctx *c);
int ensure_uniform_len(elp_ctx *c);  // sort.hip: c->uniform_len
int ensure_qual_present(elp_ctx *c, bool exact = false);  // exact: scan the whole column instead of a sample
int fetch_err(elp_ctx *c, uint32_t *words /* 4 */);
int radix_check(elp_ctx *c);  // reads the error words if radix passes were queued since they were last read
void group_release(elp_ctx *c);
int group_sendrecv(elp_ctx *c, int send_peer, const void *send_dev, size_t send_bytes, int recv_peer, void *recv_dev, size_t recv_bytes);  // group.hip
int tables_written(elp_ctx *c);  // bqsr.hip: dev_tables were just written on c->stream
int stage_reserve(elp_ctx *c, uint64_t n, uint64_t qb, uint64_t co, uint64_t sb, uint64_t lb);
int merge_spread_slots(elp_ctx *groups, elp_ctx *spread, uint64_t **slots_out);  // filter.hip: the merge order as ranks, on the device  // grows the staged columns (ctx.hip)
int stage_recode_seq(elp_ctx *c, uint64_t from, uint64_t bytes);
int stage_bam_columns(elp_ctx *c, uint32_t n_rec, uint64_t piece_bytes, uint64_t raw_end, uint64_t max_raw_rec, uint16_t split_id);  // bam.hip
uint64_t bgzf_framed_size(uint64_t n_bytes);                                   // bgzf.hip
int bgzf_frame(elp_ctx *c, const uint8_t *raw, uint64_t n_bytes, uint8_t *out);  // device to device, stored blocks
int bgzf_deflate(elp_ctx *c, const uint8_t *raw, uint64_t n_bytes, uint8_t *out, uint64_t *out_bytes);  // device to device, compressed members; *out_bytes <= bgzf_framed_size                               // BAM nibbles -> code nibbles on seq4[from, from + bytes)

}  // namespace elp
