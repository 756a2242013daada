// ctx.hip — context life cycle, staging of record batches into the HBM column store, result fetch, profiling.
// Replaces the collection side of record batching: Sam.AddNodes / Slice(&alns) (sam/filter-pipeline.go:108-124).
#include "common.hpp"

namespace elp {

// ---- ELP_DEBUG_GUARD: a pattern behind every buffer, checked at release
namespace {
constexpr size_t GUARD = 4096;
constexpr uint8_t GUARD_BYTE = 0xC3;
std::mutex guard_mu;
std::map<void *, size_t> guard_of;  // buffer -> payload bytes
int guard_check_one(void *p, size_t payload) {
  uint8_t host[GUARD];
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host, static_cast<uint8_t *>(p) + payload, GUARD, hipMemcpyDeviceToHost) != hipSuccess) return 0;
  for (size_t k = 0; k < GUARD; k++)
    if (host[k] != GUARD_BYTE) {
      fprintf(stderr, "[elp] ELP_DEBUG_GUARD: byte %zu behind the end of a %zu-byte device buffer was overwritten (0x%02x)\n", k, payload, host[k]);
      return 1;
    }
  return 0;
}
}  // namespace
size_t debug_guard_bytes() {
  static const size_t v = [] { const char *e = getenv("ELP_DEBUG_GUARD"); return (e && *e && *e != '0') ? GUARD : (size_t)0; }();
  return v;
}
void debug_guard_arm(void *p, size_t payload) {
  (void)hipMemset(static_cast<uint8_t *>(p) + payload, GUARD_BYTE, GUARD);
  (void)hipDeviceSynchronize();
  std::lock_guard<std::mutex> lk(guard_mu);
  guard_of[p] = payload;
}
void debug_guard_release(void *p) {
  if (!debug_guard_bytes()) return;
  size_t payload = 0;
  {
    std::lock_guard<std::mutex> lk(guard_mu);
    auto it = guard_of.find(p);
    if (it == guard_of.end()) return;
    payload = it->second;
    guard_of.erase(it);
  }
  if (guard_check_one(p, payload)) abort();
}
int debug_guard_check_all() {
  if (!debug_guard_bytes()) return 0;
  std::lock_guard<std::mutex> lk(guard_mu);
  int bad = 0;
  for (auto &kv : guard_of) bad += guard_check_one(kv.first, kv.second);
  return bad;
}

bool sync_spin() {
  static const bool v = [] { const char *e = getenv("ELP_SYNC_SPIN"); return !(e && *e == '0'); }();
  return v;
}
bool debug_trace() {
  static const bool v = [] { const char *e = getenv("ELP_DEBUG_TRACE"); return e && *e && *e != '0'; }();
  return v;
}
int debug_poison() {
  static const int v = [] {
    const char *e = getenv("ELP_DEBUG_POISON");
    return (e && *e) ? (int)(strtol(e, nullptr, 0) & 0xFF) : -1;
  }();
  return v;
}


int set_error(elp_ctx *c, int code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf;
  return code;
}

int prof_begin(elp_ctx *c, const char *name0) {
  // the shared radix / scan launches are booked under the stage that called them ("md_radix_scatter", "mx_radix_scatter")
  const std::string name = c->prof_prefix ? std::string(c->prof_prefix) + name0 : std::string(name0);
  auto it = c->prof_index.find(name);
  int id;
  if (it == c->prof_index.end()) {
    id = (int)c->prof_names.size();
    c->prof_names.push_back(name);
    c->prof_index[name] = id;
    c->prof_launches.push_back(0);
    c->prof_ms.push_back(0.0);
  } else {
    id = it->second;
  }
  ProfPending p;
  p.name_id = id;
  if (hipEventCreate(&p.a) != hipSuccess) return -1;
  if (hipEventCreate(&p.b) != hipSuccess) { (void)hipEventDestroy(p.a); return -1; }
  (void)hipEventRecord(p.a, c->stream);
  c->prof_pending.push_back(p);
  return (int)c->prof_pending.size() - 1;
}
void prof_end(elp_ctx *c, int pending) { (void)hipEventRecord(c->prof_pending[pending].b, c->stream); }
int prof_flush(elp_ctx *c) {
  if (c->prof_pending.empty()) return 0;
  ELP_HIP(c, elp::stream_wait(c->stream));
  for (auto &p : c->prof_pending) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      c->prof_ms[p.name_id] += ms;
      c->prof_launches[p.name_id] += 1;
    }
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  c->prof_pending.clear();
  return 0;
}

__global__ void k_rebase_offsets(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, uint64_t n_plus_1, uint64_t base,
                                 uint64_t src0) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_plus_1) dst[i] = src[i] - src0 + base;
}

// SEQ is restaged from BAM's 4-bit IUPAC codes (first base of a byte in the HIGH nibble) into "code nibbles", first base in the
// LOW nibble: A 0, C 1, G 2, T 3, anything else 8.  The per-base kernels then get the 2-bit base code and the "is A/C/G/T" test
// of 16 bases with two 64-bit ALU operations instead of a nibble swap plus a 17-operation SWAR classification per block, and the
// reference contigs (k_pack_reference) use the same code, so read-vs-reference comparison stays one XOR.
__device__ __forceinline__ uint32_t recode_byte(uint32_t b) {
  constexpr uint64_t TAB = 0x8888888388828108ull;  // nibble n of TAB = code of BAM base n: 1 -> 0, 2 -> 1, 4 -> 2, 8 -> 3, else 8
  return (uint32_t)((TAB >> (4 * (b >> 4))) & 15u) | ((uint32_t)((TAB >> (4 * (b & 15u))) & 15u) << 4);
}
__global__ __launch_bounds__(256) void k_recode_seq(uint8_t *__restrict__ seq, uint64_t nbytes) {
  const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i >= nbytes) return;
  if (i + 16 <= nbytes) {
    uint32_t w[4];
    __builtin_memcpy(w, seq + i, 16);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t x = w[k];
      w[k] = recode_byte(x & 0xFFu) | (recode_byte((x >> 8) & 0xFFu) << 8) | (recode_byte((x >> 16) & 0xFFu) << 16) | (recode_byte(x >> 24) << 24);
    }
    __builtin_memcpy(seq + i, w, 16);
  } else {
    for (uint64_t k = i; k < nbytes; k++) seq[k] = (uint8_t)recode_byte(seq[k]);
  }
}

int stage_recode_seq(elp_ctx *c, uint64_t from, uint64_t bytes) {
  hipLaunchKernelGGL(k_recode_seq, dim3(blocks_for((bytes + 15) / 16, 256)), dim3(256), 0, c->stream, c->seq4.p + from, bytes);
  ELP_HIP(c, hipGetLastError());
  return 0;
}

// the device-side error words, behind everything queued on the context's stream.  Bit 256 of word 0 (a tile look-back of a radix pass
// timed out: radix.hip; the result of that sort is wrong) is everybody's to report: the sorts do not synchronise for it themselves, the
// next read of the words - the following stage's, elp_sync's, a getter's - does.
int fetch_err(elp_ctx *c, uint32_t *words) {
  ELP_HIP(c, hipMemcpyAsync(words, c->err_flag.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  ELP_HIP(c, elp::stream_wait(c->stream));
  c->radix_check_pending = false;
  if (words[0] & 256u) {
    const uint32_t w0 = words[0] & ~256u;
    ELP_HIP(c, hipMemcpyAsync(c->err_flag.p, &w0, 4, hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, elp::stream_wait(c->stream));
    c->sorted = false;
    c->marked = false;
    return set_error(c, ELP_ERR_HIP, "radix sort: tile look-back timed out");
  }
  return 0;
}
int mailbox(elp_ctx *c) {
  if (!c->mail) ELP_HIP(c, hipHostMalloc((void **)&c->mail, 1024 * sizeof(uint32_t), hipHostMallocDefault));
  if (!c->mail_ev) ELP_HIP(c, hipEventCreateWithFlags(&c->mail_ev, hipEventDisableTiming));
  return 0;
}
int side_lane(elp_ctx *c, int lane, elp_ctx **out) {
  if (!c->side[lane]) {
    elp_ctx *s = new elp_ctx();
    s->device = c->device;
    s->n_cu = c->n_cu;
    // (elp_set_tuning "side_priority" = 1, before the lane's first use: the highest stream priority - measured: no difference)
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const hipError_t se = c->tune.side_priority != 1 ? hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking)
                                                     : hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, prio_hi);
    if (se != hipSuccess || ensure(s, s->err_flag, 4) != 0 ||
        hipMemsetAsync(s->err_flag.p, 0, 16, s->stream) != hipSuccess) {
      if (s->stream) (void)hipStreamDestroy(s->stream);
      delete s;
      return set_error(c, ELP_ERR_HIP, "the side lane's stream could not be made");
    }
    c->side[lane] = s;
  }
  if (!c->side_ev[lane]) ELP_HIP(c, hipEventCreateWithFlags(&c->side_ev[lane], hipEventDisableTiming));
  if (!c->side_done[lane]) ELP_HIP(c, hipEventCreateWithFlags(&c->side_done[lane], hipEventDisableTiming));
  // what is queued on the context's stream up to here (mark duplicates) comes first
  ELP_HIP(c, hipEventRecord(c->side_ev[lane], c->stream));
  ELP_HIP(c, hipStreamWaitEvent(c->side[lane]->stream, c->side_ev[lane], 0));
  c->side[lane]->profiling = c->profiling;
  c->side[lane]->tune = c->tune;
  *out = c->side[lane];
  return 0;
}
int side_join(elp_ctx *c, int lane) {
  ELP_HIP(c, hipEventRecord(c->side_done[lane], c->side[lane]->stream));
  ELP_HIP(c, hipStreamWaitEvent(c->stream, c->side_done[lane], 0));
  return 0;
}
void prof_merge_side(elp_ctx *c) {
  for (int lane = 0; lane < 2; lane++) {
    elp_ctx *s = c->side[lane];
    if (!s) continue;
    (void)prof_flush(s);
    for (size_t k = 0; k < s->prof_names.size(); k++) {
      if (!s->prof_launches[k]) continue;
      auto it = c->prof_index.find(s->prof_names[k]);
      int id;
      if (it == c->prof_index.end()) {
        id = (int)c->prof_names.size();
        c->prof_names.push_back(s->prof_names[k]);
        c->prof_index[s->prof_names[k]] = id;
        c->prof_launches.push_back(0);
        c->prof_ms.push_back(0.0);
      } else {
        id = it->second;
      }
      c->prof_launches[id] += s->prof_launches[k];
      c->prof_ms[id] += s->prof_ms[k];
      s->prof_launches[k] = 0;
      s->prof_ms[k] = 0.0;
    }
  }
}
int radix_check(elp_ctx *c) {
  if (!c->radix_check_pending) return 0;
  uint32_t e[4];
  return fetch_err(c, e);
}

}  // namespace elp

using namespace elp;

extern "C" {

int elp_debug_check_guards(void) { return elp::debug_guard_check_all(); }


int elp_create(int device_ordinal, elp_ctx **out) {
  if (!out) return ELP_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ELP_ERR_HIP;  // no CPU fallback by design
  if (device_ordinal < 0 || device_ordinal >= ndev) return ELP_ERR_ARG;
  if (hipSetDevice(device_ordinal) != hipSuccess) return ELP_ERR_HIP;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return ELP_ERR_HIP;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ELP_ERR_UNSUPPORTED;  // kernels are built for gfx950 only
  elp_ctx *c = new elp_ctx();
  c->device = device_ordinal;
  c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return ELP_ERR_HIP; }
  if (ensure(c, c->err_flag, 4) != 0 || hipMemsetAsync(c->err_flag.p, 0, 16, c->stream) != hipSuccess) { delete c; return ELP_ERR_HIP; }
  *out = c;
  return ELP_OK;
}

void elp_destroy(elp_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)elp::stream_wait(c->stream);
  for (auto &p : c->prof_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  if (c->lut_pinned) (void)hipHostFree(c->lut_pinned);
  if (c->lut_ev) (void)hipEventDestroy(c->lut_ev);
  if (c->apply_ev) (void)hipEventDestroy(c->apply_ev);
  for (auto p : c->h_ref_seq) if (p) (void)hipFree(p);
  for (auto p : c->h_sites) if (p) (void)hipFree(p);
  for (auto p : c->h_site_idx) if (p) (void)hipFree(p);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  for (int lane = 0; lane < 2; lane++) {
    if (c->side[lane]) {  // (its columns are views that were taken back when the pass that used them returned: it owns its scratch only)
      (void)elp::stream_wait(c->side[lane]->stream);
      for (auto &p : c->side[lane]->prof_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
      (void)hipStreamDestroy(c->side[lane]->stream);
      delete c->side[lane];
      c->side[lane] = nullptr;
    }
    if (c->side_ev[lane]) (void)hipEventDestroy(c->side_ev[lane]);
    if (c->side_done[lane]) (void)hipEventDestroy(c->side_done[lane]);
  }
  if (c->mail) (void)hipHostFree(c->mail);
  if (c->mail_ev) (void)hipEventDestroy(c->mail_ev);
  for (int k = 0; k < 2; k++) {
    if (c->bounce[k]) (void)hipHostFree(c->bounce[k]);
    if (c->bounce_ev[k]) (void)hipEventDestroy(c->bounce_ev[k]);
  }
  if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
  if (c->tables_ev) (void)hipEventDestroy(c->tables_ev);
  group_release(c);
  (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *elp_last_error(const elp_ctx *c) { return c ? c->err.c_str() : "null context"; }

int elp_sync(elp_ctx *c) {
  if (!c) return ELP_ERR_ARG;
  if (c->radix_check_pending) return radix_check(c);  // (synchronises)
  ELP_HIP(c, elp::stream_wait(c->stream));
  return 0;
}

void *elp_stream(elp_ctx *c) { return c ? (void *)c->stream : nullptr; }

int elp_set_header(elp_ctx *c, const elp_header *h) {
  if (!c || !h || h->n_ref < 0 || h->n_rg < 0 || (h->n_ref && !h->ref_len) || (h->n_rg && (!h->rg_lib || !h->rg_cov)))
    return set_error(c, ELP_ERR_ARG, "elp_set_header: bad arguments");
  ELP_HIP(c, hipSetDevice(c->device));
  c->n_ref = h->n_ref; c->n_rg = h->n_rg; c->n_lib = h->n_lib; c->n_cov = h->n_cov;
  c->apply_recs_valid = false;  // (they hold the read groups' covariates)
  c->h_ref_len.assign(h->ref_len, h->ref_len + h->n_ref);
  c->h_rg_lib.assign(h->rg_lib, h->rg_lib + h->n_rg);
  c->h_rg_cov.assign(h->rg_cov, h->rg_cov + h->n_rg);
  for (int i = 0; i < h->n_rg; i++) {
    if (c->h_rg_lib[i] != ELP_NIL16 && c->h_rg_lib[i] >= h->n_lib) return set_error(c, ELP_ERR_ARG, "rg_lib[%d] out of range", i);
    if (c->h_rg_cov[i] >= h->n_cov) return set_error(c, ELP_ERR_ARG, "rg_cov[%d] out of range", i);
  }
  ELP_TRY(ensure(c, c->ref_len, (size_t)h->n_ref + 1));
  ELP_TRY(ensure(c, c->rg_lib, (size_t)h->n_rg + 1));
  ELP_TRY(ensure(c, c->rg_cov, (size_t)h->n_rg + 1));
  if (h->n_ref) ELP_HIP(c, hipMemcpyAsync(c->ref_len.p, h->ref_len, h->n_ref * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  if (h->n_rg) {
    ELP_HIP(c, hipMemcpyAsync(c->rg_lib.p, h->rg_lib, h->n_rg * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, hipMemcpyAsync(c->rg_cov.p, h->rg_cov, h->n_rg * sizeof(uint16_t), hipMemcpyHostToDevice, c->stream));
  }
  ELP_HIP(c, elp::stream_wait(c->stream));
  c->h_ref_seq.resize(h->n_ref, nullptr);
  c->h_ref_seq_len.resize(h->n_ref, 0);
  c->h_sites.resize(h->n_ref, nullptr);
  c->h_site_idx.resize(h->n_ref, nullptr);
  c->h_n_sites.resize(h->n_ref, 0);
  c->ref_flags_dirty.assign(h->n_ref, 0);
  c->bqsr_ptrs_dirty = true;
  c->have_header = true;
  return 0;
}

}  // extern "C"
namespace elp {
int stage_reserve(elp_ctx *c, uint64_t n, uint64_t qb, uint64_t co, uint64_t sb, uint64_t lb) {
  bool keep = c->n > 0;
  ELP_TRY(ensure(c, c->refid, n, keep, c->n));
  ELP_TRY(ensure(c, c->pos, n, keep, c->n));
  ELP_TRY(ensure(c, c->next_refid, n, keep, c->n));
  ELP_TRY(ensure(c, c->pnext, n, keep, c->n));
  ELP_TRY(ensure(c, c->tlen, n, keep, c->n));
  ELP_TRY(ensure(c, c->flag, n, keep, c->n));
  ELP_TRY(ensure(c, c->rgid, n, keep, c->n));
  ELP_TRY(ensure(c, c->split, n, keep, c->n));
  ELP_TRY(ensure(c, c->mapq, n, keep, c->n));
  ELP_TRY(ensure(c, c->has_sr, n, keep, c->n));
  ELP_TRY(ensure(c, c->l_seq, n, keep, c->n));
  ELP_TRY(ensure(c, c->qname_off, n + 1, keep, c->n + 1));
  ELP_TRY(ensure(c, c->cigar_off, n + 1, keep, c->n + 1));
  ELP_TRY(ensure(c, c->seq_off, n + 1, keep, c->n + 1));
  ELP_TRY(ensure(c, c->qual_off, n + 1, keep, c->n + 1));
  ELP_TRY(ensure(c, c->qname, qb + 64, keep, c->qname_bytes));
  ELP_TRY(ensure(c, c->cigar, co + 4, keep, c->cigar_ops));
  ELP_TRY(ensure(c, c->seq4, sb + 32, keep, c->seq_bytes));
  ELP_TRY(ensure(c, c->qual, lb + 32, keep, c->qual_bytes));
  return 0;
}
}  // namespace elp
extern "C" {

int elp_reserve(elp_ctx *c, uint64_t n, uint64_t qb, uint64_t co, uint64_t sb, uint64_t lb) {
  if (!c) return ELP_ERR_ARG;
  std::lock_guard<std::mutex> g(c->stage_mu);
  ELP_HIP(c, hipSetDevice(c->device));
  // (the SEQ column starts SEQ_FRONT bytes into its allocation: a caller that sizes it exactly must not trigger a regrow on the first stage)
  return stage_reserve(c, n, qb, co, sb + elp_ctx::SEQ_FRONT, lb);
}

int elp_reset(elp_ctx *c) {
  if (!c) return ELP_ERR_ARG;
  std::lock_guard<std::mutex> g(c->stage_mu);
  c->n = c->qname_bytes = c->cigar_ops = c->qual_bytes = 0;
  c->seq_bytes = elp_ctx::SEQ_FRONT;
  c->n_sr = 0;
  c->n_filtered = 0;
  c->raw_n = c->raw_bytes = 0;
  c->max_raw_rec = 0;
  c->max_split = 0;
  c->max_qname_len = c->max_l_seq = 0;
  c->max_pos = 0;
  c->adapted = c->sorted = c->marked = false;
  c->have_qual_present = false;
  c->have_snapshot = false;
  c->flat_index_n = 0;
  c->uniform_n = ~0ull;
  return 0;
}

uint64_t elp_num_records(const elp_ctx *c) { return c ? c->n : 0; }
uint64_t elp_num_sorted(const elp_ctx *c) { return c ? c->n - c->n_sr : 0; }
uint64_t elp_num_qual_bytes(const elp_ctx *c) { return c ? c->qual_bytes : 0; }

int elp_stage(elp_ctx *c, const elp_batch *b) {
  if (!c || !b) return ELP_ERR_ARG;
  std::lock_guard<std::mutex> g(c->stage_mu);
  ELP_HIP(c, hipSetDevice(c->device));
  if (!c->have_header) return set_error(c, ELP_ERR_ARG, "elp_stage: call elp_set_header first");
  uint64_t n = b->n;
  if (n == 0) return 0;
  if (c->n + n > 0xFFFFFFF0ull) return set_error(c, ELP_ERR_UNSUPPORTED, "more than 2^32-16 records per context");
  uint64_t q0 = b->qname_off[0], c0 = b->cigar_off[0], s0 = b->seq_off[0], l0 = b->qual_off[0];
  uint64_t qb = b->qname_off[n] - q0, co = b->cigar_off[n] - c0, sb = b->seq_off[n] - s0, lb = b->qual_off[n] - l0;
  ELP_TRY(stage_reserve(c, c->n + n, c->qname_bytes + qb, c->cigar_ops + co, c->seq_bytes + sb, c->qual_bytes + lb));
  // All copies are issued first, behind the staged records (nothing is committed yet); the host-side scan for the limits the kernels
  // rely on runs while the DMA engine works (from page-locked columns - elp_pinned_alloc - the copies are asynchronous and the
  // scan is hidden; from pageable memory the runtime stages every copy itself and the call is bound by that).
  hipStream_t st = c->stream;
  uint64_t at = c->n;
#define H2D(dst, src, cnt, T) ELP_HIP(c, hipMemcpyAsync((dst), (src), (cnt) * sizeof(T), hipMemcpyHostToDevice, st))
  if (lb) H2D(c->qual.p + c->qual_bytes, b->qual + l0, lb, uint8_t);
  if (sb) {
    H2D(c->seq4.p + c->seq_bytes, b->seq4 + s0, sb, uint8_t);
    hipLaunchKernelGGL(k_recode_seq, dim3(blocks_for((sb + 15) / 16, 256)), dim3(256), 0, st, c->seq4.p + c->seq_bytes, sb);
    ELP_HIP(c, hipGetLastError());
  }
  if (qb) H2D(c->qname.p + c->qname_bytes, b->qname + q0, qb, uint8_t);
  if (co) H2D(c->cigar.p + c->cigar_ops, b->cigar + c0, co, uint32_t);
  H2D(c->refid.p + at, b->refid, n, int32_t);
  H2D(c->pos.p + at, b->pos, n, int32_t);
  H2D(c->next_refid.p + at, b->next_refid, n, int32_t);
  H2D(c->pnext.p + at, b->pnext, n, int32_t);
  H2D(c->tlen.p + at, b->tlen, n, int32_t);
  H2D(c->flag.p + at, b->flag, n, uint16_t);
  H2D(c->rgid.p + at, b->rgid, n, uint16_t);
  if (b->split) H2D(c->split.p + at, b->split, n, uint16_t);
  else ELP_HIP(c, hipMemsetAsync(c->split.p + at, 0, n * sizeof(uint16_t), st));
  H2D(c->mapq.p + at, b->mapq, n, uint8_t);
  H2D(c->l_seq.p + at, b->l_seq, n, uint32_t);
  if (b->has_sr) H2D(c->has_sr.p + at, b->has_sr, n, uint8_t);
  else ELP_HIP(c, hipMemsetAsync(c->has_sr.p + at, 0, n, st));
  // offsets: copy raw (four slices of the scratch column, so that no copy waits for a kernel), rebase on device
  ELP_TRY(ensure(c, c->stage_tmp, 4 * (n + 1)));
  unsigned blk = 256, grd = blocks_for(n + 1, blk);
  struct { const uint64_t *src; uint64_t *dst; uint64_t base; } offs[4] = {
      {b->qname_off, c->qname_off.p + at, c->qname_bytes}, {b->cigar_off, c->cigar_off.p + at, c->cigar_ops},
      {b->seq_off, c->seq_off.p + at, c->seq_bytes}, {b->qual_off, c->qual_off.p + at, c->qual_bytes}};
  for (int k = 0; k < 4; k++) H2D(c->stage_tmp.p + (size_t)k * (n + 1), offs[k].src, n + 1, uint64_t);
  for (int k = 0; k < 4; k++) {
    hipLaunchKernelGGL(k_rebase_offsets, dim3(grd), dim3(blk), 0, st, (const uint64_t *)(c->stage_tmp.p + (size_t)k * (n + 1)), offs[k].dst, n + 1, offs[k].base,
                       offs[k].src[0]);
    ELP_HIP(c, hipGetLastError());
  }
#undef H2D
  // host-side scan for limits the kernels rely on (an error leaves the context as it was: nothing has been committed)
  uint64_t n_sr = 0;
  uint32_t max_split = c->max_split;
  uint32_t max_qname_len = c->max_qname_len, max_l_seq = c->max_l_seq, max_pos = c->max_pos;  // committed when the batch is
  int scan_rc = 0;
  for (uint64_t i = 0; i < n && !scan_rc; i++) {
    uint64_t ql = b->qname_off[i + 1] - b->qname_off[i];
    if (ql > elp_ctx::MAX_QNAME) scan_rc = set_error(c, ELP_ERR_UNSUPPORTED, "record %llu: QNAME of %llu bytes (limit %u)", (unsigned long long)i, (unsigned long long)ql, elp_ctx::MAX_QNAME);
    if (b->has_sr && b->has_sr[i]) n_sr++;
    if (b->split && b->split[i] > max_split) max_split = b->split[i];
    if (ql > max_qname_len) max_qname_len = (uint32_t)ql;
    if (b->l_seq[i] > max_l_seq) max_l_seq = b->l_seq[i];
    if ((uint32_t)b->pos[i] > max_pos) max_pos = (uint32_t)b->pos[i];
    if (b->qual_off[i + 1] - b->qual_off[i] > 0x3FFFFFull || b->l_seq[i] > 0x3FFFFFu)  // FL_MAX_READ (flat.hpp)
      scan_rc = set_error(c, ELP_ERR_UNSUPPORTED, "record %llu: more than 4194303 bases", (unsigned long long)i);
    if (b->rgid[i] != ELP_NIL16 && b->rgid[i] >= c->n_rg) scan_rc = set_error(c, ELP_ERR_ARG, "record %llu: rgid %u not in header", (unsigned long long)i, b->rgid[i]);
    if (b->refid[i] >= c->n_ref) scan_rc = set_error(c, ELP_ERR_ARG, "record %llu: refid %d not in header", (unsigned long long)i, b->refid[i]);
  }
  if (scan_rc) { (void)elp::stream_wait(st); return scan_rc; }
  ELP_HIP(c, elp::stream_wait(st));  // host buffers may be reused on return
  c->n += n; c->qname_bytes += qb; c->cigar_ops += co; c->seq_bytes += sb; c->qual_bytes += lb;
  c->n_sr += n_sr;
  c->max_split = max_split;
  c->max_qname_len = max_qname_len; c->max_l_seq = max_l_seq; c->max_pos = max_pos;
  c->adapted = c->sorted = c->marked = false;
  c->have_qual_present = false;
  c->have_snapshot = false;
  c->flat_index_n = 0;
  c->uniform_n = ~0ull;
  return 0;
}

// The same with every column as an argument of its own.  cgo may pass a Go pointer to C only if the memory it points to holds no
// Go pointers (cmd/cgo "Passing pointers"); an elp_batch / elp_header in Go memory whose fields point at Go slices breaks that rule,
// a call that takes each slice's first element does not.  (With columns in C memory - elp_pinned_alloc - the struct forms are fine.)
int elp_stage_columns(elp_ctx *c, uint64_t n, const int32_t *refid, const int32_t *pos, const int32_t *next_refid, const int32_t *pnext, const int32_t *tlen,
                      const uint16_t *flag, const uint8_t *mapq, const uint16_t *rgid, const uint8_t *has_sr, const uint32_t *l_seq,
                      const uint64_t *qname_off, const uint8_t *qname, const uint64_t *cigar_off, const uint32_t *cigar, const uint64_t *seq_off,
                      const uint8_t *seq4, const uint64_t *qual_off, const uint8_t *qual, const uint16_t *split) {
  if (!c) return ELP_ERR_ARG;
  if (n && (!refid || !pos || !next_refid || !pnext || !tlen || !flag || !mapq || !rgid || !l_seq || !qname_off || !cigar_off || !seq_off || !qual_off))
    return set_error(c, ELP_ERR_ARG, "elp_stage_columns: a required column is NULL");
  elp_batch b;
  b.n = n; b.refid = refid; b.pos = pos; b.next_refid = next_refid; b.pnext = pnext; b.tlen = tlen; b.flag = flag; b.mapq = mapq; b.rgid = rgid;
  b.has_sr = has_sr; b.l_seq = l_seq; b.qname_off = qname_off; b.qname = qname; b.cigar_off = cigar_off; b.cigar = cigar; b.seq_off = seq_off;
  b.seq4 = seq4; b.qual_off = qual_off; b.qual = qual; b.split = split;
  return elp_stage(c, &b);
}

int elp_set_header_columns(elp_ctx *c, int32_t n_ref, const int32_t *ref_len, int32_t n_rg, const uint16_t *rg_lib, const uint16_t *rg_cov, int32_t n_lib,
                           int32_t n_cov) {
  elp_header h;
  h.n_ref = n_ref; h.ref_len = ref_len; h.n_rg = n_rg; h.rg_lib = rg_lib; h.rg_cov = rg_cov; h.n_lib = n_lib; h.n_cov = n_cov;
  return elp_set_header(c, &h);
}

static int d2h(elp_ctx *c, void *dst, const void *src, size_t bytes) {
  if (!bytes) return 0;
  ELP_HIP(c, hipSetDevice(c->device));
  ELP_HIP(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
  ELP_HIP(c, elp::stream_wait(c->stream));
  return 0;
}

int elp_get_permutation(elp_ctx *c, uint32_t *out) {
  if (!c || (!out && c->n)) return ELP_ERR_ARG;
  if (!c->sorted) return set_error(c, ELP_ERR_ARG, "elp_get_permutation: call elp_sort_coordinate first");
  ELP_TRY(radix_check(c));
  return d2h(c, out, c->perm.p, c->n * sizeof(uint32_t));
}
int elp_get_flags(elp_ctx *c, uint16_t *out) {
  if (!c || (!out && c->n)) return ELP_ERR_ARG;
  ELP_TRY(radix_check(c));  // (mark duplicates partitions its pairs with radix passes)
  return d2h(c, out, c->flag.p, c->n * sizeof(uint16_t));
}
int elp_get_adapted(elp_ctx *c, int32_t *upos, int32_t *score) {
  if (!c) return ELP_ERR_ARG;
  ELP_TRY(ensure_adapted(c));
  if (upos) ELP_TRY(d2h(c, upos, c->upos.p, c->n * sizeof(int32_t)));
  if (score) ELP_TRY(d2h(c, score, c->score.p, c->n * sizeof(int32_t)));
  return 0;
}
int elp_get_qual(elp_ctx *c, uint8_t *out) {
  if (!c || (!out && c->qual_bytes)) return ELP_ERR_ARG;
  return d2h(c, out, c->qual.p, c->qual_bytes);
}

int elp_snapshot(elp_ctx *c) {
  if (!c) return ELP_ERR_ARG;
  ELP_HIP(c, hipSetDevice(c->device));
  ELP_TRY(ensure(c, c->snap_flag, c->n + 1));
  ELP_TRY(ensure(c, c->snap_qual, c->qual_bytes + 16));
  if (c->n) ELP_HIP(c, hipMemcpyAsync(c->snap_flag.p, c->flag.p, c->n * sizeof(uint16_t), hipMemcpyDeviceToDevice, c->stream));
  if (c->qual_bytes) ELP_HIP(c, hipMemcpyAsync(c->snap_qual.p, c->qual.p, c->qual_bytes, hipMemcpyDeviceToDevice, c->stream));
  c->snap_n = c->n;
  c->snap_qual_bytes = c->qual_bytes;
  c->have_snapshot = true;
  return 0;
}
int elp_rollback(elp_ctx *c) {
  if (!c) return ELP_ERR_ARG;
  if (!c->have_snapshot || c->snap_n != c->n || c->snap_qual_bytes != c->qual_bytes)
    return set_error(c, ELP_ERR_ARG, "elp_rollback: no snapshot of the current record set");
  ELP_HIP(c, hipSetDevice(c->device));
  if (c->n) ELP_HIP(c, hipMemcpyAsync(c->flag.p, c->snap_flag.p, c->n * sizeof(uint16_t), hipMemcpyDeviceToDevice, c->stream));
  if (c->qual_bytes) ELP_HIP(c, hipMemcpyAsync(c->qual.p, c->snap_qual.p, c->qual_bytes, hipMemcpyDeviceToDevice, c->stream));
  c->adapted = c->sorted = c->marked = false;
  c->have_qual_present = false;
  return 0;
}

int elp_set_tuning(elp_ctx *c, const char *key, int64_t value) {
  if (!c || !key) return ELP_ERR_ARG;
  const std::string k(key);
  const int v = (int)value;
  if (k == "count_kernel") c->tune.count_kernel = v;
  else if (k == "apply_kernel") c->tune.apply_kernel = v;
  else if (k == "bgzf_piece") { if (value < 1) return set_error(c, ELP_ERR_ARG, "elp_set_tuning: bgzf_piece must be positive"); c->tune.bgzf_piece = value; }
  else if (k == "bgzf_weak_guess") c->tune.bgzf_weak_guess = v;
  else if (k == "score_kernel") { c->tune.score_kernel = v; c->adapted = false; }
  else if (k == "count3_rlog") c->tune.count3_rlog = v;
  else if (k == "qual_hint") { c->tune.qual_hint = v; c->have_qual_present = false; }
  else if (k == "qual_hint_drop") { c->tune.qual_hint_drop = v; c->have_qual_present = false; }
  else if (k == "pair_table_slots") {
    if (v == 0) { c->tune.pair_table_slots = 1 << 20; return 0; }
    if (v < 2 || v > (1 << 20) || (v & (v - 1))) return set_error(c, ELP_ERR_ARG, "elp_set_tuning: pair_table_slots must be a power of two >= 2");
    c->tune.pair_table_slots = v;
  } else if (k == "mate_path") c->tune.mate_path = v;
  else if (k == "tie_rounds") c->tune.tie_rounds = v;
  else if (k == "radix_tile") c->tune.radix_tile = v;
  else if (k == "sort_pairs") c->tune.sort_pairs = v;
  else if (k == "exchange_piece") c->tune.exchange_piece = v;
  else if (k == "bgzf_stored") c->tune.bgzf_stored = v;
  else if (k == "bgzf_inflate") c->tune.bgzf_inflate = v;
  else if (k == "bgzf_tok_fail_above") c->tune.bgzf_tok_fail_above = v;
  else if (k == "bgzf_fixed") c->tune.bgzf_fixed = v;
  else if (k == "bgzf_first_chunk_div") c->tune.bgzf_first_chunk_div = v;
  else if (k == "bgzf_tok_lds") c->tune.bgzf_tok_lds = v;
  else if (k == "bgzf_copy_chunk") c->tune.bgzf_copy_chunk = v;
  else if (k == "bgzf_inflate_piece") { if (value < 1) return set_error(c, ELP_ERR_ARG, "elp_set_tuning: bgzf_inflate_piece must be positive"); c->tune.bgzf_inflate_piece = value; }
  else if (k == "md_fused") c->tune.md_fused = v;
  else if (k == "apply_wgs") c->tune.apply_wgs = v;
  else if (k == "presort_tile") c->tune.presort_tile = v;
  else if (k == "side_priority") c->tune.side_priority = v;
  else return set_error(c, ELP_ERR_ARG, "elp_set_tuning: unknown key '%s'", key);
  return 0;
}

int elp_profile_enable(elp_ctx *c, int on) {
  if (!c) return ELP_ERR_ARG;
  if (!on) { ELP_TRY(prof_flush(c)); prof_merge_side(c); }
  c->profiling = on != 0;
  for (int lane = 0; lane < 2; lane++)
    if (c->side[lane]) c->side[lane]->profiling = c->profiling;
  return 0;
}
int elp_profile_reset(elp_ctx *c) {
  if (!c) return ELP_ERR_ARG;
  ELP_TRY(prof_flush(c));
  prof_merge_side(c);
  for (auto &v : c->prof_launches) v = 0;
  for (auto &v : c->prof_ms) v = 0.0;
  return 0;
}
int elp_profile_count(elp_ctx *c) {
  if (!c) return ELP_ERR_ARG;
  if (prof_flush(c) != 0) return ELP_ERR_HIP;
  prof_merge_side(c);
  return (int)c->prof_names.size();
}
int elp_profile_get(elp_ctx *c, int index, const char **name, uint64_t *launches, double *total_ms) {
  if (!c || index < 0 || index >= (int)c->prof_names.size()) return ELP_ERR_ARG;
  if (name) *name = c->prof_names[index].c_str();
  if (launches) *launches = c->prof_launches[index];
  if (total_ms) *total_ms = c->prof_ms[index];
  return 0;
}

}  // extern "C"
