// group.hip — the device group of an `sfm` run and its one collective, behind the C ABI.
//
// Reference: the merge phase of `elprep sfm` sums what the per-split `filter --bqsr-tables-only` runs wrote:
//   LoadAndCombineBQSRTables        filters/print-bqsr.go:310-329          (three Go maps, entry-wise sums)
//   LoadAndCombineDuplicateMetrics  filters/mark-optical-duplicates.go:711-731 (seven counters per library)
// Here every rank (one process per GPU, one elp_ctx per split it owns) keeps its dense int64 tables in HBM
// (elp_bqsr_gather_device), adds the tables of its other contexts to them on the device (elp_bqsr_tables_add), and ONE
// ncclAllReduce(ncclInt64, ncclSum) over xGMI on the context's stream merges tables and counters of all ranks in place
// (elp_bqsr_tables_allreduce).  Integer sums: bit-exact whatever the reduction order.
//
// RCCL is bound at run time (dlopen): a single-GPU `filter` run needs no communication library, and inside a process
// that already carries an RCCL (PyTorch-ROCm does) the same instance is used.
#include <dlfcn.h>

#include <rccl/rccl.h>

#include <algorithm>

#include "common.hpp"

namespace elp {

struct Rccl {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  std::string err;
};

static Rccl *rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) { r.err = std::string("cannot load RCCL: ") + dlerror(); return; }
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
    r.Send = reinterpret_cast<decltype(r.Send)>(dlsym(r.lib, "ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(dlsym(r.lib, "ncclRecv"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(dlsym(r.lib, "ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(dlsym(r.lib, "ncclGroupEnd"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd)
      r.err = "RCCL library lacks a required symbol";
  });
  return &r;
}

#define ELP_NCCL(ctx, call)                                                                                    \
  do {                                                                                                         \
    ncclResult_t r__ = (call);                                                                                 \
    if (r__ != ncclSuccess) return elp::set_error((ctx), ELP_ERR_HIP, "%s failed: %s", #call, R->GetErrorString(r__)); \
  } while (0)

__global__ __launch_bounds__(256) void k_add_i64(unsigned long long *__restrict__ dst, const unsigned long long *__restrict__ src, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

// (a context that borrowed the group of another - elp_group_share - must not keep a communicator its owner destroyed: the
// owner knows its borrowers and takes the group from them when it lets go of it; a borrower signs off at its owner's)
static std::mutex g_share_mu;
void group_release(elp_ctx *c) {
  {
    std::lock_guard<std::mutex> g(g_share_mu);
    if (c->group_owner) {
      auto &v = c->group_owner->group_borrowers;
      v.erase(std::remove(v.begin(), v.end(), c), v.end());
      c->group_owner = nullptr;
    }
    for (elp_ctx *b : c->group_borrowers) {  // this context's group goes away: so does every borrower's view of it
      b->comm = nullptr; b->comm_borrowed = false;
      b->xport = nullptr; b->xport_user = nullptr; b->p2p = nullptr; b->p2p_user = nullptr;
      b->group_rank = 0; b->group_world = 1;
      b->group_owner = nullptr;
    }
    c->group_borrowers.clear();
  }
  if (c->comm) {
    Rccl *R = rccl();
    if (R->CommDestroy && !c->comm_borrowed) (void)R->CommDestroy(static_cast<ncclComm_t>(c->comm));
    c->comm = nullptr;
    c->comm_borrowed = false;
  }
  c->xport = nullptr;  // a transport of an earlier elp_group_init_transport does not outlive the group either
  c->xport_user = nullptr;
  c->p2p = nullptr;
  c->p2p_user = nullptr;
}

// one message each way between this rank and two peers of the device group (either may be absent: peer -1 or no bytes); device buffers
int group_sendrecv(elp_ctx *c, int send_peer, const void *send_dev, size_t send_bytes, int recv_peer, void *recv_dev, size_t recv_bytes) {
  if (send_peer < 0) send_bytes = 0;
  if (recv_peer < 0) recv_bytes = 0;
  if (!send_bytes && !recv_bytes) return 0;
  ELP_HIP(c, hipSetDevice(c->device));
  if (c->p2p) {  // the caller's transport, through page-locked host memory
    const size_t need = send_bytes + recv_bytes + 64;
    if (need > c->h_pinned_cap) {
      if (c->h_pinned) (void)hipHostFree(c->h_pinned);
      c->h_pinned = nullptr; c->h_pinned_cap = 0;
      ELP_HIP(c, hipHostMalloc(&c->h_pinned, need, hipHostMallocDefault));
      c->h_pinned_cap = need;
    }
    uint8_t *hs = static_cast<uint8_t *>(c->h_pinned), *hr = hs + ((send_bytes + 63) & ~(size_t)63);
    if (send_bytes) ELP_HIP(c, hipMemcpyAsync(hs, send_dev, send_bytes, hipMemcpyDeviceToHost, c->stream));
    ELP_HIP(c, elp::stream_wait(c->stream));
    const int rc = c->p2p(c->p2p_user, send_bytes ? send_peer : -1, hs, send_bytes, recv_bytes ? recv_peer : -1, hr, recv_bytes);
    if (rc != 0) return set_error(c, ELP_ERR_HIP, "the group's transport failed (send-receive callback returned %d)", rc);
    if (recv_bytes) ELP_HIP(c, hipMemcpyAsync(recv_dev, hr, recv_bytes, hipMemcpyHostToDevice, c->stream));
    ELP_HIP(c, elp::stream_wait(c->stream));  // (the pinned buffer is reused by the next message)
    return 0;
  }
  if (!c->comm) return set_error(c, ELP_ERR_ARG, "no device group with point-to-point transport: call elp_group_init (RCCL) or elp_group_set_p2p first");
  Rccl *R = rccl();
  ELP_NCCL(c, R->GroupStart());  // (send and receive of one step progress together: no rank waits for its own send before it posts its receive)
  if (send_bytes) ELP_NCCL(c, R->Send(send_dev, send_bytes, ncclUint8, send_peer, static_cast<ncclComm_t>(c->comm), c->stream));
  if (recv_bytes) ELP_NCCL(c, R->Recv(recv_dev, recv_bytes, ncclUint8, recv_peer, static_cast<ncclComm_t>(c->comm), c->stream));
  ELP_NCCL(c, R->GroupEnd());
  return 0;
}

}  // namespace elp

using namespace elp;

extern "C" {

int elp_group_probe(void) {
  Rccl *R = rccl();
  return R->err.empty() ? 0 : ELP_ERR_UNSUPPORTED;
}

int elp_group_unique_id(uint8_t *id_out) {
  if (!id_out) return ELP_ERR_ARG;
  Rccl *R = rccl();
  if (!R->err.empty()) return ELP_ERR_UNSUPPORTED;
  static_assert(sizeof(ncclUniqueId) == ELP_GROUP_ID_BYTES, "ELP_GROUP_ID_BYTES is RCCL's NCCL_UNIQUE_ID_BYTES");
  ncclUniqueId id;
  if (R->GetUniqueId(&id) != ncclSuccess) return ELP_ERR_HIP;
  memcpy(id_out, &id, sizeof id);
  return 0;
}

int elp_group_init(elp_ctx *c, int rank, int world, const uint8_t *id) {
  if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) return set_error(c, ELP_ERR_ARG, "elp_group_init: bad arguments");
  ELP_HIP(c, hipSetDevice(c->device));
  group_release(c);
  c->group_rank = rank;
  c->group_world = world;
  if (world == 1) return 0;  // a group of one needs no communicator
  Rccl *R = rccl();
  if (!R->err.empty()) return set_error(c, ELP_ERR_UNSUPPORTED, "%s", R->err.c_str());
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof uid);
  ncclComm_t comm = nullptr;
  ELP_NCCL(c, R->CommInitRank(&comm, world, uid, rank));
  c->comm = comm;
  return 0;
}

int elp_group_init_transport(elp_ctx *c, int rank, int world, elp_allreduce_fn allreduce, void *user) {
  if (!c || world < 1 || rank < 0 || rank >= world || (world > 1 && !allreduce)) return set_error(c, ELP_ERR_ARG, "elp_group_init_transport: bad arguments");
  if (c->comm) return set_error(c, ELP_ERR_ARG, "elp_group_init_transport: the context already belongs to an RCCL group");
  c->group_rank = rank;
  c->group_world = world;
  c->xport = world > 1 ? allreduce : nullptr;
  c->xport_user = user;
  return 0;
}

int elp_group_set_p2p(elp_ctx *c, elp_sendrecv_fn fn, void *user) {
  if (!c) return ELP_ERR_ARG;
  if (c->comm) return set_error(c, ELP_ERR_ARG, "elp_group_set_p2p: the context belongs to an RCCL group (its send / receive are RCCL's)");
  c->p2p = fn;
  c->p2p_user = user;
  return 0;
}

int elp_group_share(elp_ctx *c, elp_ctx *member) {
  if (!c || !member || c == member) return ELP_ERR_ARG;
  if (c->device != member->device) return set_error(c, ELP_ERR_ARG, "elp_group_share: the two contexts live on different devices");
  group_release(c);
  c->group_rank = member->group_rank;
  c->group_world = member->group_world;
  c->comm = member->comm;
  c->comm_borrowed = member->comm != nullptr;
  c->xport = member->xport; c->xport_user = member->xport_user;
  c->p2p = member->p2p; c->p2p_user = member->p2p_user;
  {
    std::lock_guard<std::mutex> g(g_share_mu);
    elp_ctx *owner = member->group_owner ? member->group_owner : member;  // (a borrower's group is its owner's)
    c->group_owner = owner;
    owner->group_borrowers.push_back(c);
  }
  return 0;
}

int elp_group_rank(const elp_ctx *c) { return c ? c->group_rank : -1; }
int elp_group_size(const elp_ctx *c) { return c ? c->group_world : 0; }

// sum over the group of n int64 values in device memory, in place, on the ctx stream (no host hop)
static int allreduce_device(elp_ctx *c, unsigned long long *buf, size_t n) {
  if (c->group_world <= 1 || n == 0) return 0;
  if (c->xport) {  // the caller's transport: through page-locked host memory
    const size_t bytes = n * 8;
    if (bytes > c->h_pinned_cap) {
      if (c->h_pinned) (void)hipHostFree(c->h_pinned);
      c->h_pinned = nullptr; c->h_pinned_cap = 0;
      ELP_HIP(c, hipHostMalloc(&c->h_pinned, bytes, hipHostMallocDefault));
      c->h_pinned_cap = bytes;
    }
    ELP_HIP(c, hipMemcpyAsync(c->h_pinned, buf, bytes, hipMemcpyDeviceToHost, c->stream));
    ELP_HIP(c, elp::stream_wait(c->stream));
    const int rc = c->xport(c->xport_user, static_cast<int64_t *>(c->h_pinned), n);
    if (rc != 0) return set_error(c, ELP_ERR_HIP, "the group's transport failed (allreduce callback returned %d)", rc);
    ELP_HIP(c, hipMemcpyAsync(buf, c->h_pinned, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
  }
  if (!c->comm) return set_error(c, ELP_ERR_ARG, "no device group: call elp_group_init first");
  Rccl *R = rccl();
  ELP_NCCL(c, R->AllReduce(buf, buf, n, ncclInt64, ncclSum, static_cast<ncclComm_t>(c->comm), c->stream));
  return 0;
}

int elp_allreduce_i64(elp_ctx *c, int64_t *buf, size_t n) {
  if (!c || (!buf && n)) return ELP_ERR_ARG;
  if (c->group_world <= 1 || n == 0) return 0;
  ELP_HIP(c, hipSetDevice(c->device));
  unsigned long long *d;
  ELP_TRY(scratch(c, 7, n + 8, &d));
  ELP_HIP(c, hipMemcpyAsync(d, buf, n * 8, hipMemcpyHostToDevice, c->stream));
  ELP_TRY(allreduce_device(c, d, n));
  ELP_HIP(c, hipMemcpyAsync(buf, d, n * 8, hipMemcpyDeviceToHost, c->stream));
  ELP_HIP(c, elp::stream_wait(c->stream));
  return 0;
}

int elp_bqsr_tables_add(elp_ctx *dst, elp_ctx *src) {
  if (!dst || !src || dst == src) return ELP_ERR_ARG;
  if (!dst->tables_n || dst->tables_n != src->tables_n || dst->tables_max_cycle != src->tables_max_cycle || dst->n_cov != src->n_cov ||
      dst->device != src->device)
    return set_error(dst, ELP_ERR_ARG, "elp_bqsr_tables_add: both contexts need device tables of the same shape on the same device (elp_bqsr_gather_device)");
  ELP_HIP(dst, hipSetDevice(dst->device));
  ELP_HIP(dst, elp::stream_wait(src->stream));  // src's gather kernels are done before dst's stream reads its tables
  hipLaunchKernelGGL(k_add_i64, dim3(blocks_for(dst->tables_n, 256)), dim3(256), 0, dst->stream, dst->dev_tables.p, (const unsigned long long *)src->dev_tables.p,
                     dst->tables_n);
  ELP_HIP(dst, hipGetLastError());
  return tables_written(dst);
}

int elp_bqsr_tables_allreduce(elp_ctx *c, int64_t *counters, size_t n_counters) {
  if (!c || (!counters && n_counters)) return ELP_ERR_ARG;
  if (!c->tables_n) return set_error(c, ELP_ERR_ARG, "elp_bqsr_tables_allreduce: no device tables (elp_bqsr_gather_device)");
  ELP_HIP(c, hipSetDevice(c->device));
  if (c->group_world <= 1) return 0;
  // the counters ride behind the tables: one collective for both (the tables' allocation has room for ELP_TABLES_TAIL values)
  if (n_counters > elp_ctx::TABLES_TAIL) return set_error(c, ELP_ERR_ARG, "elp_bqsr_tables_allreduce: at most %zu extra values", (size_t)elp_ctx::TABLES_TAIL);
  unsigned long long *tail = c->dev_tables.p + c->tables_n;
  if (n_counters) ELP_HIP(c, hipMemcpyAsync(tail, counters, n_counters * 8, hipMemcpyHostToDevice, c->stream));
  ELP_TRY(allreduce_device(c, c->dev_tables.p, c->tables_n + n_counters));
  ELP_TRY(tables_written(c));
  if (n_counters) ELP_HIP(c, hipMemcpyAsync(counters, tail, n_counters * 8, hipMemcpyDeviceToHost, c->stream));
  ELP_HIP(c, elp::stream_wait(c->stream));
  return 0;
}

}  // extern "C"
