// What does one wave64 instruction cost on gfx950?  Issue rates of the integer VALU / SALU / LDS / byte-load instructions the
// per-base kernels (bqsr_count, bqsr_apply) are made of, measured with every SIMD saturated (8 waves per SIMD, 8 independent chains
// per wave).  Sizing question behind the round-3 redesign of those kernels (0.51 VALU wave-instructions per base).
// Not part of the product.
// build: hipcc -O3 --offload-arch=gfx950 isa_rate_probe.hip -o isa_rate_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int CHAINS = 8, UNROLL = 8;

// one asm statement per chain; OPS = instructions per statement (for the rate)
#define VALU32(NAME, ASM)                                                                                           \
  __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters, uint32_t seed) {                            \
    uint32_t a[CHAINS];                                                                                             \
    for (int c = 0; c < CHAINS; c++) a[c] = threadIdx.x * 2654435761u + seed + c;                                   \
    uint32_t k = seed | 1u, k2 = seed * 7u + 3u;                                                                    \
    unsigned long long m = 0x5555AAAA3333CCCCull ^ seed;                                                            \
    for (int i = 0; i < iters; i++) {                                                                               \
      _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                                          \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "+v"(a[c]) : "v"(k), "v"(k2), "s"(m)); \
      }                                                                                                             \
    }                                                                                                               \
    uint32_t r = 0;                                                                                                 \
    for (int c = 0; c < CHAINS; c++) r ^= a[c];                                                                     \
    if (r == 0x12345u) out[0] = r;                                                                                  \
  }
#define VALU64(NAME, ASM)                                                                                           \
  __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters, uint32_t seed) {                            \
    unsigned long long a[CHAINS];                                                                                   \
    for (int c = 0; c < CHAINS; c++) a[c] = (threadIdx.x * 2654435761ull + seed + c) * 0x9e3779b97f4a7c15ull;       \
    uint32_t k = (seed & 15u) | 1u, k2 = seed * 7u + 3u;                                                            \
    for (int i = 0; i < iters; i++) {                                                                               \
      _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                                          \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "+v"(a[c]) : "v"(k), "v"(k2));        \
      }                                                                                                             \
    }                                                                                                               \
    unsigned long long r = 0;                                                                                       \
    for (int c = 0; c < CHAINS; c++) r ^= a[c];                                                                     \
    if (r == 0x12345u) out[0] = (uint32_t)r;                                                                        \
  }
// compare -> SGPR mask
#define VCMP(NAME, ASM)                                                                                             \
  __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters, uint32_t seed) {                            \
    unsigned long long s[CHAINS];                                                                                   \
    uint32_t a = threadIdx.x * 2654435761u + seed, k = seed | 1u;                                                   \
    for (int c = 0; c < CHAINS; c++) s[c] = 0;                                                                      \
    for (int i = 0; i < iters; i++) {                                                                               \
      _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                                          \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "=s"(s[c]) : "v"(a), "v"(k));         \
      }                                                                                                             \
    }                                                                                                               \
    unsigned long long r = 0;                                                                                       \
    for (int c = 0; c < CHAINS; c++) r ^= s[c];                                                                     \
    if (r == 0x12345u) out[0] = (uint32_t)r;                                                                        \
  }
#define SALU64(NAME, ASM)                                                                                           \
  __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters, uint32_t seed) {                            \
    unsigned long long s[CHAINS];                                                                                   \
    unsigned long long k = 0x5555AAAA3333CCCCull ^ seed;                                                            \
    for (int c = 0; c < CHAINS; c++) s[c] = k * (c + 3);                                                            \
    for (int i = 0; i < iters; i++) {                                                                               \
      _Pragma("unroll") for (int u = 0; u < UNROLL; u++) {                                                          \
        _Pragma("unroll") for (int c = 0; c < CHAINS; c++) asm volatile(ASM : "+s"(s[c]) : "s"(k) : "scc");         \
      }                                                                                                             \
    }                                                                                                               \
    unsigned long long r = 0;                                                                                       \
    for (int c = 0; c < CHAINS; c++) r ^= s[c];                                                                     \
    if (r == 0x12345u) out[0] = (uint32_t)r;                                                                        \
  }

VALU32(k_add, "v_add_u32 %0, %0, %1")
VALU32(k_and, "v_and_b32 %0, %0, %1")
VALU32(k_xor, "v_xor_b32 %0, %0, %1")
VALU32(k_bfe, "v_bfe_u32 %0, %0, 3, 9")
VALU32(k_bfe_v, "v_bfe_u32 %0, %0, %1, 4")
VALU32(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1")
VALU32(k_lshl_or, "v_lshl_or_b32 %0, %0, 2, %1")
VALU32(k_add3, "v_add3_u32 %0, %0, %1, %2")
VALU32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
VALU32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 4")
VALU32(k_cndmask, "v_cndmask_b32_e64 %0, %0, %1, %3")
VALU32(k_mad24, "v_mad_u32_u24 %0, %0, %1, %2")
VALU32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
VALU32(k_perm, "v_perm_b32 %0, %0, %1, %2")
VALU32(k_lshlrev, "v_lshlrev_b32 %0, 3, %0")
VALU32(k_lshrrev_v, "v_lshrrev_b32 %0, %1, %0")
VALU32(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
VALU32(k_sdwa_add, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
VALU32(k_sdwa_lshl, "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2")
VALU32(k_dpp_row_shr, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")
VALU32(k_dpp_wave_shr, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf")
VALU32(k_dpp_add, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
VALU32(k_sad_u8, "v_sad_u8 %0, %0, %1, %2")
VALU32(k_med3, "v_med3_u32 %0, %0, %1, %2")
VALU64(k_lshl64, "v_lshlrev_b64 %0, 4, %0")
VALU64(k_lshl64_v, "v_lshlrev_b64 %0, %1, %0")
VALU64(k_lshr64_v, "v_lshrrev_b64 %0, %1, %0")
VCMP(k_cmp, "v_cmp_lt_u32_e64 %0, %1, %2")
__global__ __launch_bounds__(256) void k_readlane(uint32_t *out, int iters, uint32_t seed) {
  uint32_t s[CHAINS];
  uint32_t a = threadIdx.x * 2654435761u + seed;
  for (int c = 0; c < CHAINS; c++) s[c] = 0;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s[c]) : "v"(a));
    }
  }
  uint32_t r = 0;
  for (int c = 0; c < CHAINS; c++) r ^= s[c];
  if (r == 0x12345u) out[0] = r;
}
SALU64(k_s_and64, "s_and_b64 %0, %0, %1")
SALU64(k_s_lshl64, "s_lshl_b64 %0, %0, 1")

// a mix like the real kernels: 8 VALU + 4 SALU per group, do they overlap?
__global__ __launch_bounds__(256) void k_mix_valu_salu(uint32_t *out, int iters, uint32_t seed) {
  uint32_t a[CHAINS];
  unsigned long long s[4];
  for (int c = 0; c < CHAINS; c++) a[c] = threadIdx.x * 2654435761u + seed + c;
  unsigned long long k64 = 0x5555AAAA3333CCCCull ^ seed;
  for (int c = 0; c < 4; c++) s[c] = k64 * (c + 3);
  uint32_t k = seed | 1u;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < UNROLL; u++) {
#pragma unroll
      for (int c = 0; c < CHAINS; c++) {
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[c]) : "v"(k));
        if (c & 1) asm volatile("s_and_b64 %0, %0, %1" : "+s"(s[c >> 1]) : "s"(k64) : "scc");
      }
    }
  }
  uint32_t r = 0;
  for (int c = 0; c < CHAINS; c++) r ^= a[c];
  for (int c = 0; c < 4; c++) r ^= (uint32_t)s[c];
  if (r == 0x12345u) out[0] = r;
}

// ---- LDS: 8 operations per iteration with different immediate offsets, one wait per iteration
// MODE 0: lane * 4 (consecutive banks)  1: random word in 8192  2: all lanes one address  3: 8 lanes per address  4: lane*4 + 128*(lane&1) ...
template <int OP, int MODE>
__global__ __launch_bounds__(256) void k_lds(uint32_t *out, int iters, uint32_t seed) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[10240];
  for (int i = threadIdx.x; i < 10240; i += 256) lds[i] = i;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t h = (threadIdx.x * 2654435761u + seed);
  h ^= h >> 15; h *= 0x85ebca6bu; h ^= h >> 13;
  uint32_t addr;
  if (MODE == 0) addr = lane * 4 + wave * 256;
  else if (MODE == 1) addr = (h & 8191u) * 4;
  else if (MODE == 2) addr = wave * 64;
  else if (MODE == 3) addr = (lane >> 3) * 4 + wave * 256;
  else if (MODE == 4) addr = lane * 8 + wave * 512;           // 8-byte stride (for 64-bit ops: consecutive)
  else addr = ((h & 15u) * 8) + ((h >> 8) & 7u) * 1280;       // context-cell like: 16 cells of 8 bytes in one of 8 rows
  const uint32_t base = (uint32_t)reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void *)lds);
  addr += base;
  uint32_t one = 1, zero = 0;
  uint32_t acc = 0;
  for (int i = 0; i < iters; i++) {
    if (OP == 0) {  // ds_add_u32, no return
      asm volatile("ds_add_u32 %0, %1\n ds_add_u32 %0, %1 offset:4096\n ds_add_u32 %0, %1 offset:8192\n ds_add_u32 %0, %1 offset:2048\n"
                   "ds_add_u32 %0, %1 offset:6144\n ds_add_u32 %0, %1 offset:1024\n ds_add_u32 %0, %1 offset:5120\n ds_add_u32 %0, %1 offset:3072\n"
                   "s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(one) : "memory");
    } else if (OP == 1) {  // ds_add_u64
      asm volatile("ds_add_u64 %0, %1\n ds_add_u64 %0, %1 offset:4096\n ds_add_u64 %0, %1 offset:8192\n ds_add_u64 %0, %1 offset:2048\n"
                   "ds_add_u64 %0, %1 offset:6144\n ds_add_u64 %0, %1 offset:1024\n ds_add_u64 %0, %1 offset:5120\n ds_add_u64 %0, %1 offset:3072\n"
                   "s_waitcnt lgkmcnt(0)" :: "v"(addr), "v"((unsigned long long)one | ((unsigned long long)zero << 32)) : "memory");
    } else if (OP == 2) {  // ds_read_u8
      uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("ds_read_u8 %0, %8\n ds_read_u8 %1, %8 offset:4096\n ds_read_u8 %2, %8 offset:8192\n ds_read_u8 %3, %8 offset:2048\n"
                   "ds_read_u8 %4, %8 offset:6144\n ds_read_u8 %5, %8 offset:1024\n ds_read_u8 %6, %8 offset:5120\n ds_read_u8 %7, %8 offset:3072\n"
                   "s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
      acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    } else if (OP == 3) {  // ds_read_b32
      uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4096\n ds_read_b32 %2, %8 offset:8192\n ds_read_b32 %3, %8 offset:2048\n"
                   "ds_read_b32 %4, %8 offset:6144\n ds_read_b32 %5, %8 offset:1024\n ds_read_b32 %6, %8 offset:5120\n ds_read_b32 %7, %8 offset:3072\n"
                   "s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
      acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    } else if (OP == 4) {  // ds_add_rtn_u32
      uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("ds_add_rtn_u32 %0, %8, %9\n ds_add_rtn_u32 %1, %8, %9 offset:4096\n ds_add_rtn_u32 %2, %8, %9 offset:8192\n ds_add_rtn_u32 %3, %8, %9 offset:2048\n"
                   "ds_add_rtn_u32 %4, %8, %9 offset:6144\n ds_add_rtn_u32 %5, %8, %9 offset:1024\n ds_add_rtn_u32 %6, %8, %9 offset:5120\n ds_add_rtn_u32 %7, %8, %9 offset:3072\n"
                   "s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr), "v"(one) : "memory");
      acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    } else if (OP == 5) {  // ds_read_u16
      uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("ds_read_u16 %0, %8\n ds_read_u16 %1, %8 offset:4096\n ds_read_u16 %2, %8 offset:8192\n ds_read_u16 %3, %8 offset:2048\n"
                   "ds_read_u16 %4, %8 offset:6144\n ds_read_u16 %5, %8 offset:1024\n ds_read_u16 %6, %8 offset:5120\n ds_read_u16 %7, %8 offset:3072\n"
                   "s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr) : "memory");
      acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    } else {  // ds_bpermute_b32
      uint32_t r0, r1, r2, r3, r4, r5, r6, r7;
      asm volatile("ds_bpermute_b32 %0, %8, %9\n ds_bpermute_b32 %1, %8, %9\n ds_bpermute_b32 %2, %8, %9\n ds_bpermute_b32 %3, %8, %9\n"
                   "ds_bpermute_b32 %4, %8, %9\n ds_bpermute_b32 %5, %8, %9\n ds_bpermute_b32 %6, %8, %9\n ds_bpermute_b32 %7, %8, %9\n"
                   "s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3), "=v"(r4), "=v"(r5), "=v"(r6), "=v"(r7) : "v"(addr & 255u), "v"(one) : "memory");
      acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;
    }
  }
  __syncthreads();
  if (acc == 0x12345u || lds[threadIdx.x] == 0xFFFFFFF1u) out[0] = acc;
}

// ---- global loads: W bytes per lane, a wave reads 64 W consecutive bytes per load, LOADS loads in flight per lane
template <int W>
__global__ __launch_bounds__(256) void k_gload(const uint8_t *__restrict__ p, uint64_t bytes, uint32_t *out) {
  const uint64_t wave = ((uint64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * 256) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  constexpr int LOADS = 8;
  const uint64_t span = 64ull * W;  // bytes per wave-load
  uint32_t acc = 0;
  for (uint64_t o = wave * span * LOADS; o + span * LOADS <= bytes; o += nwaves * span * LOADS) {
#pragma unroll
    for (int l = 0; l < LOADS; l++) {
      const uint8_t *q = p + o + (uint64_t)l * span + (uint64_t)lane * W;
      if (W == 1) acc += *q;
      else if (W == 2) acc += *reinterpret_cast<const uint16_t *>(q);
      else if (W == 4) acc += *reinterpret_cast<const uint32_t *>(q);
      else { const uint4 v = *reinterpret_cast<const uint4 *>(q); acc += v.x ^ v.y ^ v.z ^ v.w; }
    }
  }
  if (acc == 0x12345u) out[0] = acc;
}

typedef void (*kern_t)(uint32_t *, int, uint32_t);
struct Row { const char *name; kern_t k; int ops_per_stmt; };

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int sel_lo = argc > 1 ? atoi(argv[1]) : 0, sel_hi = argc > 2 ? atoi(argv[2]) : 1 << 30;
  int test_no = 0;
  hipDeviceProp_t pr;
  CK(hipGetDeviceProperties(&pr, 0));
  const int cus = pr.multiProcessorCount;
  const double clk = pr.clockRate * 1e3;  // Hz
  printf("device %s, %d CUs, clock %.0f MHz\n", pr.name, cus, clk / 1e6);
  uint32_t *out;
  CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;
  const int grid = cus * 8;  // 8 blocks of 4 waves per CU = 8 waves per SIMD
  auto run = [&](const char *name, kern_t k, double stmts_per_iter) {
    const int me = test_no++;
    if (me < sel_lo || me >= sel_hi) return;
    printf("[%d] ", me);
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, iters, 12345u + r);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (t < best) best = t;
    }
    const double wave_instr = (double)grid * 4 * iters * stmts_per_iter;
    const double per_simd = wave_instr / (cus * 4.0);
    printf("%-28s %8.3f ms  %7.2f cycles per wave-instruction per SIMD (at %.0f MHz)  %6.2f per CU\n", name, best, best * 1e-3 * clk / per_simd, clk / 1e6,
           best * 1e-3 * clk / (wave_instr / cus));
  };
  const double S = (double)CHAINS * UNROLL;
#define R(K) run(#K, K, S)
  R(k_add); R(k_and); R(k_xor); R(k_bfe); R(k_bfe_v); R(k_lshl_add); R(k_lshl_or); R(k_add3); R(k_and_or); R(k_alignbit); R(k_cndmask);
  R(k_mad24); R(k_mul_lo); R(k_perm); R(k_lshlrev); R(k_lshrrev_v); R(k_pk_add_u16); R(k_sdwa_add); R(k_sdwa_lshl); R(k_dpp_row_shr);
  R(k_dpp_wave_shr); R(k_dpp_add); R(k_sad_u8); R(k_med3); R(k_lshl64); R(k_lshl64_v); R(k_lshr64_v); R(k_cmp); R(k_readlane);
  R(k_s_and64); R(k_s_lshl64);
  run("k_mix_valu_salu (8 VALU+4 SALU)", k_mix_valu_salu, S);  // per VALU instruction
#define RL(OP, MODE, NAME) run(NAME, k_lds<OP, MODE>, 8.0)
  RL(0, 0, "ds_add_u32 consecutive"); RL(0, 1, "ds_add_u32 random"); RL(0, 2, "ds_add_u32 one address"); RL(0, 3, "ds_add_u32 8 lanes/address");
  RL(0, 5, "ds_add_u32 ctx-like");
  RL(1, 4, "ds_add_u64 consecutive"); RL(1, 1, "ds_add_u64 random"); RL(1, 5, "ds_add_u64 ctx-like");
  RL(2, 0, "ds_read_u8 consecutive words"); RL(2, 1, "ds_read_u8 random");
  RL(5, 0, "ds_read_u16 consecutive words");
  RL(3, 0, "ds_read_b32 consecutive"); RL(3, 1, "ds_read_b32 random");
  RL(4, 0, "ds_add_rtn_u32 consecutive"); RL(4, 1, "ds_add_rtn_u32 random");
  RL(6, 0, "ds_bpermute_b32");
  // global loads
  const uint64_t bytes = 1ull << 31;
  uint8_t *buf = nullptr;
  if (sel_hi > test_no) {
    CK(hipMalloc(&buf, bytes));
    CK(hipMemset(buf, 1, bytes));
  }
  auto runl = [&](const char *name, void (*k)(const uint8_t *, uint64_t, uint32_t *), int w, uint64_t use) {
    const int me = test_no++;
    if (me < sel_lo || me >= sel_hi || !buf) return;
    printf("[%d] ", me);
    float best = 1e9;
    for (int r = 0; r < 3; r++) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(cus * 8), dim3(256), 0, 0, (const uint8_t *)buf, use, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (t < best) best = t;
    }
    const double instr = (double)use / (64.0 * w);
    printf("%-28s %8.3f ms  %8.1f GB/s  %6.2f cycles per wave-load per CU\n", name, best, use / (best * 1e-3) / 1e9, best * 1e-3 * clk / (instr / cus));
  };
  runl("global_load_ubyte", k_gload<1>, 1, bytes / 8);
  runl("global_load_ushort", k_gload<2>, 2, bytes / 4);
  runl("global_load_dword", k_gload<4>, 4, bytes / 2);
  runl("global_load_dwordx4", k_gload<16>, 16, bytes);
  // the same out of L2 / MALL (64 MB footprint read repeatedly is not what the kernels do; skip)
  return 0;
}
