// unaligned_load_probe — what a streaming read of 16 bytes per lane reaches on gfx950 by ALIGNMENT and by the lane -> address pattern.
// (Round 3: adapt_score, bqsr_count and bqsr_apply all read the QUAL column with global_load_dwordx4 at addresses r * 150 + 16 j - two-byte
// aligned at best - and all three sit at 3.7 - 4.0 TB/s while a device-to-device copy reaches 5.5.)
//   hipcc -O3 --offload-arch=gfx950 unaligned_load_probe.hip -o unaligned_load_probe && ./unaligned_load_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// mode 0: lane-contiguous 16-byte chunks, every address shifted by `shift` bytes
// mode 1: reads of `len` bytes, 64 / bpr reads per wave, lane l = block l % bpr of read l / bpr (the one-length kernels' pattern)
// mode 2: lane-contiguous ALIGNED chunks covering the same span as mode 1 (what a realigned variant would load)
template <int DEPTH>
__global__ __launch_bounds__(256) void k_read(const uint8_t *__restrict__ p, uint64_t bytes, int mode, uint32_t shift, uint32_t len, uint32_t *out) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (uint64_t)gridDim.x * 4;
  const uint32_t bpr = (len + 15u) >> 4, rpw = 64u / bpr, slot = lane / bpr, jb = lane - slot * bpr;
  const uint64_t span = mode == 1 ? (uint64_t)rpw * len : 1024u;
  const uint32_t off = mode == 1 ? (slot < rpw ? slot * len + 16u * jb : 0u) : 16u * lane + shift;
  uint32_t acc = 0;
  const uint64_t trips = (bytes - 4096) / span;
  for (uint64_t t = wave; t + (DEPTH - 1) * waves < trips; t += DEPTH * waves) {
    uint4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      uint64_t base = (t + d * waves) * span;
      if (mode == 2) base &= ~(uint64_t)15;
      __builtin_memcpy(&v[d], p + base + off, 16);
    }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const uint64_t bytes = 6ull << 30;
  uint8_t *p;
  uint32_t *out;
  CK(hipMalloc(&p, bytes));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(p, 1, bytes));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  struct T { const char *name; int mode; uint32_t shift, len; } tests[] = {
      {"contiguous, aligned", 0, 0, 0},       {"contiguous, +1 byte", 0, 1, 0},      {"contiguous, +2 bytes", 0, 2, 0},
      {"contiguous, +4 bytes", 0, 4, 0},      {"contiguous, +8 bytes", 0, 8, 0},     {"reads of 150 bytes", 1, 0, 150},
      {"reads of 151 bytes", 1, 0, 151},      {"reads of 152 bytes", 1, 0, 152},     {"reads of 160 bytes", 1, 0, 160},
      {"reads of 144 bytes (9 blocks)", 1, 0, 144}, {"aligned chunks over spans of 6 x 150", 2, 0, 150}};
  for (int waves_per_simd : {4, 8})
    for (auto &t : tests) {
      const int grid = 256 * waves_per_simd;  // 4 waves per workgroup
      float best = 1e9f;
      for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(256), 0, 0, p, bytes, t.mode, t.shift, t.len ? t.len : 16u, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
      }
      // bytes actually requested by the lanes
      const uint32_t bpr = ((t.len ? t.len : 16u) + 15u) >> 4, rpw = 64u / bpr;
      const double frac = t.mode == 1 ? (double)(rpw * bpr * 16) / (double)(rpw * t.len) : 1.0;
      printf("%d waves/SIMD  %-42s %8.3f ms  %6.2f TB/s of the span (lanes request %.2fx the span)\n", waves_per_simd, t.name, best, (double)(bytes - 4096) / best / 1e9, frac);
      fflush(stdout);
    }
  return 0;
}
