// How fast are random device-scope atomics as a function of the footprint of the array they hit?  (Sizing question behind a
// "seen twice" filter in front of the duplicate-marking hash tables.)  Not part of the product.
// build: hipcc -O3 --offload-arch=gfx950 random_atomic_probe.hip -o random_atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ uint64_t mix64(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
template <int MODE>  // 0: atomicOr on a bit, 1: atomicCAS on a word (insert), 2: plain load, 3: atomicMin whose result is not used, 4: plain store
__global__ __launch_bounds__(256) void k(uint32_t *a, uint64_t words_mask, uint64_t n, uint32_t *sink) {
  const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint64_t h = mix64(i * 0x9e3779b97f4a7c15ull + 12345);
  uint32_t r;
  if (MODE == 0) r = atomicOr(&a[(h >> 5) & words_mask], 1u << (h & 31));
  else if (MODE == 1) r = atomicCAS(&a[h & words_mask], 0xFFFFFFFFu, (uint32_t)i);
  else if (MODE == 2) r = a[h & words_mask];
  else if (MODE == 3) { atomicMin(&a[h & words_mask], (uint32_t)i); r = 0; }
  else { a[h & words_mask] = (uint32_t)i; r = 0; }
  if (r == 0x12345678u) *sink = r;
}
int main(int argc, char **argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 48000000ull;
  uint32_t *a, *sink;
  const uint64_t max_words = 1ull << 28;  // 1 GiB
  CK(hipMalloc(&a, max_words * 4)); CK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (uint64_t words = 1ull << 18; words <= max_words; words <<= 2) {
    float ms[5];
    for (int mode = 0; mode < 5; mode++) {
      float best = 1e9;
      for (int it = 0; it < 3; it++) {
        CK(hipMemset(a, (mode == 1 || mode == 3) ? 0xFF : 0, words * 4));
        CK(hipEventRecord(e0));
        const unsigned grid = (unsigned)((n + 255) / 256);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, a, words - 1, n, sink);
        else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, a, words - 1, n, sink);
        else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, a, words - 1, n, sink);
        else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, a, words - 1, n, sink);
        else hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, a, words - 1, n, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (t < best) best = t;
      }
      ms[mode] = best;
    }
    printf("footprint %7.1f MB: %llu random accesses: atomicOr %.3f ms, atomicCAS %.3f ms, load %.3f ms, atomicMin (no result) %.3f ms, store %.3f ms\n", words * 4 / 1048576.0, (unsigned long long)n, ms[0], ms[1], ms[2], ms[3], ms[4]);
  }
  return 0;
}
