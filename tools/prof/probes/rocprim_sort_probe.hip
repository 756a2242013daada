// Reference point for the radix sort: rocPRIM's device radix sort on the same shape of problem as the coordinate sort's primary
// pass (n pairs of 64-bit key / 32-bit value, 32 live key bits).  Not part of the product; prints ms per sort.
// build: hipcc -O3 --offload-arch=gfx950 rocprim_sort_probe.hip -o rocprim_sort_probe
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 50000000ull;
  const int bits = argc > 2 ? atoi(argv[2]) : 32;
  std::vector<uint64_t> hk(n);
  std::vector<uint32_t> hv(n);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; hk[i] = s & ((1ull << bits) - 1); hv[i] = (uint32_t)i; }
  uint64_t *k0, *k1; uint32_t *v0, *v1;
  CK(hipMalloc(&k0, n * 8)); CK(hipMalloc(&k1, n * 8)); CK(hipMalloc(&v0, n * 4)); CK(hipMalloc(&v1, n * 4));
  CK(hipMemcpy(k0, hk.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice));
  size_t tmp_bytes = 0;
  CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, n, 0, bits));
  void *tmp; CK(hipMalloc(&tmp, tmp_bytes));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int it = 0; it < 4; it++) {
    CK(hipEventRecord(a));
    CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, 0, bits));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("rocprim radix_sort_pairs u64/u32 n=%zu bits=%d tmp=%zu MB: %.3f ms\n", n, bits, tmp_bytes >> 20, ms);
  }
  return 0;
}
