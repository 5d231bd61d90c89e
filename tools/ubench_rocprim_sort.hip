// Measurement tool (not part of the product): how fast does rocPRIM's radix_sort_pairs (onesweep on gfx950) sort the
// mapper's two workloads?  Yardstick for csrc/scan_sort.hip.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_rocprim_sort.hip -o /tmp/ubench_rocprim_sort
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdlib>
#include <vector>

static void run(size_t n, unsigned begin_bit, unsigned end_bit, unsigned key_mask, const char* what) {
  std::vector<unsigned> hk(n);
  std::vector<int> hv(n);
  unsigned s = 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; hk[i] = (s >> 3) & key_mask; hv[i] = (int)i; }
  unsigned *k0, *k1; int *v0, *v1;
  hipMalloc(&k0, n * 4); hipMalloc(&k1, n * 4); hipMalloc(&v0, n * 4); hipMalloc(&v1, n * 4);
  hipMemcpy(k0, hk.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(v0, hv.data(), n * 4, hipMemcpyHostToDevice);
  size_t tmp_bytes = 0;
  rocprim::radix_sort_pairs(nullptr, tmp_bytes, k0, k1, v0, v1, n, begin_bit, end_bit);
  void* tmp; hipMalloc(&tmp, tmp_bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, begin_bit, end_bit);
  hipEventRecord(a);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) rocprim::radix_sort_pairs(tmp, tmp_bytes, k0, k1, v0, v1, n, begin_bit, end_bit);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  printf("%s: n=%zu bits [%u,%u) tmp=%zu KB  %.1f us per sort\n", what, n, begin_bit, end_bit, tmp_bytes >> 10, ms * 1000.f / iters);
  hipFree(k0); hipFree(k1); hipFree(v0); hipFree(v1); hipFree(tmp);
}

int main() {
  run(6000000, 0, 32, 0xffffffffu >> 3, "depth sort");
  run(12760000, 0, 14, 0x3fffu, "tile sort");
  run(12760000, 0, 16, 0xffffu, "tile sort 16 bit");
  return 0;
}
