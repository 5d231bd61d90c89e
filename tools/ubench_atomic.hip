// Microbenchmark for a bucket-by-tile mapper: throughput of device-scope integer atomics (returning and not) on
// T counters from K threads in gaussian order, and of the scattered 8-byte bucket writes that would follow.
// hipcc --offload-arch=gfx950 -O3 tools/ubench_atomic.hip -o tools/ubench_atomic.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ unsigned hash_u32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// overlap i belongs to gaussian i / 2 (K/N = 2.1 on config D); a gaussian sits on a random tile and its second overlap is the tile to the right
__device__ __forceinline__ int tile_of(long i, int tiles) { return (int)((hash_u32((unsigned)(i >> 1)) + (unsigned)(i & 1)) % (unsigned)tiles); }

template <int MODE> __global__ void __launch_bounds__(256) k(int* __restrict__ counters, int* __restrict__ slots, uint2* __restrict__ buckets, long n, int tiles, int cap) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int t = tile_of(i, tiles);
  if (MODE == 0) { __hip_atomic_fetch_add(&counters[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }                 // histogram only
  if (MODE == 1) { slots[i] = __hip_atomic_fetch_add(&counters[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }      // claim a slot
  if (MODE == 2) { const int s = __hip_atomic_fetch_add(&counters[t], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                   if (s < cap) buckets[(long)t * cap + s] = make_uint2((unsigned)i, (unsigned)s); }                       // claim + scattered 8 B write
}

template <int MODE> void run(const char* name, int* counters, int* slots, uint2* buckets, long n, int tiles, int cap) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipMemsetAsync(counters, 0, tiles * sizeof(int));
    hipEventRecord(e0);
    k<MODE><<<(unsigned)((n + 255) / 256), 256>>>(counters, slots, buckets, n, tiles, cap);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("%-44s tiles %6d  %8.3f ms  %7.2f G atomics/s\n", name, tiles, best, n / best * 1e-6);
}

int main() {
  const long n = 12760302; const int cap = 2048;
  int *counters, *slots; uint2* buckets;
  hipMalloc(&counters, 65536 * sizeof(int)); hipMalloc(&slots, n * sizeof(int)); hipMalloc(&buckets, (size_t)16384 * cap * sizeof(uint2));
  for (int tiles : {16384, 4096, 65536}) {
    run<0>("atomic add, no return", counters, slots, buckets, n, tiles, cap);
    run<1>("atomic add, returning, slot stored in order", counters, slots, buckets, n, tiles, cap);
  }
  run<2>("returning + 8 B write into the tile bucket", counters, slots, buckets, n, 16384, cap);
  return 0;
}
