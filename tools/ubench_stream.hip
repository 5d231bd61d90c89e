// Measurement tool: what does this chip stream?  Read-only, write-only and copy over 1.2 GB buffers, 128-bit accesses,
// plain and nontemporal.  The ceiling the per-gaussian passes (sh_fwd, gaussian_bwd) are judged against.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_stream.hip -o tools/ubench_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float vec4 __attribute__((ext_vector_type(4)));

template <bool NT> __global__ void __launch_bounds__(256) k_read(const vec4* __restrict__ a, size_t n, float* out) {
  vec4 acc = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    acc += NT ? __builtin_nontemporal_load(a + i) : a[i];
  if (acc.x + acc.y + acc.z + acc.w == 123.456f) *out = 1.0f;
}
template <bool NT> __global__ void __launch_bounds__(256) k_write(vec4* __restrict__ a, size_t n) {
  const vec4 v = {1, 2, 3, 4};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    if (NT) __builtin_nontemporal_store(v, a + i); else a[i] = v;
  }
}
template <bool NT> __global__ void __launch_bounds__(256) k_copy(const vec4* __restrict__ a, vec4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const vec4 v = a[i];
    if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
  }
}
// one write-heavy mix like gaussian_bwd: read 1 part, write 2.4 parts
template <bool NT> __global__ void __launch_bounds__(256) k_mix(const vec4* __restrict__ a, vec4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const vec4 v = a[i];
    if (NT) { __builtin_nontemporal_store(v, b + 2 * i); __builtin_nontemporal_store(v, b + 2 * i + 1); }
    else { b[2 * i] = v; b[2 * i + 1] = v; }
  }
}

template <typename F> static float time_ms(F f) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(s);
  for (int i = 0; i < 20; ++i) f();
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  return ms / 20;
}

int main() {
  const size_t bytes = 1200ull << 20, n = bytes / 16;
  vec4 *a, *b; float* o;
  hipMalloc(&a, bytes); hipMalloc(&b, 2 * bytes); hipMalloc(&o, 4);
  hipMemset(a, 0, bytes); hipMemset(b, 0, 2 * bytes);
  for (int blocks : {2048, 8192, 65536}) {
    const dim3 g(blocks), t(256);
    float r0 = time_ms([&] { k_read<false><<<g, t>>>(a, n, o); }), r1 = time_ms([&] { k_read<true><<<g, t>>>(a, n, o); });
    float w0 = time_ms([&] { k_write<false><<<g, t>>>(a, n); }), w1 = time_ms([&] { k_write<true><<<g, t>>>(a, n); });
    float c0 = time_ms([&] { k_copy<false><<<g, t>>>(a, b, n); }), c1 = time_ms([&] { k_copy<true><<<g, t>>>(a, b, n); });
    float m0 = time_ms([&] { k_mix<false><<<g, t>>>(a, b, n); }), m1 = time_ms([&] { k_mix<true><<<g, t>>>(a, b, n); });
    const double gb = bytes / 1e9;
    printf("blocks %6d  read %.2f / nt %.2f TB/s   write %.2f / nt %.2f   copy %.2f / nt %.2f   read1+write2 %.2f / nt %.2f\n", blocks,
           gb / r0, gb / r1, gb / w0, gb / w1, 2 * gb / c0, 2 * gb / c1, 3 * gb / m0, 3 * gb / m1);
  }
  return 0;
}
