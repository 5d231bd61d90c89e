// morton.hip — 63-bit Morton (Z-order) codes of 3D points on a regular grid (reference
// misc/morton_sort.py:13-99: 21 bits per axis, x in bit 0, y in bit 1, z in bit 2 of every triple; cell =
// clamp((p - lower) / inc, 0, size - 1) truncated to an integer).  One thread per point, HBM bound
// (12 B in, 8 B out); the codes feed ms_radix_sort_pairs (64-bit keys) for spatially coherent orderings.
#include "common.h"

namespace ms {

// spread the low 21 bits of x so that bit i lands at bit 3 i
__device__ __forceinline__ uint64_t spread21(uint64_t x) {
  x &= 0x1fffffull;
  x = (x | (x << 32)) & 0x1f00000000ffffull;
  x = (x | (x << 16)) & 0x1f0000ff0000ffull;
  x = (x | (x << 8)) & 0x100f00f00f00f00full;
  x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
  x = (x | (x << 2)) & 0x1249249249249249ull;
  return x;
}

__global__ void __launch_bounds__(256)
morton_codes64_kernel(const float* __restrict__ points, int64_t n, float lx, float ly, float lz, float ix, float iy, float iz,
                      unsigned size, uint64_t* __restrict__ codes) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float hi = (float)(size - 1);
  // clamp BEFORE the integer conversion (NaN -> 0 through fmaxf/fminf semantics)
  const float cx = fminf(fmaxf((points[i * 3 + 0] - lx) / ix, 0.0f), hi);
  const float cy = fminf(fmaxf((points[i * 3 + 1] - ly) / iy, 0.0f), hi);
  const float cz = fminf(fmaxf((points[i * 3 + 2] - lz) / iz, 0.0f), hi);
  codes[i] = spread21((uint64_t)(unsigned)cx) | (spread21((uint64_t)(unsigned)cy) << 1) |
             (spread21((uint64_t)(unsigned)cz) << 2);
}

}  // namespace ms

extern "C" int ms_morton_codes64(const float* points3, int64_t n, const float* lower3_host, const float* inc3_host,
                                 uint32_t size, uint64_t* out_codes, void* stream) {
  MS_CHECK_ARG(n >= 0 && lower3_host && inc3_host && size >= 1 && size <= (1u << 21), "bad arguments");
  MS_CHECK_ARG(inc3_host[0] > 0.0f && inc3_host[1] > 0.0f && inc3_host[2] > 0.0f, "cell size must be positive");
  if (n == 0) return 0;
  MS_CHECK_ARG(points3 && out_codes, "null pointer");
  ms::morton_codes64_kernel<<<(unsigned)ms::div_up(n, 256), 256, 0, (hipStream_t)stream>>>(
      points3, n, lower3_host[0], lower3_host[1], lower3_host[2], inc3_host[0], inc3_host[1], inc3_host[2], size, out_codes);
  MS_CHECK_LAUNCH();
  return 0;
}
