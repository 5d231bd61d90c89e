// common.h — shared host/device helpers for the gfx950 kernels (wave64 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mi355_splat.h"
#include "splat_math.h"

namespace ms {

// ---- error reporting ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define MS_CHECK_ARG(cond, msg)                                  \
  do {                                                           \
    if (!(cond)) {                                               \
      ms::set_error("%s: %s", __func__, msg);                    \
      return MS_ERR_BAD_ARG;                                     \
    }                                                            \
  } while (0)

#define MS_CHECK_LAUNCH()                                                         \
  do {                                                                            \
    hipError_t e__ = hipGetLastError();                                           \
    if (e__ != hipSuccess) {                                                      \
      ms::set_error("%s: kernel launch failed: %s", __func__, hipGetErrorString(e__)); \
      return (int)e__;                                                            \
    }                                                                             \
  } while (0)

#define MS_CHECK_HIP(expr)                                                        \
  do {                                                                            \
    hipError_t e__ = (expr);                                                      \
    if (e__ != hipSuccess) {                                                      \
      ms::set_error("%s: %s failed: %s", __func__, #expr, hipGetErrorString(e__)); \
      return (int)e__;                                                            \
    }                                                                             \
  } while (0)

static inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// ---- wave64 primitives -------------------------------------------------------------------------
constexpr int WAVE = 64;

// Splat rows (round 5): ONE 64-byte row per gaussian — [points7 (0..6) | depth (7) | colour (8..10) | unused] — from
// which the product raster kernels gather a splat with three 16-byte loads out of ONE 128-byte line instead of ten
// 4-byte loads out of two or three lines (28- and 12-byte rows straddle): config D at tile 16, same box, backward
// 1.331 -> 1.249 ms, forward 0.626 -> 0.596 ms (tools/rbench.py --rows; 48-byte rows: 1.266 / 0.604; the two arrays
// merely padded to 32- and 16-byte rows: 1.305 / 0.630 — it is the line count, not the load width).  Offered through
// the C-ABI (ms_splat_rows_pack, ms_raster_fwd_rows, ms_raster_bwd_moments_rows); the frame executor can fill such a
// table in its projection and SH kernels but does not by default: see frame.hip, frame_uses_rows.
constexpr int SPLAT_ROW = 16;
constexpr int SPLAT_ROW_COLOUR = 8;

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// DPP move: returns src shuffled by the DPP control, lanes without a source keep `old`.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf, bool BOUND_CTRL = false>
__device__ __forceinline__ float dpp_f32(float old, float src) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL,
                                                    ROW_MASK, BANK_MASK, BOUND_CTRL));
}

// Sum over the 64 lanes of a wave; the total is valid in lane 63 (other lanes hold partials).
// row_shr:1,2,4,8 build row (16-lane) totals in lane 15 of each row, row_bcast:15 / row_bcast:31
// carry them across rows — 6 VALU+DPP instructions per value, no LDS traffic.  This replaces the
// reference's 32-lane shfl_down tree + shared-memory atomics (taichi_lib/concurrent.py:11-23,69-86).
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_f32<0x111>(0.f, v);                    // row_shr:1
  v += dpp_f32<0x112>(0.f, v);                    // row_shr:2
  v += dpp_f32<0x114>(0.f, v);                    // row_shr:4
  v += dpp_f32<0x118>(0.f, v);                    // row_shr:8
  v += dpp_f32<0x142, 0xa>(0.f, v);               // row_bcast:15 -> rows 1 and 3
  v += dpp_f32<0x143, 0xc>(0.f, v);               // row_bcast:31 -> rows 2 and 3
  return v;
}

// f64 path (gradcheck-style tests only): plain xor butterfly through ds_bpermute.
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float atomic_add(float* p, float v) { return atomicAdd(p, v); }
__device__ __forceinline__ double atomic_add(double* p, double v) { return atomicAdd(p, v); }

// fire-and-forget float atomic add (no return value => global_atomic_add_f32 without glc)
__device__ __forceinline__ void atomic_add_noret(float* p, float v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void atomic_add_noret(double* p, double v) {
  __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Sum NV per-thread values over a 256-thread block and add the block totals to dst[0..NV) with ONE atomic per
// value per block (wave DPP reduce -> LDS -> first NV threads).  For whole-launch sums (camera gradients):
// combined with a grid-stride loop over a bounded number of blocks this keeps the same-address atomic count
// in the thousands instead of one per wave.
template <typename T, int NV>
__device__ __forceinline__ void block_sum_commit(const T (&v)[NV], T* __restrict__ dst, T* lds /* >= 4 * NV */) {
  const int lane = lane_id(), wave = threadIdx.x >> 6, waves = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    const T s = wave_sum_to_lane63(v[k]);
    if (lane == 63) lds[wave * NV + k] = s;
  }
  __syncthreads();
  if ((int)threadIdx.x < NV) {
    T t = T(0);
    for (int w = 0; w < waves; ++w) t += lds[w * NV + threadIdx.x];
    if (t != T(0)) atomic_add_noret(dst + threadIdx.x, t);
  }
}

// 32 bit sort key of the depth pre-sort: float bits (non-negative depths keep their order) or the 16 bit quantisation
// of make_sort_key (tile_mapper.py:55-61).  near_plane > 0 fuses ndc_depth (torch_lib/projection.py:120-123,
// renderer.py:67): evaluated in double from the depth's own precision, then rounded once to the float the key is made of.
template <typename T>
__device__ __forceinline__ uint32_t depth_sort_key(T depth, int depth16, double near_plane, double far_plane) {
  float d;
  if (near_plane > 0.0) {
    const double dd = (double)depth;
    d = (float)(1.0 - (1.0 / dd - 1.0 / far_plane) / (1.0 / near_plane - 1.0 / far_plane));
  } else {
    d = (float)depth;
  }
  return depth16 ? (uint32_t)(fminf(fmaxf(d, 0.0f), 1.0f) * 65535.0f) : __float_as_uint(d);
}

}  // namespace ms
