// mapper.hip — OBB-vs-tile overlap test of the tile mapper: per-gaussian overlap count and
// (sort key, point index) emission.  Replaces tile_overlaps_kernel / generate_sort_keys_kernel
// (mapper/tile_mapper.py:76-86,115-146) on top of grid_query.py:10-91 (see splat_math.h).
//
// The overlap decisions are float comparisons that must agree between the count and the emit
// pass (and with the CPU oracle), so this file is compiled with FP contraction off: no FMA
// formation, every product and sum rounded individually, exactly like the oracle's numpy float32.
#pragma clang fp contract(off)
#include "common.h"
#include "frame_internal.h"

namespace ms {

__device__ __forceinline__ void load_point7(const float* __restrict__ points, int64_t i, float g[7]) {
#pragma unroll
  for (int k = 0; k < 7; ++k) g[k] = points[i * 7 + k];
}


// ---- a wave's gaussians have very different tile spans (a heavy-tailed scene: a few splats cover thousands of tiles
// next to splats that cover two) and a thread that walks 2 500 tiles keeps its 63 neighbours waiting: 3 x the cost per
// overlap of config D for both passes (tools/sweep_scenes.py, round 5).  Spans far above the wave's average are
// therefore walked by the WHOLE wave, one gaussian at a time, a lane per tile; everything else stays a thread per
// gaussian (when all spans are large the lanes are busy anyway, and the cooperative walk would only add its set-up).
// The tests are the same obb_test_tile() calls either way, so count and emit agree tile for tile.
__device__ __forceinline__ float readlane_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ ObbQuery wave_broadcast(const ObbQuery& q, int lane) {
  ObbQuery r;
  r.inv00 = readlane_f(q.inv00, lane); r.inv01 = readlane_f(q.inv01, lane);
  r.inv10 = readlane_f(q.inv10, lane); r.inv11 = readlane_f(q.inv11, lane);
  r.rel_min_x = readlane_f(q.rel_min_x, lane); r.rel_min_y = readlane_f(q.rel_min_y, lane);
  r.min_tile_x = __builtin_amdgcn_readlane(q.min_tile_x, lane); r.min_tile_y = __builtin_amdgcn_readlane(q.min_tile_y, lane);
  r.span_x = __builtin_amdgcn_readlane(q.span_x, lane); r.span_y = __builtin_amdgcn_readlane(q.span_y, lane);
  return r;
}
// lanes whose span the wave takes over: above 16 tiles AND above twice the wave's mean span (all 64 lanes must call)
__device__ __forceinline__ unsigned long long wide_span_lanes(const ObbQuery& q, bool valid, int* my_span) {
  const int span = valid && q.span_x > 0 && q.span_y > 0 ? q.span_x * q.span_y : 0;
  *my_span = span;
  // (spans are small integers: their sum is exact in float)
  const int total = (int)readlane_f(wave_sum_to_lane63((float)span), 63);
  int thr = total / 32;                                    // 2 x mean over 64 lanes
  thr = thr > 16 ? thr : 16;
  return __ballot(span > thr);
}
// the cooperative walk of ONE gaussian (query b, wave-uniform): on_hits(hit, tile_x, tile_y) is called by all lanes
// once per group of 64 tiles (row-major over the span)
template <typename F>
__device__ __forceinline__ void wave_walk_span(const ObbQuery& b, int tile_size, int row_begin, int row_end, F on_hits) {
  const int n = b.span_x * b.span_y, lane = lane_id();
  for (int base = 0; base < n; base += 64) {
    const int idx = base + lane;
    bool hit = false;
    int tu = 0, tv = 0;
    if (idx < n) {
      tv = idx / b.span_x; tu = idx - tv * b.span_x;
      const int ty = b.min_tile_y + tv;
      hit = ty >= row_begin && ty < row_end && obb_test_tile(b, tu, tv, tile_size);
    }
    on_hits(hit, b.min_tile_x + tu, b.min_tile_y + tv);
  }
}
__device__ __forceinline__ int count_span(const ObbQuery& q, bool valid, int tile_size, int row_begin, int row_end) {
  int span;
  unsigned long long wide = wide_span_lanes(q, valid, &span);
  const bool mine_is_wide = (wide >> lane_id()) & 1ull;
  int count = 0;
  if (span > 0 && !mine_is_wide) {
    for (int tv = 0; tv < q.span_y; ++tv) {
      const int ty = q.min_tile_y + tv;
      if (ty < row_begin || ty >= row_end) continue;
      for (int tu = 0; tu < q.span_x; ++tu)
        if (obb_test_tile(q, tu, tv, tile_size)) ++count;
    }
  }
  while (wide != 0) {
    const int src = __builtin_ctzll(wide);
    wide &= wide - 1;
    const ObbQuery b = wave_broadcast(q, src);
    int total = 0;
    wave_walk_span(b, tile_size, row_begin, row_end, [&](bool hit, int, int) { total += __builtin_popcountll(__ballot(hit)); });
    if (lane_id() == src) count = total;
  }
  return count;
}
// emit(position, tile_x, tile_y, source lane or -1): the thread-per-gaussian walk calls it for its own gaussian (source
// -1: the caller's own registers), the cooperative walk for gaussian `source` of the wave
template <typename F>
__device__ __forceinline__ void emit_span(const ObbQuery& q, bool valid, int64_t first, int tile_size, int row_begin,
                                          int row_end, F emit) {
  int span;
  unsigned long long wide = wide_span_lanes(q, valid, &span);
  const bool mine_is_wide = (wide >> lane_id()) & 1ull;
  if (span > 0 && !mine_is_wide) {
    int64_t o = first;
    // same (x outer, y inner) order as ti.grouped(ti.ndrange(span.x, span.y)); the order within one gaussian is
    // irrelevant after the sort (all its tiles differ)
    for (int tu = 0; tu < q.span_x; ++tu) {
      for (int tv = 0; tv < q.span_y; ++tv) {
        const int ty = q.min_tile_y + tv;
        if (ty < row_begin || ty >= row_end) continue;
        if (obb_test_tile(q, tu, tv, tile_size)) emit(o++, q.min_tile_x + tu, ty, -1);
      }
    }
  }
  while (wide != 0) {
    const int src = __builtin_ctzll(wide);
    wide &= wide - 1;
    const ObbQuery b = wave_broadcast(q, src);
    const int lo = __builtin_amdgcn_readlane((int)(uint32_t)first, src), hi = __builtin_amdgcn_readlane((int)(uint32_t)(first >> 32), src);
    int64_t o = (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
    wave_walk_span(b, tile_size, row_begin, row_end, [&](bool hit, int tx, int ty) {
      const unsigned long long m = __ballot(hit);
      if (hit) emit(o + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)), tx, ty, src);
      o += __builtin_popcountll(m);
    });
  }
}

__global__ void __launch_bounds__(256)
tile_count_kernel(const float* __restrict__ points, const int32_t* __restrict__ order,
                  const uint32_t* __restrict__ cull_keys, int64_t v, int image_w,
                  int image_h, int tile_size, float alpha_threshold, int row_begin, int row_end,
                  int32_t* __restrict__ counts, float* __restrict__ ordered_points) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // frame executor: culled gaussians stay in place and sort last (CULLED_DEPTH_KEY); they overlap nothing and their
  // slot of the ordered copy is never read (the emit pass skips rows without overlaps)
  const bool valid = i < v && !(cull_keys && cull_keys[i] == CULLED_DEPTH_KEY);
  float g[7] = {0.f, 0.f, 1.f, 0.f, 1.f, 1.f, 0.f};
  if (valid) {
    load_point7(points, order ? (int64_t)order[i] : i, g);
    if (ordered_points) {
#pragma unroll
      for (int k = 0; k < 7; ++k) ordered_points[i * 7 + k] = g[k];   // gathered once, re-read linearly by the emit
    }
  }
  const ObbQuery q = obb_grid_query(g, image_w, image_h, tile_size, alpha_threshold);
  const int count = count_span(q, valid, tile_size, row_begin, row_end);
  if (i < v) counts[i] = count;
}

// MODE 0: key = tile_id << 32 | float_bits(depth)   (tile_mapper.py:36-42)
// MODE 1: key = tile_id << 16 | u16(clamp(depth, 0, 1) * 65535)   (tile_mapper.py:55-61)
// MODE 2: key = tile_id only — gaussians are visited in depth order (`order`), so a STABLE sort by
//         tile alone reproduces the (tile, depth, point) order with a third of the radix passes
template <typename KeyT, int MODE>
__global__ void __launch_bounds__(256)
tile_emit_kernel(const float* __restrict__ points, const float* __restrict__ depth,
                 const int32_t* __restrict__ order, const int32_t* __restrict__ cum, int64_t v,
                 int image_w, int image_h, int tile_size, float alpha_threshold, int row_begin,
                 int row_end, int points_ordered, const int32_t* __restrict__ k_limit,
                 KeyT* __restrict__ keys, int32_t* __restrict__ values) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // frame executor: *k_limit == 0 <=> the overlap total exceeds the capacity of keys / values: write nothing
  if (k_limit && *k_limit == 0) return;                // (uniform over the launch)
  // culled gaussians of the frame executor (no entry in the ordered copy): zero overlaps, nothing to emit
  const bool valid = i < v && !(k_limit && cum[i + 1] == cum[i]);
  const int64_t src = valid ? (order ? (int64_t)order[i] : i) : 0;
  float g[7] = {0.f, 0.f, 1.f, 0.f, 1.f, 1.f, 0.f};
  if (valid) load_point7(points, points_ordered ? i : src, g);
  const ObbQuery q = obb_grid_query(g, image_w, image_h, tile_size, alpha_threshold);
  const int tiles_wide = image_w / tile_size;
  KeyT depth_key = 0;
  if (valid) {
    if (MODE == 1) {
      const float c = fminf(fmaxf(depth[src], 0.0f), 1.0f);
      depth_key = (KeyT)(uint32_t)(c * 65535.0f);
    } else if (MODE == 0) {
      depth_key = (KeyT)__float_as_uint(depth[src]);   // non-negative float bits keep their order
    }
  }
  const uint32_t dk_lo = (uint32_t)depth_key, dk_hi = (uint32_t)((uint64_t)depth_key >> 32);
  emit_span(q, valid, valid ? (int64_t)cum[i] : 0, tile_size, row_begin, row_end, [&](int64_t o, int tx, int ty, int from) {
    const int64_t tile_id = (int64_t)tx + (int64_t)ty * tiles_wide;
    KeyT dk = depth_key;
    int32_t who = (int32_t)src;
    if (from >= 0) {            // cooperative walk: the key parts and the index of gaussian `from` of the wave
      dk = (KeyT)(((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)dk_hi, from) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)dk_lo, from));
      who = __builtin_amdgcn_readlane((int32_t)src, from);
    }
    if constexpr (MODE == 2) keys[o] = (KeyT)tile_id;
    else if constexpr (MODE == 1) keys[o] = (KeyT)(dk | ((KeyT)tile_id << 16));
    else keys[o] = (KeyT)(dk | ((KeyT)tile_id << 32));
    values[o] = who;
  });
}

// ---- frame executor, direct order (no depth pre-sort): the gaussians are visited in storage order, the pairs are
// sorted by tile with a stable radix sort and each tile's run is then depth-sorted on its own (tile_sort.hip).  The
// result is the same (tile, depth key, point index) order as MODE 0 / MODE 2 above.
template <typename T>
__global__ void __launch_bounds__(256, 8)      // 64 VGPRs: eight waves per SIMD (the cooperative walk had taken it to 66)
tile_count_direct_kernel(const float* __restrict__ points, const T* __restrict__ cull_depth, int64_t v, int image_w,
                         int image_h, int tile_size, float alpha_threshold, int row_begin, int row_end,
                         int32_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = i < v && !(cull_depth && !(cull_depth[i] > T(0)));      // same rule as DepthPairs::key
  float g[7] = {0.f, 0.f, 1.f, 0.f, 1.f, 1.f, 0.f};
  if (valid) load_point7(points, i, g);
  const ObbQuery q = obb_grid_query(g, image_w, image_h, tile_size, alpha_threshold);
  const int count = count_span(q, valid, tile_size, row_begin, row_end);
  if (i < v) counts[i] = count;
}

// key = tile_id << 32 | depth_sort_key(depth) (the 32 bit key of the depth pre-sort), value = point index
template <typename T>
__global__ void __launch_bounds__(256)
tile_emit_direct_kernel(const float* __restrict__ points, const T* __restrict__ depth, const int32_t* __restrict__ cum,
                        int64_t v, int image_w, int image_h, int tile_size, float alpha_threshold, int row_begin,
                        int row_end, int depth16, double near_plane, double far_plane,
                        const int32_t* __restrict__ k_limit, uint64_t* __restrict__ keys, int32_t* __restrict__ values) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k_limit && *k_limit == 0) return;                // overflow of the caller's capacity: write nothing (uniform)
  const int64_t first = i < v ? (int64_t)cum[i] : 0;
  const bool valid = i < v && cum[i + 1] != first;     // not culled, and overlaps a tile of this strip
  float g[7] = {0.f, 0.f, 1.f, 0.f, 1.f, 1.f, 0.f};
  if (valid) load_point7(points, i, g);
  const ObbQuery q = obb_grid_query(g, image_w, image_h, tile_size, alpha_threshold);
  const int tiles_wide = image_w / tile_size;
  const uint32_t depth_key = valid ? (uint32_t)depth_sort_key(depth[i], depth16, near_plane, far_plane) : 0u;
  emit_span(q, valid, first, tile_size, row_begin, row_end, [&](int64_t o, int tx, int ty, int from) {
    const uint64_t tile_id = (uint64_t)((int64_t)tx + (int64_t)ty * tiles_wide);
    uint32_t dk = depth_key;
    int32_t who = (int32_t)i;
    if (from >= 0) { dk = (uint32_t)__builtin_amdgcn_readlane((int)depth_key, from); who = __builtin_amdgcn_readlane((int32_t)i, from); }
    keys[o] = (tile_id << 32) | dk;
    values[o] = who;
  });
}

// 32 bit sort keys of the depth pre-sort: float bits (non-negative depths) or the 16 bit quantisation.
// near_plane > 0 fuses ndc_depth (torch_lib/projection.py:120-123, renderer.py:67): evaluated in double
// from the depth's own precision, then rounded once to the float the key is made of.
template <typename T>
__global__ void __launch_bounds__(256)
depth_keys_kernel(const T* __restrict__ depth, int64_t v, int depth16, double near_plane, double far_plane,
                  uint32_t* __restrict__ keys, int32_t* __restrict__ values) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= v) return;
  keys[i] = depth_sort_key(depth[i], depth16, near_plane, far_plane);
  values[i] = (int32_t)i;
}

void tile_count_launch(const float* points7, const int32_t* order, const uint32_t* cull_keys, int64_t v, int image_w,
                       int image_h, int tile_size, float alpha_threshold, int row_begin, int row_end,
                       int32_t* out_counts, float* out_ordered, hipStream_t s) {
  tile_count_kernel<<<dim3((unsigned)div_up(v, 256)), dim3(256), 0, s>>>(
      points7, order, cull_keys, v, image_w, image_h, tile_size, alpha_threshold, row_begin, row_end, out_counts, out_ordered);
}

void tile_emit_ordered_launch(const float* ordered_points7, const int32_t* order, const int32_t* cum, int64_t v,
                              int image_w, int image_h, int tile_size, float alpha_threshold, int row_begin,
                              int row_end, const int32_t* k_limit_dev, uint32_t* out_keys, int32_t* out_values,
                              hipStream_t s) {
  tile_emit_kernel<uint32_t, 2><<<dim3((unsigned)div_up(v, 256)), dim3(256), 0, s>>>(
      ordered_points7, nullptr, order, cum, v, image_w, image_h, tile_size, alpha_threshold, row_begin, row_end, 1,
      k_limit_dev, out_keys, out_values);
}

void tile_count_direct_launch(const float* points7, const void* cull_depth, int dtype, int64_t v, int image_w, int image_h,
                              int tile_size, float alpha_threshold, int row_begin, int row_end, int32_t* out_counts,
                              hipStream_t s) {
  const dim3 grid((unsigned)div_up(v, 256)), block(256);
  if (dtype == MS_F64)
    tile_count_direct_kernel<double><<<grid, block, 0, s>>>(points7, (const double*)cull_depth, v, image_w, image_h, tile_size,
                                                            alpha_threshold, row_begin, row_end, out_counts);
  else
    tile_count_direct_kernel<float><<<grid, block, 0, s>>>(points7, (const float*)cull_depth, v, image_w, image_h, tile_size,
                                                           alpha_threshold, row_begin, row_end, out_counts);
}

void tile_emit_direct_launch(const float* points7, const void* depth, int dtype, const int32_t* cum, int64_t v, int image_w,
                             int image_h, int tile_size, float alpha_threshold, int row_begin, int row_end, int depth16,
                             double ndc_near, double ndc_far, const int32_t* k_limit_dev, uint64_t* out_keys,
                             int32_t* out_values, hipStream_t s) {
  const dim3 grid((unsigned)div_up(v, 256)), block(256);
  if (dtype == MS_F64)
    tile_emit_direct_kernel<double><<<grid, block, 0, s>>>(points7, (const double*)depth, cum, v, image_w, image_h, tile_size,
                                                           alpha_threshold, row_begin, row_end, depth16, ndc_near, ndc_far,
                                                           k_limit_dev, out_keys, out_values);
  else
    tile_emit_direct_kernel<float><<<grid, block, 0, s>>>(points7, (const float*)depth, cum, v, image_w, image_h, tile_size,
                                                          alpha_threshold, row_begin, row_end, depth16, ndc_near, ndc_far,
                                                          k_limit_dev, out_keys, out_values);
}

}  // namespace ms

using namespace ms;

extern "C" int ms_depth_sort_keys(const void* depth, int64_t v, int depth16, double ndc_near, double ndc_far,
                                  uint32_t* out_keys, int32_t* out_values, int dtype, void* stream) {
  MS_CHECK_ARG(v >= 0, "v < 0");
  MS_CHECK_ARG(dtype == MS_F32 || dtype == MS_F64, "dtype must be MS_F32 or MS_F64");
  if (v == 0) return 0;
  MS_CHECK_ARG(depth && out_keys && out_values, "null pointer");
  const dim3 grid((unsigned)div_up(v, 256)), block(256);
  if (dtype == MS_F32)
    depth_keys_kernel<float><<<grid, block, 0, (hipStream_t)stream>>>((const float*)depth, v, depth16, ndc_near, ndc_far, out_keys, out_values);
  else
    depth_keys_kernel<double><<<grid, block, 0, (hipStream_t)stream>>>((const double*)depth, v, depth16, ndc_near, ndc_far, out_keys, out_values);
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_tile_count(const float* points7, const int32_t* order, int64_t v, int image_w, int image_h, int tile_size,
                             float alpha_threshold, int tile_row_begin, int tile_row_end,
                             int32_t* out_counts, float* out_ordered_points7, void* stream) {
  MS_CHECK_ARG(v >= 0, "v < 0");
  MS_CHECK_ARG(tile_size > 0 && image_w > 0 && image_h > 0, "bad image/tile size");
  MS_CHECK_ARG(image_w % tile_size == 0 && image_h % tile_size == 0, "image size must be padded to the tile size");
  if (v == 0) return 0;
  MS_CHECK_ARG(points7 && out_counts, "null pointer");
  tile_count_kernel<<<dim3((unsigned)div_up(v, 256)), dim3(256), 0, (hipStream_t)stream>>>(
      points7, order, nullptr, v, image_w, image_h, tile_size, alpha_threshold, tile_row_begin, tile_row_end, out_counts, out_ordered_points7);
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_tile_emit(const float* points7, const float* depth, const int32_t* order, const int32_t* cum,
                            int64_t v, int image_w, int image_h, int tile_size, float alpha_threshold,
                            int tile_row_begin, int tile_row_end, int key_mode, int points_are_ordered,
                            void* out_keys, int32_t* out_values, void* stream) {
  MS_CHECK_ARG(v >= 0, "v < 0");
  MS_CHECK_ARG(key_mode >= 0 && key_mode <= 2, "key_mode must be 0 (tile|depth32), 1 (tile|depth16) or 2 (tile only)");
  MS_CHECK_ARG(tile_size > 0 && image_w > 0 && image_h > 0, "bad image/tile size");
  MS_CHECK_ARG(image_w % tile_size == 0 && image_h % tile_size == 0, "image size must be padded to the tile size");
  if (key_mode == 1) {
    const int64_t tiles = (int64_t)(image_w / tile_size) * (image_h / tile_size);
    MS_CHECK_ARG(tiles <= 65536, "use_depth16 keys hold a 16 bit tile id: too many tiles");
  }
  if (v == 0) return 0;
  MS_CHECK_ARG(points7 && cum && out_keys && out_values, "null pointer");
  MS_CHECK_ARG(key_mode == 2 || depth, "depth is null");
  const dim3 block(256), grid((unsigned)div_up(v, 256));
  hipStream_t s = (hipStream_t)stream;
#define MS_EMIT(KeyT, MODE) tile_emit_kernel<KeyT, MODE><<<grid, block, 0, s>>>(points7, depth, order, cum, v, image_w, image_h, tile_size, alpha_threshold, tile_row_begin, tile_row_end, points_are_ordered, nullptr, (KeyT*)out_keys, out_values)
  if (key_mode == 0) MS_EMIT(uint64_t, 0);
  else if (key_mode == 1) MS_EMIT(uint32_t, 1);
  else MS_EMIT(uint32_t, 2);
#undef MS_EMIT
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_tile_emit_keys64(const float* points7, const void* depth, int depth_dtype, const int32_t* cum, int64_t v,
                                   int image_w, int image_h, int tile_size, float alpha_threshold, int tile_row_begin,
                                   int tile_row_end, int depth16, double ndc_near, double ndc_far, uint64_t* out_keys,
                                   int32_t* out_values, void* stream) {
  MS_CHECK_ARG(v >= 0, "v < 0");
  MS_CHECK_ARG(depth_dtype == MS_F32 || depth_dtype == MS_F64, "depth_dtype must be MS_F32 or MS_F64");
  MS_CHECK_ARG(tile_size > 0 && image_w > 0 && image_h > 0, "bad image/tile size");
  MS_CHECK_ARG(image_w % tile_size == 0 && image_h % tile_size == 0, "image size must be padded to the tile size");
  if (depth16) {
    const int64_t tiles = (int64_t)(image_w / tile_size) * (image_h / tile_size);
    MS_CHECK_ARG(tiles <= 65536, "use_depth16 keys hold a 16 bit tile id: too many tiles");   // tile_mapper.py:49-53
  }
  if (v == 0) return 0;
  MS_CHECK_ARG(points7 && depth && cum && out_keys && out_values, "null pointer");
  tile_emit_direct_launch(points7, depth, depth_dtype, cum, v, image_w, image_h, tile_size, alpha_threshold, tile_row_begin,
                          tile_row_end, depth16, ndc_near > 0.0 ? ndc_near : 0.0, ndc_far, nullptr, out_keys, out_values,
                          (hipStream_t)stream);
  MS_CHECK_LAUNCH();
  return 0;
}
