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

__global__ void __launch_bounds__(256)
tile_count_kernel(const float* __restrict__ points, const int32_t* __restrict__ order,
                  const uint32_t* __restrict__ cull_keys, int64_t v, int image_w,
                  int image_h, int tile_size, float alpha_threshold, int row_begin, int row_end,
                  int32_t* __restrict__ counts, float* __restrict__ ordered_points) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= v) return;
  // frame executor: culled gaussians stay in place and sort last (CULLED_DEPTH_KEY); they overlap nothing and their
  // slot of the ordered copy is never read (the emit pass skips rows without overlaps)
  if (cull_keys && cull_keys[i] == CULLED_DEPTH_KEY) { counts[i] = 0; return; }
  float g[7];
  load_point7(points, order ? (int64_t)order[i] : i, g);
  if (ordered_points) {
#pragma unroll
    for (int k = 0; k < 7; ++k) ordered_points[i * 7 + k] = g[k];   // gathered once, re-read linearly by the emit
  }
  const ObbQuery q = obb_grid_query(g, image_w, image_h, tile_size, alpha_threshold);
  int count = 0;
  for (int tv = 0; tv < q.span_y; ++tv) {
    const int ty = q.min_tile_y + tv;
    if (ty < row_begin || ty >= row_end) continue;
    for (int tu = 0; tu < q.span_x; ++tu)
      if (obb_test_tile(q, tu, tv, tile_size)) ++count;
  }
  counts[i] = count;
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
  if (i >= v) return;
  // frame executor: *k_limit == 0 <=> the overlap total exceeds the capacity of keys / values: write nothing
  if (k_limit && *k_limit == 0) return;
  // culled gaussians of the frame executor (no entry in the ordered copy): zero overlaps, nothing to emit
  if (k_limit && cum[i + 1] == cum[i]) return;
  const int64_t src = order ? (int64_t)order[i] : i;
  float g[7];
  load_point7(points, points_ordered ? i : src, g);
  const ObbQuery q = obb_grid_query(g, image_w, image_h, tile_size, alpha_threshold);
  const int tiles_wide = image_w / tile_size;
  KeyT depth_key = 0;
  if (MODE == 1) {
    const float c = fminf(fmaxf(depth[src], 0.0f), 1.0f);
    depth_key = (KeyT)(uint32_t)(c * 65535.0f);
  } else if (MODE == 0) {
    depth_key = (KeyT)__float_as_uint(depth[src]);   // non-negative float bits keep their order
  }
  int64_t o = cum[i];
  // same (x outer, y inner) order as ti.grouped(ti.ndrange(span.x, span.y)); the order within
  // one gaussian is irrelevant after the sort (all its tiles differ)
  for (int tu = 0; tu < q.span_x; ++tu) {
    for (int tv = 0; tv < q.span_y; ++tv) {
      const int ty = q.min_tile_y + tv;
      if (ty < row_begin || ty >= row_end) continue;
      if (obb_test_tile(q, tu, tv, tile_size)) {
        const int64_t tile_id = (int64_t)(q.min_tile_x + tu) + (int64_t)ty * tiles_wide;
        keys[o] = MODE == 2 ? (KeyT)tile_id
                            : (MODE == 1 ? (KeyT)(depth_key | ((KeyT)tile_id << 16)) : (KeyT)(depth_key | ((KeyT)tile_id << 32)));
        values[o] = (int32_t)src;
        ++o;
      }
    }
  }
}

// ---- frame executor, direct order (no depth pre-sort): the gaussians are visited in storage order, the pairs are
// sorted by tile with a stable radix sort and each tile's run is then depth-sorted on its own (tile_sort.hip).  The
// result is the same (tile, depth key, point index) order as MODE 0 / MODE 2 above.
template <typename T>
__global__ void __launch_bounds__(256)
tile_count_direct_kernel(const float* __restrict__ points, const T* __restrict__ cull_depth, int64_t v, int image_w,
                         int image_h, int tile_size, float alpha_threshold, int row_begin, int row_end,
                         int32_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= v) return;
  if (cull_depth && !(cull_depth[i] > T(0))) { counts[i] = 0; return; }      // same rule as DepthPairs::key
  float g[7];
  load_point7(points, i, g);
  const ObbQuery q = obb_grid_query(g, image_w, image_h, tile_size, alpha_threshold);
  int count = 0;
  for (int tv = 0; tv < q.span_y; ++tv) {
    const int ty = q.min_tile_y + tv;
    if (ty < row_begin || ty >= row_end) continue;
    for (int tu = 0; tu < q.span_x; ++tu)
      if (obb_test_tile(q, tu, tv, tile_size)) ++count;
  }
  counts[i] = count;
}

// key = tile_id << 32 | depth_sort_key(depth) (the 32 bit key of the depth pre-sort), value = point index
template <typename T>
__global__ void __launch_bounds__(256)
tile_emit_direct_kernel(const float* __restrict__ points, const T* __restrict__ depth, const int32_t* __restrict__ cum,
                        int64_t v, int image_w, int image_h, int tile_size, float alpha_threshold, int row_begin,
                        int row_end, int depth16, double near_plane, double far_plane,
                        const int32_t* __restrict__ k_limit, uint64_t* __restrict__ keys, int32_t* __restrict__ values) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= v) return;
  if (k_limit && *k_limit == 0) return;                // overflow of the caller's capacity: write nothing
  int64_t o = cum[i];
  if (cum[i + 1] == o) return;                         // culled, or overlaps no tile of this strip
  float g[7];
  load_point7(points, i, g);
  const ObbQuery q = obb_grid_query(g, image_w, image_h, tile_size, alpha_threshold);
  const int tiles_wide = image_w / tile_size;
  const uint64_t depth_key = depth_sort_key(depth[i], depth16, near_plane, far_plane);
  for (int tu = 0; tu < q.span_x; ++tu) {
    for (int tv = 0; tv < q.span_y; ++tv) {
      const int ty = q.min_tile_y + tv;
      if (ty < row_begin || ty >= row_end) continue;
      if (obb_test_tile(q, tu, tv, tile_size)) {
        const uint64_t tile_id = (uint64_t)((int64_t)(q.min_tile_x + tu) + (int64_t)ty * tiles_wide);
        keys[o] = (tile_id << 32) | depth_key;
        values[o] = (int32_t)i;
        ++o;
      }
    }
  }
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
