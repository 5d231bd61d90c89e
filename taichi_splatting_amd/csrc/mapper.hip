// mapper.hip — OBB-vs-tile overlap test of the tile mapper: per-gaussian overlap count and
// (sort key, point index) emission.  Replaces tile_overlaps_kernel / generate_sort_keys_kernel
// (mapper/tile_mapper.py:76-86,115-146) on top of grid_query.py:10-91 (see splat_math.h).
//
// The overlap decisions are float comparisons that must agree between the count and the emit
// pass (and with the CPU oracle), so this file is compiled with FP contraction off: no FMA
// formation, every product and sum rounded individually, exactly like the oracle's numpy float32.
#pragma clang fp contract(off)
#include "common.h"

namespace ms {

__device__ __forceinline__ void load_point7(const float* __restrict__ points, int64_t i, float g[7]) {
#pragma unroll
  for (int k = 0; k < 7; ++k) g[k] = points[i * 7 + k];
}

__global__ void __launch_bounds__(256)
tile_count_kernel(const float* __restrict__ points, int64_t v, int image_w, int image_h, int tile_size,
                  float alpha_threshold, int row_begin, int row_end, int32_t* __restrict__ counts) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= v) return;
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

template <typename KeyT, bool DEPTH16>
__global__ void __launch_bounds__(256)
tile_emit_kernel(const float* __restrict__ points, const float* __restrict__ depth,
                 const int32_t* __restrict__ cum, int64_t v, int image_w, int image_h, int tile_size,
                 float alpha_threshold, int row_begin, int row_end, KeyT* __restrict__ keys,
                 int32_t* __restrict__ values) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= v) return;
  float g[7];
  load_point7(points, i, g);
  const ObbQuery q = obb_grid_query(g, image_w, image_h, tile_size, alpha_threshold);
  const int tiles_wide = image_w / tile_size;
  const float d = depth[i];
  KeyT depth_key;
  if (DEPTH16) {
    // tile_mapper.py:55-61: clamp(depth, 0, 1) * 65535 truncated to an integer
    const float c = fminf(fmaxf(d, 0.0f), 1.0f);
    depth_key = (KeyT)(uint32_t)(c * 65535.0f);
  } else {
    // tile_mapper.py:36-42: non-negative float bits keep their order as unsigned integers
    depth_key = (KeyT)__float_as_uint(d);
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
        keys[o] = DEPTH16 ? (KeyT)(depth_key | ((KeyT)tile_id << 16)) : (KeyT)(depth_key | ((KeyT)tile_id << 32));
        values[o] = (int32_t)i;
        ++o;
      }
    }
  }
}

}  // namespace ms

using namespace ms;

extern "C" int ms_tile_count(const float* points7, int64_t v, int image_w, int image_h, int tile_size,
                             float alpha_threshold, int tile_row_begin, int tile_row_end,
                             int32_t* out_counts, void* stream) {
  MS_CHECK_ARG(v >= 0, "v < 0");
  MS_CHECK_ARG(tile_size > 0 && image_w > 0 && image_h > 0, "bad image/tile size");
  MS_CHECK_ARG(image_w % tile_size == 0 && image_h % tile_size == 0, "image size must be padded to the tile size");
  if (v == 0) return 0;
  MS_CHECK_ARG(points7 && out_counts, "null pointer");
  tile_count_kernel<<<dim3((unsigned)div_up(v, 256)), dim3(256), 0, (hipStream_t)stream>>>(
      points7, v, image_w, image_h, tile_size, alpha_threshold, tile_row_begin, tile_row_end, out_counts);
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_tile_emit(const float* points7, const float* depth, const int32_t* cum, int64_t v,
                            int image_w, int image_h, int tile_size, float alpha_threshold,
                            int tile_row_begin, int tile_row_end, int key_bytes, void* out_keys,
                            int32_t* out_values, void* stream) {
  MS_CHECK_ARG(v >= 0, "v < 0");
  MS_CHECK_ARG(key_bytes == 4 || key_bytes == 8, "key_bytes must be 4 or 8");
  MS_CHECK_ARG(tile_size > 0 && image_w > 0 && image_h > 0, "bad image/tile size");
  MS_CHECK_ARG(image_w % tile_size == 0 && image_h % tile_size == 0, "image size must be padded to the tile size");
  if (key_bytes == 4) {
    const int64_t tiles = (int64_t)(image_w / tile_size) * (image_h / tile_size);
    MS_CHECK_ARG(tiles <= 65536, "use_depth16 keys hold a 16 bit tile id: too many tiles");
  }
  if (v == 0) return 0;
  MS_CHECK_ARG(points7 && depth && cum && out_keys && out_values, "null pointer");
  const dim3 block(256), grid((unsigned)div_up(v, 256));
  if (key_bytes == 8)
    tile_emit_kernel<uint64_t, false><<<grid, block, 0, (hipStream_t)stream>>>(
        points7, depth, cum, v, image_w, image_h, tile_size, alpha_threshold, tile_row_begin, tile_row_end,
        (uint64_t*)out_keys, out_values);
  else
    tile_emit_kernel<uint32_t, true><<<grid, block, 0, (hipStream_t)stream>>>(
        points7, depth, cum, v, image_w, image_h, tile_size, alpha_threshold, tile_row_begin, tile_row_end,
        (uint32_t*)out_keys, out_values);
  MS_CHECK_LAUNCH();
  return 0;
}
