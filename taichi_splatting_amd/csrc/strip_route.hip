// strip_route.hip — multi-GPU routing of projected splats to tile-row strips (new design,
// the reference is single GPU).
//
// Rank r renders the tile rows [bounds[r], bounds[r+1]).  A projected splat must reach every rank whose
// strip intersects the tile-row span of the mapper's grid query (splat_math.h obb_grid_query, reference
// taichi_lib/grid_query.py:10-40).  This is a STABLE multi-way split of the local splats by destination
// rank (a splat that straddles a boundary is replicated), done in three HBM-bound passes:
//
//   route_count   1 thread / splat: destination range [first, first + copies) from the row span, and a
//                 per-(destination, block) histogram of the splats routed there (LDS counters).
//   route_offsets 1 workgroup / destination: exclusive scan of that destination's per-block counts
//                 (in place) and its total = the all-to-all split size.
//   route_pack    1 thread / splat: stable rank inside the block per destination (ballot + mbcnt, cross
//                 wave through LDS) -> row slot in the send buffer; writes the 4 (9 + F) byte row
//                 [packed 2D (7) | colour (F) | depth | global id bits] and the slot's source index.
//
// Order inside a destination = local splat index, so depth ties keep breaking by gaussian index after the
// exchange exactly as in the single-GPU sort order.
#include "common.h"

namespace ms {

constexpr int ROUTE_BLOCK = 256;
constexpr int ROUTE_MAX_WORLD = 64;

struct StripBounds {
  int world;
  int b[ROUTE_MAX_WORLD + 1];
};

// first destination and number of destinations of one splat; copies == 0: culled by the mapper
__device__ __forceinline__ void route_of(const float* __restrict__ g, int image_h, int tile_size,
                                         float alpha_threshold, const StripBounds& sb, int& first, int& copies) {
  const float my = g[1], ax = g[2], ay = g[3], sx = g[4], sy = g[5], alpha = g[6];
  const float gs = sqrtf(2.0f * logf(alpha / alpha_threshold));
  const float v1y = ay * sx * gs, v2y = ax * sy * gs;
  const float ey = sqrtf(v1y * v1y + v2y * v2y) + 0.01f;       // slack: rounding may only add rows
  first = 0; copies = 0;
  if (!(ey == ey) || !(my == my) || !(fabsf(ey) < 3.0e38f) || !(fabsf(my) < 3.0e38f)) return;
  const float ts = (float)tile_size;
  const int tiles_high = (image_h + tile_size - 1) / tile_size;
  float lo_f = floorf((my - ey) / ts), hi_f = ceilf((my + ey) / ts);
  lo_f = fminf(fmaxf(lo_f, 0.0f), (float)tiles_high);
  hi_f = fminf(fmaxf(hi_f, lo_f + 1.0f), (float)tiles_high);
  const int lo = (int)lo_f, hi = (int)hi_f;
  if (hi <= lo) return;
  // rank whose strip holds row lo / row hi - 1 (first r with bounds[r + 1] > row)
  int r0 = 0;
  while (r0 < sb.world - 1 && sb.b[r0 + 1] <= lo) ++r0;
  int r1 = r0;
  while (r1 < sb.world - 1 && sb.b[r1 + 1] <= hi - 1) ++r1;
  first = r0; copies = r1 - r0 + 1;
}

__global__ void __launch_bounds__(ROUTE_BLOCK)
route_count_kernel(const float* __restrict__ points7, const float* __restrict__ depth, int v, int image_h, int tile_size,
                   float alpha_threshold, StripBounds sb, int nblocks, int32_t* __restrict__ route,
                   int32_t* __restrict__ block_counts) {
  __shared__ int s_count[ROUTE_MAX_WORLD];
  if (threadIdx.x < ROUTE_MAX_WORLD) s_count[threadIdx.x] = 0;
  __syncthreads();
  const int i = blockIdx.x * ROUTE_BLOCK + threadIdx.x;
  if (i < v) {
    int first = 0, copies = 0;
    // frame executor: the projection does not compact; a culled gaussian (depth <= 0) goes nowhere
    if (!depth || depth[i] > 0.0f)
      route_of(points7 + (int64_t)i * 7, image_h, tile_size, alpha_threshold, sb, first, copies);
    route[i] = first | (copies << 16);
    for (int d = first; d < first + copies; ++d) atomicAdd(&s_count[d], 1);
  }
  __syncthreads();
  if (threadIdx.x < sb.world) block_counts[(int64_t)threadIdx.x * nblocks + blockIdx.x] = s_count[threadIdx.x];
}

// one workgroup per destination: in-place exclusive scan over the blocks + total
__global__ void __launch_bounds__(1024)
route_offsets_kernel(int32_t* __restrict__ block_counts, int nblocks, int64_t* __restrict__ send_counts) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  int32_t* c = block_counts + (int64_t)blockIdx.x * nblocks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nblocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int x = i < nblocks ? c[i] : 0;
    int incl = x;
    for (int off = 1; off < 64; off <<= 1) {
      const int y = __shfl_up(incl, off, 64);
      if (lane >= off) incl += y;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int wave_base = 0;
    for (int w = 0; w < wave; ++w) wave_base += s_wave[w];
    const int carry = s_carry;
    if (i < nblocks) c[i] = carry + wave_base + incl - x;
    __syncthreads();
    if (threadIdx.x == 1023) s_carry = carry + wave_base + incl;
    __syncthreads();
  }
  if (threadIdx.x == 0) send_counts[blockIdx.x] = s_carry;
}

__global__ void __launch_bounds__(ROUTE_BLOCK)
route_pack_kernel(const float* __restrict__ points7, const float* __restrict__ feats, const float* __restrict__ depths,
                  const int64_t* __restrict__ ids, int f, int v, int world, int nblocks, int64_t index_offset,
                  const int32_t* __restrict__ route, const int32_t* __restrict__ block_offsets,
                  const int64_t* __restrict__ send_counts, int64_t bucket_capacity, int32_t* __restrict__ overflow,
                  float* __restrict__ rows, int64_t* __restrict__ send_index, int32_t* __restrict__ slots,
                  float* __restrict__ colour_rows) {
  constexpr int WAVES = ROUTE_BLOCK / 64;
  __shared__ int s_wave_count[WAVES][ROUTE_MAX_WORLD];
  __shared__ int64_t s_bucket_start[ROUTE_MAX_WORLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * ROUTE_BLOCK + threadIdx.x;
  int first = 0, copies = 0;
  if (i < v) { const int r = route[i]; first = r & 0xffff; copies = r >> 16; }
  if (threadIdx.x == 0) {
    // bucket_capacity > 0: every destination owns a FIXED range of `bucket_capacity` rows (the all-to-all then has
    // equal splits known to the host without reading the counts back); rows that do not fit are dropped and flagged
    int64_t acc = 0;
    for (int d = 0; d < world; ++d) {
      s_bucket_start[d] = bucket_capacity > 0 ? (int64_t)d * bucket_capacity : acc;
      acc += send_counts[d];
      if (bucket_capacity > 0 && blockIdx.x == 0 && send_counts[d] > bucket_capacity && overflow) *overflow = 1;
    }
  }
  // wave totals per destination -> LDS (uniform loop: every lane takes part in every ballot)
  for (int d = 0; d < world; ++d) {
    const unsigned long long m = __ballot(copies > 0 && d >= first && d < first + copies);
    if (lane == 0) s_wave_count[wave][d] = __popcll(m);
  }
  __syncthreads();
  // colour_rows != NULL: the colours travel in a buffer (and a collective) of their own — rows = [packed 2D | depth | id]
  const int width = colour_rows ? 9 : 9 + f;
  float g[7];
  float depth = 0.f;
  int id_bits = 0;
  if (copies > 0) {
#pragma unroll
    for (int k = 0; k < 7; ++k) g[k] = points7[(int64_t)i * 7 + k];
    depth = depths[i];
    id_bits = (int)((ids ? ids[i] : (int64_t)i) + index_offset);
  }
  const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int d = 0; d < world; ++d) {
    const bool in = copies > 0 && d >= first && d < first + copies;
    const unsigned long long m = __ballot(in);
    if (!in) continue;
    int before = 0;
    for (int w = 0; w < wave; ++w) before += s_wave_count[w][d];
    const int64_t in_bucket = (int64_t)block_offsets[(int64_t)d * nblocks + blockIdx.x] + before + __popcll(m & lt_mask);
    if (bucket_capacity > 0 && in_bucket >= bucket_capacity) {
      if (slots) slots[(int64_t)i * world + (d - first)] = -1;        // dropped copy (overflow is flagged)
      continue;
    }
    const int64_t slot = s_bucket_start[d] + in_bucket;
    if (slots) slots[(int64_t)i * world + (d - first)] = (int32_t)slot;
    float* row = rows + slot * width;
#pragma unroll
    for (int k = 0; k < 7; ++k) row[k] = g[k];
    if (colour_rows) {
      for (int k = 0; k < f; ++k) colour_rows[slot * f + k] = feats[(int64_t)i * f + k];
      row[7] = depth;
      row[8] = __int_as_float(id_bits);
    } else {
      for (int k = 0; k < f; ++k) row[7 + k] = feats[(int64_t)i * f + k];
      row[7 + f] = depth;
      row[8 + f] = __int_as_float(id_bits);
    }
    send_index[slot] = i;
  }
}

__global__ void __launch_bounds__(256)
strip_unpack_kernel(const float* __restrict__ rows, int64_t m, int f, float* __restrict__ points7,
                    float* __restrict__ feats, float* __restrict__ depths, int64_t* __restrict__ ids) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const float* row = rows + i * (9 + f);
#pragma unroll
  for (int k = 0; k < 7; ++k) points7[i * 7 + k] = row[k];
  for (int k = 0; k < f; ++k) feats[i * f + k] = row[7 + k];
  depths[i] = row[7 + f];
  ids[i] = (int64_t)__float_as_int(row[8 + f]);
}

// gradients coming home: row `slot` of the reverse all-to-all belongs to local splat send_index[slot].  One
// thread per ELEMENT (coalesced reads of the rows, near-coalesced writes: slots of a bucket are in index
// order).  A splat with a single copy (the common case) is a plain store; only splats that straddle a strip
// boundary need float atomics to sum their copies (onto the zero-initialised output).
__global__ void __launch_bounds__(256)
return_grads_kernel(const float* __restrict__ back, const int64_t* __restrict__ send_index,
                    const int32_t* __restrict__ route, int f, int64_t total, float* __restrict__ grad_points7,
                    float* __restrict__ grad_feats) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int width = 7 + f;
  const int64_t slot = e / width;
  const int k = (int)(e - slot * width);
  const int64_t i = send_index[slot];
  if (i < 0) return;                      // unused slot of a fixed-capacity bucket
  float* dst = k < 7 ? grad_points7 + i * 7 + k : grad_feats + i * f + (k - 7);
  const float v = back[e];
  if ((route[i] >> 16) == 1) *dst = v;
  else atomic_add_noret(dst, v);
}

// the same into interleaved rows (V, 7 + f): element e of the reverse buffer goes to element k of row send_index[slot]
template <int WIDTH>
__global__ void __launch_bounds__(256)
return_rows_kernel(const float* __restrict__ back, const int64_t* __restrict__ send_index,
                   const int32_t* __restrict__ route, int64_t slots, float* __restrict__ rows) {
  // one thread per slot; rows of an even number of floats move as 64-bit words
  const int64_t slot = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (slot >= slots) return;
  const int64_t i = send_index[slot];
  if (i < 0) return;                      // unused slot of a fixed-capacity bucket
  const float* src = back + slot * WIDTH;
  float* dst = rows + i * WIDTH;
  float v[WIDTH];
  if (WIDTH % 2 == 0) {
#pragma unroll
    for (int k = 0; k < WIDTH; k += 2) { const float2 p = *reinterpret_cast<const float2*>(src + k); v[k] = p.x; v[k + 1] = p.y; }
  } else {
#pragma unroll
    for (int k = 0; k < WIDTH; ++k) v[k] = src[k];
  }
  if ((route[i] >> 16) == 1) {
    if (WIDTH % 2 == 0) {
#pragma unroll
      for (int k = 0; k < WIDTH; k += 2) *reinterpret_cast<float2*>(dst + k) = make_float2(v[k], v[k + 1]);
    } else {
#pragma unroll
      for (int k = 0; k < WIDTH; ++k) dst[k] = v[k];
    }
  } else {
#pragma unroll
    for (int k = 0; k < WIDTH; ++k) atomic_add_noret(dst + k, v[k]);
  }
}

}  // namespace ms

using namespace ms;

static int fill_bounds(const int32_t* bounds, int world, int image_h, int tile_size, StripBounds* sb, const char* fn) {
  if (!bounds || world < 1 || world > ROUTE_MAX_WORLD) {
    set_error("%s: world must be in [1, %d] and bounds non-null", fn, ROUTE_MAX_WORLD);
    return MS_ERR_BAD_ARG;
  }
  const int tiles_high = (image_h + tile_size - 1) / tile_size;
  sb->world = world;
  for (int r = 0; r <= world; ++r) {
    sb->b[r] = bounds[r];
    if (bounds[r] < 0 || bounds[r] > tiles_high || (r > 0 && bounds[r] < bounds[r - 1])) {
      set_error("%s: bounds must be non-decreasing in [0, tiles_high]", fn);
      return MS_ERR_BAD_ARG;
    }
  }
  if (bounds[0] != 0 || bounds[world] != tiles_high) {
    set_error("%s: bounds must cover [0, tiles_high = %d]", fn, tiles_high);
    return MS_ERR_BAD_ARG;
  }
  return 0;
}

extern "C" int ms_strip_route_blocks(int v) { return (int)div_up(v > 0 ? v : 1, ROUTE_BLOCK); }

extern "C" int ms_strip_route_count(const float* points7, const float* depth, int v, int image_h, int tile_size,
                                    float alpha_threshold, const int32_t* bounds_host, int world, int32_t* out_route,
                                    int32_t* out_block_counts, int64_t* out_send_counts, void* stream) {
  MS_CHECK_ARG(v >= 0 && image_h > 0 && tile_size > 0, "bad sizes");
  MS_CHECK_ARG(out_block_counts && out_send_counts && (v == 0 || (points7 && out_route)), "null pointer");
  StripBounds sb;
  int rc = fill_bounds(bounds_host, world, image_h, tile_size, &sb, "ms_strip_route_count");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int nblocks = ms_strip_route_blocks(v);
  route_count_kernel<<<nblocks, ROUTE_BLOCK, 0, s>>>(points7, depth, v, image_h, tile_size, alpha_threshold, sb, nblocks,
                                                     out_route, out_block_counts);
  route_offsets_kernel<<<world, 1024, 0, s>>>(out_block_counts, nblocks, out_send_counts);
  MS_CHECK_LAUNCH();
  return 0;
}

static int route_pack_launch(const float* points7, const float* features, const float* depths, const int64_t* ids, int f,
                             int v, int world, int64_t index_offset, const int32_t* route, const int32_t* block_offsets,
                             const int64_t* send_counts, int64_t bucket_capacity, int32_t* overflow_flag, float* out_rows,
                             int64_t* out_send_index, int32_t* out_slots, float* out_colour_rows, void* stream) {
  MS_CHECK_ARG(v >= 0 && f >= 0 && world >= 1 && world <= ROUTE_MAX_WORLD, "bad sizes");
  if (v == 0) return 0;
  MS_CHECK_ARG(points7 && depths && route && block_offsets && send_counts && (f == 0 || features), "null pointer");
  MS_CHECK_ARG(out_rows && out_send_index, "null output");
  const int nblocks = ms_strip_route_blocks(v);
  route_pack_kernel<<<nblocks, ROUTE_BLOCK, 0, (hipStream_t)stream>>>(points7, features, depths, ids, f, v, world,
                                                                       nblocks, index_offset, route, block_offsets,
                                                                       send_counts, bucket_capacity, overflow_flag,
                                                                       out_rows, out_send_index, out_slots, out_colour_rows);
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_strip_route_pack_slots(const float* points7, const float* features, const float* depths,
                                         const int64_t* ids, int f, int v, int world, int64_t index_offset,
                                         const int32_t* route, const int32_t* block_offsets, const int64_t* send_counts,
                                         int64_t bucket_capacity, int32_t* overflow_flag,
                                         float* out_rows, int64_t* out_send_index, int32_t* out_slots, void* stream) {
  return route_pack_launch(points7, features, depths, ids, f, v, world, index_offset, route, block_offsets, send_counts,
                           bucket_capacity, overflow_flag, out_rows, out_send_index, out_slots, nullptr, stream);
}

extern "C" int ms_strip_route_pack_split(const float* points7, const float* features, const float* depths,
                                         const int64_t* ids, int f, int v, int world, int64_t index_offset,
                                         const int32_t* route, const int32_t* block_offsets, const int64_t* send_counts,
                                         int64_t bucket_capacity, int32_t* overflow_flag, float* out_geometry_rows,
                                         float* out_colour_rows, int64_t* out_send_index, int32_t* out_slots, void* stream) {
  MS_CHECK_ARG(out_colour_rows != nullptr && f >= 1, "ms_strip_route_pack_split: colour rows");
  return route_pack_launch(points7, features, depths, ids, f, v, world, index_offset, route, block_offsets, send_counts,
                           bucket_capacity, overflow_flag, out_geometry_rows, out_send_index, out_slots, out_colour_rows, stream);
}

extern "C" int ms_strip_route_pack(const float* points7, const float* features, const float* depths,
                                   const int64_t* ids, int f, int v, int world, int64_t index_offset,
                                   const int32_t* route, const int32_t* block_offsets, const int64_t* send_counts,
                                   int64_t bucket_capacity, int32_t* overflow_flag,
                                   float* out_rows, int64_t* out_send_index, void* stream) {
  return ms_strip_route_pack_slots(points7, features, depths, ids, f, v, world, index_offset, route, block_offsets,
                                   send_counts, bucket_capacity, overflow_flag, out_rows, out_send_index, nullptr, stream);
}

extern "C" int ms_strip_unpack(const float* rows, int64_t m, int f, float* out_points7, float* out_features,
                               float* out_depths, int64_t* out_ids, void* stream) {
  MS_CHECK_ARG(m >= 0 && f >= 0, "bad sizes");
  if (m == 0) return 0;
  // (f = 0, out_features = NULL: the 9-float geometry rows of ms_strip_route_pack_split)
  MS_CHECK_ARG(rows && out_points7 && out_depths && out_ids && (f == 0 || out_features), "null pointer");
  strip_unpack_kernel<<<(unsigned)div_up(m, 256), 256, 0, (hipStream_t)stream>>>(rows, m, f, out_points7, out_features,
                                                                                   out_depths, out_ids);
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_strip_return_grads(const float* back_rows, const int64_t* send_index, const int32_t* route,
                                     int f, int64_t s, float* grad_points7, float* grad_features, void* stream) {
  MS_CHECK_ARG(s >= 0 && f >= 0, "bad sizes");
  if (s == 0) return 0;
  MS_CHECK_ARG(back_rows && send_index && route && grad_points7 && (f == 0 || grad_features), "null pointer");
  const int64_t total = s * (7 + f);
  return_grads_kernel<<<(unsigned)div_up(total, 256), 256, 0, (hipStream_t)stream>>>(back_rows, send_index, route, f,
                                                                                       total, grad_points7, grad_features);
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_strip_return_rows(const float* back_rows, const int64_t* send_index, const int32_t* route,
                                    int f, int64_t s, float* grad_rows, void* stream) {
  MS_CHECK_ARG(s >= 0 && f >= 0, "bad sizes");
  if (s == 0) return 0;
  MS_CHECK_ARG(back_rows && send_index && route && grad_rows, "null pointer");
  MS_CHECK_ARG(f >= 1 && f <= 4, "ms_strip_return_rows: 1..4 colour channels");
  const dim3 grid((unsigned)div_up(s, 256)), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (f) {
    case 1: return_rows_kernel<8><<<grid, block, 0, st>>>(back_rows, send_index, route, s, grad_rows); break;
    case 2: return_rows_kernel<9><<<grid, block, 0, st>>>(back_rows, send_index, route, s, grad_rows); break;
    case 3: return_rows_kernel<10><<<grid, block, 0, st>>>(back_rows, send_index, route, s, grad_rows); break;
    default: return_rows_kernel<11><<<grid, block, 0, st>>>(back_rows, send_index, route, s, grad_rows); break;
  }
  MS_CHECK_LAUNCH();
  return 0;
}
