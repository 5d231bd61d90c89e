// raster.hip — tile-based alpha compositing of sorted 2D gaussians: forward and backward.
//
// Work decomposition (gfx950, wave64):
//   * one workgroup per screen tile (tile_size^2 threads), one wavefront per 8x8 pixel patch,
//     one lane per pixel.  The per-gaussian gradient sum of the backward pass is therefore a
//     pure in-wave reduction (DPP row_shr / row_bcast, common.h) followed by one atomic per value
//     from lane 63 — no LDS atomics, no cross-wave combine (the reference needs a 32-lane shuffle
//     tree + shared atomics + global atomics, rasterizer/backward.py:200-224).
//   * the tile's depth-sorted splat list is staged through LDS in batches; the staging thread also
//     derives the axis-aligned extent of the splat's contribution ellipse (alpha_pt * g >
//     alpha_threshold  <=>  |x|_ellipse < sqrt(2 ln(alpha_pt / threshold))).
//   * every wave tests the 64 staged splats of a round against ITS patch (one splat per lane),
//     ballots the hits and walks only the set bits with a scalar loop: splats whose ellipse
//     misses the patch cost 1/64 of an evaluation instead of a full one.  The cull is
//     conservative (extent inflated), so results are those of the exhaustive loop.
//
// Semantics follow rasterizer/forward.py:39-135 and rasterizer/backward.py:97-224, with the
// corrected in-group loop bound.
#include "common.h"

namespace ms {

constexpr int RASTER_MAX_F = 4;

template <typename T> struct RasterParams {
  int width, height, tiles_wide, tile_begin;   // tile_begin = first tile id of the strip
  T clamp_max_alpha, alpha_threshold, saturate_threshold;
};

// staged splat: AoS so the wave-uniform reads in the hot loop are a few wide broadcast LDS reads
template <typename T, int F> struct alignas(16) Splat {
  T mx, my, ax, ay;       // mean, axis
  T isx, isy, sx, sy;     // 1/sigma and sigma
  T alpha;
  T f[F];
};

template <typename T> struct CullBox { T cx, cy, ex, ey; };

__device__ __forceinline__ float fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ double fast_exp(double x) { return exp(x); }
__device__ __forceinline__ float fast_div(float a, float b) { return __fdividef(a, b); }
__device__ __forceinline__ double fast_div(double a, double b) { return a / b; }

template <typename T, int F, int BATCH>
__device__ __forceinline__ void stage_batch(const T* __restrict__ points, const T* __restrict__ feats,
                                            const int32_t* __restrict__ o2p, int begin, int count,
                                            T alpha_threshold, bool cull, Splat<T, F>* s_splat,
                                            CullBox<T>* s_cull, int32_t* s_id) {
  for (int t = threadIdx.x; t < count; t += blockDim.x) {
    const int32_t id = o2p[begin + t];
    const T* g = points + (int64_t)id * 7;
    Splat<T, F> s;
    s.mx = g[0]; s.my = g[1]; s.ax = g[2]; s.ay = g[3];
    s.sx = g[4]; s.sy = g[5]; s.alpha = g[6];
    s.isx = T(1) / s.sx; s.isy = T(1) / s.sy;
#pragma unroll
    for (int c = 0; c < F; ++c) s.f[c] = feats[(int64_t)id * F + c];
    s_splat[t] = s;
    s_id[t] = id;

    CullBox<T> b;
    b.cx = s.mx; b.cy = s.my;
    if (cull) {
      // half extents of the bounding box of the ellipse  alpha * g == threshold; NaN (alpha below
      // the threshold) fails every comparison below and culls the splat, which cannot contribute
      const T gs = t_sqrt(2 * t_log(s.alpha / alpha_threshold));
      const T v1x = s.ax * s.sx * gs, v1y = s.ay * s.sx * gs;
      const T v2x = -s.ay * s.sy * gs, v2y = s.ax * s.sy * gs;
      b.ex = t_sqrt(v1x * v1x + v2x * v2x) * T(1.001) + T(0.01);
      b.ey = t_sqrt(v1y * v1y + v2y * v2y) * T(1.001) + T(0.01);
    } else {
      b.ex = T(1e30); b.ey = T(1e30);
    }
    s_cull[t] = b;
  }
}

template <typename T>
__device__ __forceinline__ bool patch_hit(const CullBox<T>& b, T x0, T y0) {
  // pixel centres of the patch span [x0 + 0.5, x0 + 7.5]
  return (b.cx + b.ex >= x0 + T(0.5)) && (b.cx - b.ex <= x0 + T(7.5)) &&
         (b.cy + b.ey >= y0 + T(0.5)) && (b.cy - b.ey <= y0 + T(7.5));
}

template <typename T, int F, bool AA>
__device__ __forceinline__ T splat_pdf(const Splat<T, F>& s, T px, T py) {
  if (AA) {
    const T g[6] = {s.mx, s.my, s.ax, s.ay, s.sx, s.sy};
    return gaussian_pdf_antialias(px, py, g);
  } else {
    const T dx = px - s.mx, dy = py - s.my;
    const T tx = (dx * s.ax + dy * s.ay) * s.isx;
    const T ty = (dy * s.ax - dx * s.ay) * s.isy;
    return fast_exp(T(-0.5) * (tx * tx + ty * ty));
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <typename T, int F, int TS, bool AA, bool BLEND, bool VIS>
__global__ void __launch_bounds__(TS * TS)
raster_fwd_kernel(const T* __restrict__ points, const T* __restrict__ feats,
                  const int32_t* __restrict__ ranges, const int32_t* __restrict__ o2p,
                  RasterParams<T> rp, T* __restrict__ image, T* __restrict__ image_alpha,
                  T* __restrict__ visibility) {
  constexpr int THREADS = TS * TS;
  constexpr int BATCH = THREADS < 256 ? THREADS : 256;
  constexpr int WAVES_WIDE = TS / 8;

  __shared__ Splat<T, F> s_splat[BATCH];
  __shared__ CullBox<T> s_cull[BATCH];
  __shared__ int32_t s_id[BATCH];

  const int tile_id = rp.tile_begin + blockIdx.x;
  const int tile_u = tile_id % rp.tiles_wide, tile_v = tile_id / rp.tiles_wide;
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int patch_x = tile_u * TS + (wave % WAVES_WIDE) * 8;
  const int patch_y = tile_v * TS + (wave / WAVES_WIDE) * 8;
  const int pix_x = patch_x + (lane & 7), pix_y = patch_y + (lane >> 3);
  const T px = T(pix_x) + T(0.5), py = T(pix_y) + T(0.5);
  const bool in_bounds = pix_x < rp.width && pix_y < rp.height;

  T C[F];
#pragma unroll
  for (int c = 0; c < F; ++c) C[c] = T(0);
  T W = in_bounds ? T(0) : T(1);
  bool saturated = false;

  const int start = ranges[tile_id * 2 + 0], end = ranges[tile_id * 2 + 1];

  for (int begin = start; begin < end; begin += BATCH) {
    const int count = (end - begin) < BATCH ? (end - begin) : BATCH;
    __syncthreads();   // previous batch fully consumed
    stage_batch<T, F, BATCH>(points, feats, o2p, begin, count, rp.alpha_threshold, !AA, s_splat, s_cull, s_id);
    __syncthreads();

    for (int r = 0; r < count; r += 64) {
      const int j = r + lane;
      bool hit = false;
      if (j < count) hit = patch_hit(s_cull[j], T(patch_x), T(patch_y));
      unsigned long long m = __ballot(hit);
      while (m) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        const Splat<T, F>& s = s_splat[r + b];

        T alpha = s.alpha * splat_pdf<T, F, AA>(s, px, py);
        alpha = t_min(alpha, rp.clamp_max_alpha);
        T weight = T(0);
        if (alpha > rp.alpha_threshold) {
          if (BLEND) {
            weight = alpha * (T(1) - W);
            W += weight;
#pragma unroll
            for (int c = 0; c < F; ++c) C[c] += s.f[c] * weight;
          } else if (!saturated) {
            // quantile render (forward.py:107-112): take the feature of the splat at which the
            // accumulated weight first reaches 1 - saturate_threshold
            weight = alpha * (T(1) - W);
            W += weight;
            if (W >= T(1) - rp.saturate_threshold) {
#pragma unroll
              for (int c = 0; c < F; ++c) C[c] = s.f[c];
              saturated = true;
            }
          }
        }
        if (VIS) {
          if (__ballot(weight != T(0))) {
            const T total = wave_sum_to_lane63(weight);
            if (lane == 63) atomic_add_noret(visibility + s_id[r + b], total);
          }
        }
      }
    }
  }

  if (in_bounds) {
    const int64_t p = (int64_t)pix_y * rp.width + pix_x;
#pragma unroll
    for (int c = 0; c < F; ++c) image[p * F + c] = C[c];
    image_alpha[p] = BLEND ? W : (W > T(0) ? T(1) : T(0));
  }
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
template <typename T, int F, int TS, bool AA, bool HEUR>
__global__ void __launch_bounds__(TS * TS)
raster_bwd_kernel(const T* __restrict__ points, const T* __restrict__ feats,
                  const int32_t* __restrict__ ranges, const int32_t* __restrict__ o2p,
                  const T* __restrict__ image, const T* __restrict__ grad_image, RasterParams<T> rp,
                  T* __restrict__ grad_points, T* __restrict__ grad_feats, T* __restrict__ heuristic) {
  constexpr int THREADS = TS * TS;
  constexpr int BATCH = THREADS < 256 ? THREADS : 256;
  constexpr int WAVES_WIDE = TS / 8;

  __shared__ Splat<T, F> s_splat[BATCH];
  __shared__ CullBox<T> s_cull[BATCH];
  __shared__ int32_t s_id[BATCH];

  const int tile_id = rp.tile_begin + blockIdx.x;
  const int tile_u = tile_id % rp.tiles_wide, tile_v = tile_id / rp.tiles_wide;
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int patch_x = tile_u * TS + (wave % WAVES_WIDE) * 8;
  const int patch_y = tile_v * TS + (wave / WAVES_WIDE) * 8;
  const int pix_x = patch_x + (lane & 7), pix_y = patch_y + (lane >> 3);
  const T px = T(pix_x) + T(0.5), py = T(pix_y) + T(0.5);
  const bool in_bounds = pix_x < rp.width && pix_y < rp.height;

  // per-pixel state (backward.py:97-110): W accumulated weight, R colour still to come, G dL/dC
  T R[F], G[F];
  T W = T(1);
#pragma unroll
  for (int c = 0; c < F; ++c) { R[c] = T(0); G[c] = T(0); }
  if (in_bounds) {
    const int64_t p = (int64_t)pix_y * rp.width + pix_x;
#pragma unroll
    for (int c = 0; c < F; ++c) { R[c] = image[p * F + c]; G[c] = grad_image[p * F + c]; }
    W = T(0);
  }

  const int start = ranges[tile_id * 2 + 0], end = ranges[tile_id * 2 + 1];

  for (int begin = start; begin < end; begin += BATCH) {
    const int count = (end - begin) < BATCH ? (end - begin) : BATCH;
    // tile-wide early out once every pixel is saturated (backward.py:116)
    if (__syncthreads_and(W >= rp.saturate_threshold)) break;
    stage_batch<T, F, BATCH>(points, feats, o2p, begin, count, rp.alpha_threshold, !AA, s_splat, s_cull, s_id);
    __syncthreads();

    // wave-wide early out (backward.py:142)
    if (__ballot(W < rp.saturate_threshold) == 0) continue;

    for (int r = 0; r < count; r += 64) {
      const int j = r + lane;
      bool hit = false;
      if (j < count) hit = patch_hit(s_cull[j], T(patch_x), T(patch_y));
      unsigned long long m = __ballot(hit);
      while (m) {
        const int b = __builtin_ctzll(m);
        m &= m - 1;
        const Splat<T, F>& s = s_splat[r + b];

        T dmean[2], daxis[2], dsigma[2];
        T p;
        if (AA) {
          const T g[6] = {s.mx, s.my, s.ax, s.ay, s.sx, s.sy};
          p = gaussian_pdf_antialias_with_grad(px, py, g, dmean, daxis, dsigma);
        } else {
          const T dx = px - s.mx, dy = py - s.my;
          const T tx = (dx * s.ax + dy * s.ay) * s.isx;
          const T ty = (dy * s.ax - dx * s.ay) * s.isy;
          const T tx2 = tx * tx, ty2 = ty * ty;
          p = fast_exp(T(-0.5) * (tx2 + ty2));
          dsigma[0] = tx2 * p * s.isx;
          dsigma[1] = ty2 * p * s.isy;
          const T tx_s = tx * s.isx * p, ty_s = ty * s.isy * p;
          daxis[0] = -(tx_s * dx + ty_s * dy);
          daxis[1] = ty_s * dx - tx_s * dy;
          dmean[0] = tx_s * s.ax - ty_s * s.ay;
          dmean[1] = tx_s * s.ay + ty_s * s.ax;
        }

        T acc[7 + F + 2];
#pragma unroll
        for (int k = 0; k < 7 + F + 2; ++k) acc[k] = T(0);

        T alpha = s.alpha * p;
        const bool active = alpha > rp.alpha_threshold && W < rp.saturate_threshold;
        if (active) {
          alpha = t_min(alpha, rp.clamp_max_alpha);
          const T Ti = T(1) - W;
          const T weight = alpha * Ti;
          W += weight;
          const T inv = fast_div(T(1), T(1) - alpha);
          T alpha_grad = T(0);
#pragma unroll
          for (int c = 0; c < F; ++c) {
            R[c] -= s.f[c] * weight;
            alpha_grad += (s.f[c] * Ti - R[c] * inv) * G[c];
            acc[7 + c] = weight * G[c];
          }
          const T aag = s.alpha * alpha_grad;   // straight-through clamp (backward.py:158-163)
          acc[0] = aag * dmean[0]; acc[1] = aag * dmean[1];
          acc[2] = aag * daxis[0]; acc[3] = aag * daxis[1];
          acc[4] = aag * dsigma[0]; acc[5] = aag * dsigma[1];
          acc[6] = p * alpha_grad;
          if (HEUR) {
            acc[7 + F] = aag * aag;
            acc[7 + F + 1] = t_abs(acc[0]) + t_abs(acc[1]);
          }
        }

        if (__ballot(active)) {
          const int32_t id = s_id[r + b];
          if (grad_points) {
#pragma unroll
            for (int k = 0; k < 7; ++k) {
              const T total = wave_sum_to_lane63(acc[k]);
              if (lane == 63) atomic_add_noret(grad_points + (int64_t)id * 7 + k, total);
            }
          }
          if (grad_feats) {
#pragma unroll
            for (int c = 0; c < F; ++c) {
              const T total = wave_sum_to_lane63(acc[7 + c]);
              if (lane == 63) atomic_add_noret(grad_feats + (int64_t)id * F + c, total);
            }
          }
          if (HEUR) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const T total = wave_sum_to_lane63(acc[7 + F + k]);
              if (lane == 63) atomic_add_noret(heuristic + (int64_t)id * 2 + k, total);
            }
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
template <typename T>
static RasterParams<T> make_raster_params(int w, int h, const ms_raster_config* cfg, int row_begin) {
  RasterParams<T> rp;
  rp.width = w; rp.height = h;
  rp.tiles_wide = (w + cfg->tile_size - 1) / cfg->tile_size;
  rp.tile_begin = row_begin * rp.tiles_wide;
  rp.clamp_max_alpha = (T)cfg->clamp_max_alpha;
  rp.alpha_threshold = (T)cfg->alpha_threshold;
  rp.saturate_threshold = (T)cfg->saturate_threshold;
  return rp;
}

template <typename T, int F, int TS>
static void launch_fwd(const void* points, const void* feats, const int32_t* ranges, const int32_t* o2p,
                       int w, int h, const ms_raster_config* cfg, void* image, void* alpha, void* vis,
                       int row_begin, int num_tiles, hipStream_t s) {
  const RasterParams<T> rp = make_raster_params<T>(w, h, cfg, row_begin);
  const dim3 grid((unsigned)num_tiles), block(TS * TS);
#define MS_FWD(AA, BLEND, VIS)                                                                        \
  raster_fwd_kernel<T, F, TS, AA, BLEND, VIS><<<grid, block, 0, s>>>(                                 \
      (const T*)points, (const T*)feats, ranges, o2p, rp, (T*)image, (T*)alpha, (T*)vis)
  const bool aa = cfg->antialias, blend = cfg->use_alpha_blending, visf = cfg->compute_visibility && vis;
  if (!blend) { if (aa) MS_FWD(true, false, false); else MS_FWD(false, false, false); }
  else if (aa) { if (visf) MS_FWD(true, true, true); else MS_FWD(true, true, false); }
  else { if (visf) MS_FWD(false, true, true); else MS_FWD(false, true, false); }
#undef MS_FWD
}

template <typename T, int F, int TS>
static void launch_bwd(const void* points, const void* feats, const int32_t* ranges, const int32_t* o2p,
                       const void* image, const void* grad_image, int w, int h, const ms_raster_config* cfg,
                       void* gp, void* gf, void* heur, int row_begin, int num_tiles, hipStream_t s) {
  const RasterParams<T> rp = make_raster_params<T>(w, h, cfg, row_begin);
  const dim3 grid((unsigned)num_tiles), block(TS * TS);
#define MS_BWD(AA, HEUR)                                                                           \
  raster_bwd_kernel<T, F, TS, AA, HEUR><<<grid, block, 0, s>>>(                                    \
      (const T*)points, (const T*)feats, ranges, o2p, (const T*)image, (const T*)grad_image, rp,   \
      (T*)gp, (T*)gf, (T*)heur)
  const bool aa = cfg->antialias, hf = cfg->compute_point_heuristic && heur;
  if (aa) { if (hf) MS_BWD(true, true); else MS_BWD(true, false); }
  else { if (hf) MS_BWD(false, true); else MS_BWD(false, false); }
#undef MS_BWD
}

template <typename T, int F, typename... Args>
static int dispatch_ts_fwd(int ts, Args... args) {
  switch (ts) {
    case 8: launch_fwd<T, F, 8>(args...); return 0;
    case 16: launch_fwd<T, F, 16>(args...); return 0;
    case 32: launch_fwd<T, F, 32>(args...); return 0;
  }
  return MS_ERR_UNSUPPORTED;
}

template <typename T, int F, typename... Args>
static int dispatch_ts_bwd(int ts, Args... args) {
  switch (ts) {
    case 8: launch_bwd<T, F, 8>(args...); return 0;
    case 16: launch_bwd<T, F, 16>(args...); return 0;
    case 32: launch_bwd<T, F, 32>(args...); return 0;
  }
  return MS_ERR_UNSUPPORTED;
}

}  // namespace ms

using namespace ms;

static int check_raster_common(const ms_raster_config* cfg, int w, int h, int f, int dtype, int* row_begin,
                               int* row_end, const char* fn) {
  if (!cfg) { set_error("%s: cfg is null", fn); return MS_ERR_BAD_ARG; }
  if (w <= 0 || h <= 0) { set_error("%s: bad image size %dx%d", fn, w, h); return MS_ERR_BAD_ARG; }
  if (dtype != MS_F32 && dtype != MS_F64) { set_error("%s: dtype must be MS_F32 or MS_F64", fn); return MS_ERR_BAD_ARG; }
  if (cfg->tile_size != 8 && cfg->tile_size != 16 && cfg->tile_size != 32) {
    set_error("%s: tile_size must be 8, 16 or 32 (got %d)", fn, cfg->tile_size); return MS_ERR_UNSUPPORTED;
  }
  if (f < 1 || f > RASTER_MAX_F) {
    set_error("%s: feature size %d not in [1, %d] (split the channels on the host)", fn, f, RASTER_MAX_F);
    return MS_ERR_UNSUPPORTED;
  }
  const int tiles_high = (h + cfg->tile_size - 1) / cfg->tile_size;
  if (*row_begin < 0) *row_begin = 0;
  if (*row_end > tiles_high) *row_end = tiles_high;
  return 0;
}

extern "C" int ms_raster_fwd(const void* points7, const void* features, const int32_t* tile_ranges,
                             const int32_t* overlap_to_point, int image_w, int image_h, int f,
                             const ms_raster_config* cfg, void* out_image, void* out_alpha,
                             void* out_visibility, int tile_row_begin, int tile_row_end, int dtype,
                             void* stream) {
  int rc = check_raster_common(cfg, image_w, image_h, f, dtype, &tile_row_begin, &tile_row_end, "ms_raster_fwd");
  if (rc) return rc;
  MS_CHECK_ARG(tile_ranges && out_image && out_alpha, "null pointer");
  if (tile_row_end <= tile_row_begin) return 0;
  const int tiles_wide = (image_w + cfg->tile_size - 1) / cfg->tile_size;
  const int num_tiles = (tile_row_end - tile_row_begin) * tiles_wide;
  hipStream_t s = (hipStream_t)stream;
#define MS_GO(T, F) rc = dispatch_ts_fwd<T, F>(cfg->tile_size, points7, features, tile_ranges, overlap_to_point, image_w, image_h, cfg, out_image, out_alpha, out_visibility, tile_row_begin, num_tiles, s)
  if (dtype == MS_F32) {
    switch (f) { case 1: MS_GO(float, 1); break; case 2: MS_GO(float, 2); break; case 3: MS_GO(float, 3); break; default: MS_GO(float, 4); break; }
  } else {
    switch (f) { case 1: MS_GO(double, 1); break; case 2: MS_GO(double, 2); break; case 3: MS_GO(double, 3); break; default: MS_GO(double, 4); break; }
  }
#undef MS_GO
  if (rc) return rc;
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_raster_bwd(const void* points7, const void* features, const int32_t* tile_ranges,
                             const int32_t* overlap_to_point, const void* image, const void* grad_image,
                             int image_w, int image_h, int f, const ms_raster_config* cfg,
                             void* grad_points7, void* grad_features, void* point_heuristic,
                             int tile_row_begin, int tile_row_end, int dtype, void* stream) {
  int rc = check_raster_common(cfg, image_w, image_h, f, dtype, &tile_row_begin, &tile_row_end, "ms_raster_bwd");
  if (rc) return rc;
  MS_CHECK_ARG(tile_ranges && image && grad_image, "null pointer");
  MS_CHECK_ARG(cfg->use_alpha_blending, "backward requires use_alpha_blending (reference: tests/test_rasterizer.py:92-94)");
  if (tile_row_end <= tile_row_begin) return 0;
  if (!grad_points7 && !grad_features && !point_heuristic) return 0;
  const int tiles_wide = (image_w + cfg->tile_size - 1) / cfg->tile_size;
  const int num_tiles = (tile_row_end - tile_row_begin) * tiles_wide;
  hipStream_t s = (hipStream_t)stream;
#define MS_GO(T, F) rc = dispatch_ts_bwd<T, F>(cfg->tile_size, points7, features, tile_ranges, overlap_to_point, image, grad_image, image_w, image_h, cfg, grad_points7, grad_features, point_heuristic, tile_row_begin, num_tiles, s)
  if (dtype == MS_F32) {
    switch (f) { case 1: MS_GO(float, 1); break; case 2: MS_GO(float, 2); break; case 3: MS_GO(float, 3); break; default: MS_GO(float, 4); break; }
  } else {
    switch (f) { case 1: MS_GO(double, 1); break; case 2: MS_GO(double, 2); break; case 3: MS_GO(double, 3); break; default: MS_GO(double, 4); break; }
  }
#undef MS_GO
  if (rc) return rc;
  MS_CHECK_LAUNCH();
  return 0;
}
