// raster.hip — tile-based alpha compositing of sorted 2D gaussians: generic forward / backward
// templates (float and double, F = 1..4 channels, tile 8/16/32, plain or antialiased pdf, blending or
// quantile render, visibility, point heuristics) and the C-ABI dispatch.  The product path (float, RGB,
// plain pdf, blending) is dispatched to the hand-tuned kernels of raster_fast.hip; everything else —
// in particular the float64 instantiations the reference-style gradcheck tests run — uses the
// templates below.
//
// Work decomposition (gfx950, wave64), shared with raster_fast.hip:
//   * one workgroup per screen tile (tile_size^2 threads), one wavefront per 8x8 pixel patch,
//     one lane per pixel, so the per-gaussian gradient sum of the backward pass is a pure in-wave
//     reduction — no LDS atomics, no cross-wave combine (the reference needs a 32-lane shuffle
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
#include <stdlib.h>
#include <string.h>

#include "raster_common.h"
#include "frame_internal.h"

namespace ms {

constexpr int RASTER_MAX_F = 4;          // forward, and backward without point heuristics: wider features are chunked on the host
constexpr int RASTER_MAX_F_HEUR = 16;    // backward WITH point heuristics: F = 8 and 16 are instantiated too (see ms_raster_bwd)
// number of per-pixel gradient values a backward pass reduces per splat, padded to the butterfly's 16 slots
template <int F> struct GradSlots { static constexpr int N = F <= 4 ? 16 : 7 + F + 2; };

template <typename T> struct RasterParams {
  int width, height, tiles_wide, tile_begin;   // tile_begin = first tile id of the strip
  T clamp_max_alpha, alpha_threshold, saturate_threshold;
};

// staged splat: AoS so the wave-uniform reads in the hot loop are a few wide broadcast LDS reads.
// The pdf is evaluated in the splat's normalised frame:  X = dx*A + dy*B,  Y = dx*C + dy*D  with
// A = ax/sx, B = ay/sx, C = -ay/sy, D = ax/sy  (so X = d.axis/sigma_x, Y = d.perp(axis)/sigma_y,
// taichi_lib/generic.py:311-317), g = exp(-(X^2 + Y^2)/2).
template <typename T, int F> struct alignas(16) Splat {
  T mx, my, A, B;
  T C, D, isx, isy;
  T alpha;
  T f[F];
  T ax, ay, sx, sy;       // only read by the antialiased pdf
};
template <typename T> struct CullBox { T cx, cy, ex, ey; };

// exp(-r2 / 2) as ONE v_exp_f32 (2^x) after a single multiply; ~1 ulp, far inside the 1e-4 budget
__device__ __forceinline__ float gauss_exp(float r2) { return __builtin_amdgcn_exp2f(r2 * -0.72134752044448170368f); }
__device__ __forceinline__ double gauss_exp(double r2) { return exp(-0.5 * r2); }
// 1 / x as ONE v_rcp_f32 (1 ulp)
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double fast_rcp(double x) { return 1.0 / x; }

template <typename T, int F, int BATCH>
__device__ __forceinline__ void stage_batch(const T* __restrict__ points, const T* __restrict__ feats,
                                            const int32_t* __restrict__ o2p, int begin, int count,
                                            T alpha_threshold, bool antialias, Splat<T, F>* s_splat,
                                            CullBox<T>* s_cull, int32_t* s_id) {
  for (int t = threadIdx.x; t < count; t += blockDim.x) {
    const int32_t id = o2p[begin + t];
    const T* g = points + (int64_t)id * 7;
    Splat<T, F> s;
    s.mx = g[0]; s.my = g[1]; s.ax = g[2]; s.ay = g[3];
    s.sx = g[4]; s.sy = g[5]; s.alpha = g[6];
    s.isx = T(1) / s.sx; s.isy = T(1) / s.sy;
    s.A = s.ax * s.isx; s.B = s.ay * s.isx;
    s.C = -s.ay * s.isy; s.D = s.ax * s.isy;
#pragma unroll
    for (int c = 0; c < F; ++c) s.f[c] = feats[(int64_t)id * F + c];
    s_splat[t] = s;
    s_id[t] = id;

    CullBox<T> b;
    b.cx = s.mx; b.cy = s.my;
    if (!antialias) {
      // half extents of the bounding box of the ellipse  alpha * g == threshold; NaN (alpha below
      // the threshold) fails every comparison below and culls the splat, which cannot contribute
      const T gs = t_sqrt(2 * t_log(s.alpha / alpha_threshold));
      const T v1x = s.ax * s.sx * gs, v1y = s.ay * s.sx * gs;
      const T v2x = -s.ay * s.sy * gs, v2y = s.ax * s.sy * gs;
      b.ex = t_sqrt(v1x * v1x + v2x * v2x) * T(1.001) + T(0.01);
      b.ey = t_sqrt(v1y * v1y + v2y * v2y) * T(1.001) + T(0.01);
    } else {
      // antialiased pdf (generic.py:341-357): p = 2 pi ix iy, ix = sx (S((tx + .5)/sx) - S((tx - .5)/sx)),
      // S(z) = 1 / (1 + exp(-h(z))), h = 1.6 z + 0.07 z^3.  For |tx| >= .5: ix <= sx (1 - S(z)) <= sx exp(-h(z))
      // with z = (|tx| - .5) / sx, and iy <= sy.  So alpha p > threshold needs exp(-h(zx)) > q / (sx sy),
      // q = threshold / (2 pi alpha), i.e. h(zx) < L = ln(sx sy / q): |tx| < .5 + sx zx*, likewise |ty|; the
      // contribution region lies inside that oriented rectangle.  h is convex and increasing for z >= 0, so
      // Newton from z0 = L / 1.6 >= root descends monotonically and every iterate bounds the root from above.
      const T q = alpha_threshold / (T(6.283185307179586) * s.alpha);
      const T L = t_log(s.sx * s.sy / q);
      if (!(L > T(0))) {
        b.ex = T(-1e30); b.ey = T(-1e30);      // alpha * p <= threshold everywhere (also NaN / alpha <= 0)
      } else {
        T z = L / T(1.6);
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const T h = T(1.6) * z + T(0.07) * z * z * z;
          z -= (h - L) / (T(1.6) + T(0.21) * z * z);
        }
        z = z * T(1.001) + T(0.001);
        const T rx = T(0.5) + s.sx * z, ry = T(0.5) + s.sy * z;
        b.ex = t_abs(s.ax) * rx + t_abs(s.ay) * ry + T(0.01);
        b.ey = t_abs(s.ay) * rx + t_abs(s.ax) * ry + T(0.01);
      }
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

// ---- antialiased pdf, float: v_exp_f32 / v_rcp_f32 and the staged 1/sigma instead of expf and divisions
// (generic.py:341-404; the double instantiation keeps the exact formulation of splat_math.h)
__device__ __forceinline__ float aa_sigmoid(float x, float inv_sigma, float& z) {
  z = x * inv_sigma;
  // exp(-(1.6 z + 0.07 z^3)) = exp2(z (-1.6 log2e - 0.07 log2e z^2))
  const float e = __builtin_amdgcn_exp2f(z * (-2.30831206544f - 0.100988652863f * z * z));
  return __builtin_amdgcn_rcpf(1.0f + e);
}

template <int F>
__device__ __forceinline__ float aa_pdf(const Splat<float, F>& s, float px, float py) {
  const float dx = px - s.mx, dy = py - s.my;
  const float tx = dx * s.ax + dy * s.ay, ty = dy * s.ax - dx * s.ay;
  float z;
  const float ix = aa_sigmoid(tx + 0.5f, s.isx, z) - aa_sigmoid(tx - 0.5f, s.isx, z);
  const float iy = aa_sigmoid(ty + 0.5f, s.isy, z) - aa_sigmoid(ty - 0.5f, s.isy, z);
  return 6.283185307179586f * (s.sx * ix) * (s.sy * iy);
}

template <int F>
__device__ __forceinline__ double aa_pdf(const Splat<double, F>& s, double px, double py) {
  const double g[6] = {s.mx, s.my, s.ax, s.ay, s.sx, s.sy};
  return gaussian_pdf_antialias(px, py, g);
}

template <int F>
__device__ __forceinline__ float aa_pdf_with_grad(const Splat<float, F>& s, float px, float py, float gm[2],
                                                  float ga[2], float gs[2]) {
  const float dx = px - s.mx, dy = py - s.my;
  const float tx = dx * s.ax + dy * s.ay, ty = dy * s.ax - dx * s.ay;
  // S, dS/dx and dS/dsigma at the four cell edges (generic.py:360-368)
  float S[4], dS[4], dSs[4];
  const float xs[4] = {tx + 0.5f, tx - 0.5f, ty + 0.5f, ty - 0.5f};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float inv = k < 2 ? s.isx : s.isy;
    float z;
    S[k] = aa_sigmoid(xs[k], inv, z);
    const float d = (1.6f + 0.21f * z * z) * S[k] * (1.0f - S[k]);
    dS[k] = d * inv;
    dSs[k] = -dS[k] * z;
  }
  const float tau = 6.283185307179586f;
  const float ix = s.sx * (S[0] - S[1]), iy = s.sy * (S[2] - S[3]);
  const float dSx = iy * s.sx * (dS[0] - dS[1]), dSy = ix * s.sy * (dS[2] - dS[3]);
  gm[0] = tau * (dSy * s.ay - dSx * s.ax);
  gm[1] = -tau * (dSx * s.ay + dSy * s.ax);
  gs[0] = tau * iy * (S[0] - S[1] + (dSs[0] - dSs[1]) * s.sx);
  gs[1] = tau * ix * (S[2] - S[3] + (dSs[2] - dSs[3]) * s.sy);
  ga[0] = tau * (dSx * dx + dSy * dy);
  ga[1] = tau * (dSx * dy - dSy * dx);
  return tau * ix * iy;
}

template <int F>
__device__ __forceinline__ double aa_pdf_with_grad(const Splat<double, F>& s, double px, double py, double gm[2],
                                                   double ga[2], double gs[2]) {
  const double g[6] = {s.mx, s.my, s.ax, s.ay, s.sx, s.sy};
  return gaussian_pdf_antialias_with_grad(px, py, g, gm, ga, gs);
}

template <typename T, int F, bool AA>
__device__ __forceinline__ T splat_pdf(const Splat<T, F>& s, T px, T py) {
  if (AA) {
    return aa_pdf<F>(s, px, py);
  } else {
    const T dx = px - s.mx, dy = py - s.my;
    const T X = dx * s.A + dy * s.B;
    const T Y = dx * s.C + dy * s.D;
    return gauss_exp(X * X + Y * Y);
  }
}

// Gradient commit of the generic kernels.  float: the halving butterfly of raster_common.h
// (wave_reduce16: the NV totals end in NV distinct lanes, ONE global_atomic_add_f32 commits them);
// double (test-only path): plain ds_bpermute butterflies + one atomic per value from lane 63.

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
    stage_batch<T, F, BATCH>(points, feats, o2p, begin, count, rp.alpha_threshold, AA, s_splat, s_cull, s_id);
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
          } else {
            // quantile render (forward.py:102-112): the weight keeps accumulating for every gated splat (it is what
            // `visibility` sums in this mode too, forward.py:114-126); the pixel takes the feature of the splat at which
            // the accumulated weight FIRST reaches 1 - saturate_threshold
            weight = alpha * (T(1) - W);
            W += weight;
            if (!saturated && W >= T(1) - rp.saturate_threshold) {
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

// destination word of the value a lane holds after wave_reduce16: value k < 7 -> grad_points[id][k],
// k < 7+F -> grad_feats[id][k-7], then the two heuristics
template <typename T, int F, bool HEUR>
__device__ __forceinline__ void grad_target(int lane, T* grad_points, T* grad_feats, T* heuristic,
                                            T*& base, int& stride) {
  base = nullptr; stride = 0;
  if ((lane & 15) < 12) return;
  const int k = butterfly_slot(lane);
  if (k < 7) { if (grad_points) { base = grad_points + k; stride = 7; } }
  else if (k < 7 + F) { if (grad_feats) { base = grad_feats + (k - 7); stride = F; } }
  else if (HEUR && k < 7 + F + 2) { if (heuristic) { base = heuristic + (k - 7 - F); stride = 2; } }
}

// General form (double, and float with more than 4 channels): one tree reduction + one atomic per value.
// The float instantiations for F <= 4 — everything the product path launches — are specialised below with the
// halving butterfly (one atomic instruction per splat).
template <typename T, int F, bool HEUR>
__device__ __forceinline__ void commit_gradients(const T (&v)[GradSlots<F>::N], int32_t id, int lane, T* tgt_base,
                                                 int tgt_stride, T* gp, T* gf, T* heur) {
  for (int k = 0; k < 7 + F + (HEUR ? 2 : 0); ++k) {
    const T total = wave_sum_to_lane63(v[k]);
    if (lane == 63) {
      if (k < 7) { if (gp) atomic_add_noret(gp + (int64_t)id * 7 + k, total); }
      else if (k < 7 + F) { if (gf) atomic_add_noret(gf + (int64_t)id * F + (k - 7), total); }
      else if (heur) atomic_add_noret(heur + (int64_t)id * 2 + (k - 7 - F), total);
    }
  }
}

template <>
__device__ __forceinline__ void commit_gradients<float, 1, false>(const float (&v)[16], int32_t id, int lane, float* b, int st, float*, float*, float*) {
  const float total = wave_reduce16(v, lane & 1, lane & 2);
  if (b) atomic_add_noret(b + (int64_t)id * st, total);
}
#define MS_COMMIT_F32(F, HEUR)                                                                                   \
  template <>                                                                                                    \
  __device__ __forceinline__ void commit_gradients<float, F, HEUR>(const float (&v)[16], int32_t id, int lane,   \
                                                                   float* b, int st, float*, float*, float*) {   \
    const float total = wave_reduce16(v, lane & 1, lane & 2);                                                                  \
    if (b) atomic_add_noret(b + (int64_t)id * st, total);                                                        \
  }
MS_COMMIT_F32(1, true) MS_COMMIT_F32(2, false) MS_COMMIT_F32(2, true) MS_COMMIT_F32(3, false)
MS_COMMIT_F32(3, true) MS_COMMIT_F32(4, false) MS_COMMIT_F32(4, true)
#undef MS_COMMIT_F32

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
  constexpr int BATCH_MAX = F > 4 ? 128 : 256;       // wide records: keep the staging arrays inside 64 KB of LDS
  constexpr int BATCH = THREADS < BATCH_MAX ? THREADS : BATCH_MAX;
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

  // float path: which output word this lane commits after the butterfly (see wave_reduce16)
  T* tgt_base = nullptr;
  int tgt_stride = 0;
  grad_target<T, F, HEUR>(lane, grad_points, grad_feats, heuristic, tgt_base, tgt_stride);

  for (int begin = start; begin < end; begin += BATCH) {
    const int count = (end - begin) < BATCH ? (end - begin) : BATCH;
    // tile-wide early out once every pixel is saturated (backward.py:116)
    if (__syncthreads_and(W >= rp.saturate_threshold)) break;
    stage_batch<T, F, BATCH>(points, feats, o2p, begin, count, rp.alpha_threshold, AA, s_splat, s_cull, s_id);
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

        // per-pixel partial gradients of the packed 2D gaussian, before the aag factor:
        //   gm = dp/dmean, ga = dp/daxis, gs = dp/dsigma   (generic.py:321-336 / :371-404)
        T gm[2], ga[2], gs[2];
        T p;
        if (AA) {
          p = aa_pdf_with_grad<F>(s, px, py, gm, ga, gs);
        } else {
          const T dx = px - s.mx, dy = py - s.my;
          const T X = dx * s.A + dy * s.B;
          const T Y = dx * s.C + dy * s.D;
          p = gauss_exp(X * X + Y * Y);
          const T pX = p * X, pY = p * Y;
          gm[0] = pX * s.A + pY * s.C;           // p (X/sx * axis + Y/sy * perp(axis))
          gm[1] = pX * s.B + pY * s.D;
          const T u = pX * s.isx, w = pY * s.isy;
          ga[0] = -(u * dx + w * dy);            // p (X/sx * -d + Y/sy * perp(d))
          ga[1] = w * dx - u * dy;
          gs[0] = u * X;                         // p (X^2 / sx, Y^2 / sy)
          gs[1] = w * Y;
        }

        const T alpha_raw = s.alpha * p;
        const bool active = alpha_raw > rp.alpha_threshold && W < rp.saturate_threshold;
        const T alpha = t_min(alpha_raw, rp.clamp_max_alpha);
        const T Ti = T(1) - W;
        const T weight = active ? alpha * Ti : T(0);
        W += weight;
        const T inv = fast_rcp(T(1) - alpha);
        T alpha_grad = T(0);
#pragma unroll
        for (int c = 0; c < F; ++c) {
          R[c] -= s.f[c] * weight;
          alpha_grad += (s.f[c] * Ti - R[c] * inv) * G[c];
        }
        alpha_grad = active ? alpha_grad : T(0);
        const T aag = s.alpha * alpha_grad;      // straight-through clamp (backward.py:158-163)

        if (__ballot(active) == 0) continue;     // no pixel of this patch contributes

        constexpr int NV = 7 + F + (HEUR ? 2 : 0);
        T v[GradSlots<F>::N];
        v[0] = aag * gm[0]; v[1] = aag * gm[1];
        v[2] = aag * ga[0]; v[3] = aag * ga[1];
        v[4] = aag * gs[0]; v[5] = aag * gs[1];
        v[6] = p * alpha_grad;
#pragma unroll
        for (int c = 0; c < F; ++c) v[7 + c] = weight * G[c];
        if (HEUR) {
          v[7 + F] = aag * aag;
          v[7 + F + 1] = t_abs(v[0]) + t_abs(v[1]);
        }
#pragma unroll
        for (int k = NV; k < GradSlots<F>::N; ++k) v[k] = T(0);

        const int32_t id = s_id[r + b];
        commit_gradients<T, F, HEUR>(v, id, lane, tgt_base, tgt_stride, grad_points, grad_feats, heuristic);
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
  if (!blend) {
    if (aa) { if (visf) MS_FWD(true, false, true); else MS_FWD(true, false, false); }
    else { if (visf) MS_FWD(false, false, true); else MS_FWD(false, false, false); }
  }
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

// F = 8 / 16: backward with point heuristics only (prune_cost and split_score are not linear in the channels, so
// they cannot be assembled from 4-channel chunks: backward.py:171-194 forms alpha_grad over ALL channels first)
template <typename T, int F, int TS>
static void launch_bwd_wide(const void* points, const void* feats, const int32_t* ranges, const int32_t* o2p,
                            const void* image, const void* grad_image, int w, int h, const ms_raster_config* cfg,
                            void* gp, void* gf, void* heur, int row_begin, int num_tiles, hipStream_t s) {
  const RasterParams<T> rp = make_raster_params<T>(w, h, cfg, row_begin);
  const dim3 grid((unsigned)num_tiles), block(TS * TS);
  if (cfg->antialias)
    raster_bwd_kernel<T, F, TS, true, true><<<grid, block, 0, s>>>((const T*)points, (const T*)feats, ranges, o2p, (const T*)image,
                                                                   (const T*)grad_image, rp, (T*)gp, (T*)gf, (T*)heur);
  else
    raster_bwd_kernel<T, F, TS, false, true><<<grid, block, 0, s>>>((const T*)points, (const T*)feats, ranges, o2p, (const T*)image,
                                                                    (const T*)grad_image, rp, (T*)gp, (T*)gf, (T*)heur);
}

template <typename T, int F, typename... Args>
static int dispatch_ts_bwd_wide(int ts, Args... args) {
  switch (ts) {
    case 8: launch_bwd_wide<T, F, 8>(args...); return 0;
    case 16: launch_bwd_wide<T, F, 16>(args...); return 0;
    case 32: launch_bwd_wide<T, F, 32>(args...); return 0;
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

// product-path kernels (float, F = 3, plain pdf, blending): raster_fast.hip
bool ms_raster_fwd_fast(const void* points, const void* feats, const int32_t* ranges, const int32_t* o2p,
                        int w, int h, const ms_raster_config* cfg, void* image, void* alpha, void* visibility,
                        int row_begin, int num_tiles, hipStream_t s, const float* splat_rows, const SplitScratch* split,
                        int32_t* long_run_word);
bool ms_raster_bwd_fast(const void* points, const void* feats, const int32_t* ranges, const int32_t* o2p,
                        const void* image, const void* grad_image, int w, int h, const ms_raster_config* cfg,
                        void* gp, void* gf, void* heur, int row_begin, int num_tiles, hipStream_t s);

static int check_raster_common(const ms_raster_config* cfg, int w, int h, int f, int dtype, int* row_begin,
                               int* row_end, const char* fn, int max_f = RASTER_MAX_F) {
  if (!cfg) { set_error("%s: cfg is null", fn); return MS_ERR_BAD_ARG; }
  if (w <= 0 || h <= 0) { set_error("%s: bad image size %dx%d", fn, w, h); return MS_ERR_BAD_ARG; }
  if (dtype != MS_F32 && dtype != MS_F64) { set_error("%s: dtype must be MS_F32 or MS_F64", fn); return MS_ERR_BAD_ARG; }
  if (cfg->tile_size != 8 && cfg->tile_size != 16 && cfg->tile_size != 32) {
    set_error("%s: tile_size must be 8, 16 or 32 (got %d)", fn, cfg->tile_size); return MS_ERR_UNSUPPORTED;
  }
  if (f < 1 || f > max_f || (f > RASTER_MAX_F && f != 8 && f != 16)) {
    set_error("%s: feature size %d not supported: 1..%d, or 8 / 16 for the backward pass with point heuristics "
              "(split or zero-pad the channels on the host)", fn, f, RASTER_MAX_F);
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
  return raster_fwd_launch(points7, features, nullptr, tile_ranges, overlap_to_point, image_w, image_h, f, cfg, out_image,
                           out_alpha, out_visibility, tile_row_begin, tile_row_end, dtype, stream);
}

extern "C" size_t ms_raster_split_scratch_bytes(int64_t k_capacity, int tile_size, int split_min_run, int split_seg_len) {
  if (k_capacity < 0 || (tile_size != 8 && tile_size != 16 && tile_size != 32) || split_min_run < 0 || split_seg_len < 0) return 0;
  return split_scratch_bytes(k_capacity, tile_size, split_params(tile_size, split_min_run, split_seg_len));
}

extern "C" int ms_raster_fwd_split(const float* points7, const float* features, const int32_t* tile_ranges,
                                   const int32_t* overlap_to_point, int64_t k_capacity, int image_w, int image_h,
                                   const ms_raster_config* cfg, float* out_image, float* out_alpha,
                                   float* out_visibility, void* split_scratch, int split_min_run, int split_seg_len,
                                   int tile_row_begin, int tile_row_end, void* stream) {
  MS_CHECK_ARG(cfg && split_scratch && k_capacity >= 0 && k_capacity < (1ll << 31), "null pointer / capacity outside [0, 2^31)");
  MS_CHECK_ARG(split_min_run >= 0 && split_seg_len >= 0, "negative split parameter");
  MS_CHECK_ARG((reinterpret_cast<uintptr_t>(split_scratch) & 255) == 0, "split_scratch must be 256-byte aligned");
  if (!raster_uses_splat_rows(cfg, 3, MS_F32)) {
    set_error("ms_raster_fwd_split: float32 RGB, plain pdf, alpha blending");
    return MS_ERR_UNSUPPORTED;
  }
  const SplitScratch sc = split_scratch_carve(split_scratch, k_capacity, cfg->tile_size,
                                              split_params(cfg->tile_size, split_min_run, split_seg_len));
  return raster_fwd_launch(points7, features, nullptr, tile_ranges, overlap_to_point, image_w, image_h, 3, cfg, out_image,
                           out_alpha, out_visibility, tile_row_begin, tile_row_end, MS_F32, stream, &sc, nullptr);
}

namespace ms {
__global__ void __launch_bounds__(256)
splat_rows_pack_kernel(const float* __restrict__ points, const float* __restrict__ depth, const float* __restrict__ colours,
                       int64_t n, float* __restrict__ rows) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* g = points + i * 7;
  float4* row = reinterpret_cast<float4*>(rows + i * SPLAT_ROW);
  row[0] = float4{g[0], g[1], g[2], g[3]};
  row[1] = float4{g[4], g[5], g[6], depth ? depth[i] : 0.0f};
  row[2] = float4{colours[i * 3 + 0], colours[i * 3 + 1], colours[i * 3 + 2], 0.0f};
}
}  // namespace ms

static_assert(MS_SPLAT_ROW == ms::SPLAT_ROW, "include/mi355_splat.h and common.h disagree on the splat row");

extern "C" int ms_splat_rows_pack(const float* points7, const float* depth, const float* colours3, int64_t n, float* rows,
                                  void* stream) {
  MS_CHECK_ARG(n >= 0, "n < 0");
  if (n == 0) return 0;
  MS_CHECK_ARG(points7 && colours3 && rows, "null pointer");
  MS_CHECK_ARG((reinterpret_cast<uintptr_t>(rows) & 63) == 0, "rows must be 64-byte aligned");
  splat_rows_pack_kernel<<<dim3((unsigned)div_up(n, 256)), dim3(256), 0, (hipStream_t)stream>>>(points7, depth, colours3, n, rows);
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_raster_fwd_rows(const float* rows, const int32_t* tile_ranges, const int32_t* overlap_to_point,
                                  int image_w, int image_h, const ms_raster_config* cfg, float* out_image,
                                  float* out_alpha, float* out_visibility, int tile_row_begin, int tile_row_end,
                                  void* stream) {
  MS_CHECK_ARG(cfg && rows, "null pointer");
  if (!raster_uses_splat_rows(cfg, 3, MS_F32)) {
    set_error("ms_raster_fwd_rows: splat rows serve float32 RGB, plain pdf, alpha blending, tile 8 / 16 / 32");
    return MS_ERR_UNSUPPORTED;
  }
  return raster_fwd_launch(rows, rows, rows, tile_ranges, overlap_to_point, image_w, image_h, 3, cfg, out_image, out_alpha,
                           out_visibility, tile_row_begin, tile_row_end, MS_F32, stream);
}

bool ms::raster_uses_splat_rows(const ms_raster_config* cfg, int f, int dtype) {
  return dtype == MS_F32 && f == 3 && !cfg->antialias && cfg->use_alpha_blending &&
         (cfg->tile_size == 8 || cfg->tile_size == 16 || cfg->tile_size == 32);
}

// ms_raster_fwd with the frame executor's splat-row table (frame_internal.h): the product kernels gather from it when
// the configuration is theirs (raster_uses_splat_rows), everything else reads the dense arrays as before
int ms::raster_fwd_launch(const void* points7, const void* features, const float* splat_rows, const int32_t* tile_ranges,
                      const int32_t* overlap_to_point, int image_w, int image_h, int f, const ms_raster_config* cfg,
                      void* out_image, void* out_alpha, void* out_visibility, int tile_row_begin, int tile_row_end,
                      int dtype, void* stream, const SplitScratch* split, int32_t* long_run_word) {
  int rc = check_raster_common(cfg, image_w, image_h, f, dtype, &tile_row_begin, &tile_row_end, "ms_raster_fwd");
  if (rc) return rc;
  MS_CHECK_ARG(tile_ranges && out_image && out_alpha, "null pointer");
  if (tile_row_end <= tile_row_begin) return 0;
  const int tiles_wide = (image_w + cfg->tile_size - 1) / cfg->tile_size;
  const int num_tiles = (tile_row_end - tile_row_begin) * tiles_wide;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MS_F32 && f == 3 && !cfg->antialias && cfg->use_alpha_blending) {
    void* vis = (cfg->compute_visibility && out_visibility) ? out_visibility : nullptr;
    if (ms_raster_fwd_fast(points7, features, tile_ranges, overlap_to_point, image_w, image_h, cfg, out_image,
                           out_alpha, vis, tile_row_begin, num_tiles, s, splat_rows, split, long_run_word)) {
      MS_CHECK_LAUNCH();
      return 0;
    }
  }
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
  const bool wide = cfg && cfg->compute_point_heuristic && point_heuristic;
  int rc = check_raster_common(cfg, image_w, image_h, f, dtype, &tile_row_begin, &tile_row_end, "ms_raster_bwd",
                               wide ? RASTER_MAX_F_HEUR : RASTER_MAX_F);
  if (rc) return rc;
  MS_CHECK_ARG(tile_ranges && image && grad_image, "null pointer");
  MS_CHECK_ARG(cfg->use_alpha_blending, "backward requires use_alpha_blending (reference: tests/test_rasterizer.py:92-94)");
  if (tile_row_end <= tile_row_begin) return 0;
  if (!grad_points7 && !grad_features && !point_heuristic) return 0;
  const int tiles_wide = (image_w + cfg->tile_size - 1) / cfg->tile_size;
  const int num_tiles = (tile_row_end - tile_row_begin) * tiles_wide;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == MS_F32 && f == 3 && !cfg->antialias) {
    if (ms_raster_bwd_fast(points7, features, tile_ranges, overlap_to_point, image, grad_image, image_w, image_h,
                           cfg, grad_points7, grad_features, point_heuristic, tile_row_begin, num_tiles, s)) {
      MS_CHECK_LAUNCH();
      return 0;
    }
  }
  if (f > RASTER_MAX_F) {
#define MS_WIDE(T, F) rc = dispatch_ts_bwd_wide<T, F>(cfg->tile_size, points7, features, tile_ranges, overlap_to_point, image, grad_image, image_w, image_h, cfg, grad_points7, grad_features, point_heuristic, tile_row_begin, num_tiles, s)
    if (dtype == MS_F32) { if (f == 8) MS_WIDE(float, 8); else MS_WIDE(float, 16); }
    else { if (f == 8) MS_WIDE(double, 8); else MS_WIDE(double, 16); }
#undef MS_WIDE
    if (rc) return rc;
    MS_CHECK_LAUNCH();
    return 0;
  }
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
