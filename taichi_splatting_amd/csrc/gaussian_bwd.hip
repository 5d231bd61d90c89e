// gaussian_bwd.hip — the whole per-gaussian backward of a frame in ONE pass over the gaussians:
//
//   raster-backward moment rows (raster_bwd_scan.hip)  ->  d(packed 2D gaussian), d(colour)       [finalize]
//   d(packed 2D gaussian), d(depth)                    ->  d(position, log_scaling, rotation, alpha_logit, camera)
//   d(colour) through the clamp mask                   ->  d(SH parameters)
//
// replacing three launches (raster_moments_finalize_kernel, project_bwd_kernel, sh_bwd_params_kernel) that each
// re-read the packed splats / wrote and re-read the 40-byte 2D-boundary gradients, plus the 64 B-per-gaussian zero
// fill of the moments buffer: the pass re-zeroes the rows it reads, so the buffer stays clean from frame to frame.
// Reference counterparts: indexed_project_kernel.grad (perspective/projection.py:167-188), evaluate_sh_at_kernel.grad
// (indexed_spherical_harmonics.py:153-160) and the per-pixel gradient formulas of taichi_lib/generic.py:321-336.
//
// The frame executor does not compact: all n gaussians are visited in place, the culled ones (forward depth <= 0)
// write zero rows.  The forward projection is recomputed (as project_bwd_kernel does), so the packed 2D gaussian is
// not read back either.  Same FP contraction setting as projection.hip: the recomputed forward is bit-identical.
#pragma clang fp contract(off)
#include "raster_common.h"
#include "frame_internal.h"

#ifndef MS_GB_NT
#define MS_GB_NT 1
#endif

namespace ms {

// gradient rows are written once and read by somebody else much later (the optimiser): streaming stores
template <typename V>
__device__ __forceinline__ void stream_store(V* p, V v) {
#if MS_GB_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

constexpr int GB_MAX_F = 4;      // SH colour channels (sh.hip: SH_MAX_F)

template <typename T> struct GaussBwdDev {
  int64_t n;
  const T *position, *log_scaling, *rotation, *alpha_logit, *Tcw, *proj;
  ProjParams<T> pp;
  const T* depth;
  void* moments;
  const int32_t* fixed_exp;
  const T *grad_points7, *grad_colours;
  int gp_stride, gc_stride;       // floats per row of grad_points7 / grad_colours (7 / f unless interleaved)
  int boundary_cov;               // given rows are [d mean | dL/d(a, b, c) | 0 | d alpha] (MS_BOUNDARY_COVARIANCE)
  int gather_world;               // > 0: rows gathered from the reverse exchange's receive buffer (frame_internal.h)
  const T* gather_rows;
  const int32_t *gather_slots, *gather_route;
  const T *extra_points7, *extra_depth, *extra_colours;
  int f;
  const T *camera_position, *colours;
  T *grad_position, *grad_log_scaling, *grad_rotation, *grad_alpha_logit, *grad_feature, *grad_camera;
  T *store_points7, *store_colours, *point_heuristic, *point_visibility;
};

// MOM: the 2D-boundary gradients come from the moment rows (float32, RGB); FIXED: rows of 64-bit fixed point.
// DEG >= 0: d(colour) -> d(SH parameters); DEG == -1: plain colours (grad_feature (n, 3) from the moment rows only —
// with gradient arrays the caller already holds d(colour)).
template <typename T, int DEG, bool MOM, bool FIXED>
__global__ void __launch_bounds__(256)
gaussian_bwd_kernel(const GaussBwdDev<T> a) {
  constexpr int D = DEG >= 0 ? (DEG + 1) * (DEG + 1) : 1;
  constexpr int YS = D + 1;                      // padded row stride: conflict-free per-lane writes
  __shared__ T s_Y[DEG >= 0 ? 4 : 1][DEG >= 0 ? 64 * YS : 1];
  __shared__ T s_g[DEG >= 0 ? 4 : 1][DEG >= 0 ? 64 * GB_MAX_F : 1];
  __shared__ T s_cam[4 * 16];
  // FIXED: the wave's 64 fixed-point rows (64 x 128 bytes, contiguous) come in as coalesced 16-byte loads, are turned
  // into floats on the way and handed to their lanes through LDS (round 5; a lane walking its own row with twelve
  // 8-byte loads and stores made this pass 0.43 -> 1.16 ms on config D)
  __shared__ float4 s_mom[(MOM && FIXED) ? 4 : 1][(MOM && FIXED) ? 64 * 3 : 1];

  const int wave = threadIdx.x >> 6, lane = lane_id();
  T cam_grad[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) cam_grad[k] = T(0);

  Camera<T> cam;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) cam.t[i][j] = a.Tcw[i * 4 + j];
  cam.fx = a.proj[0]; cam.fy = a.proj[1]; cam.cx = a.proj[2]; cam.cy = a.proj[3];

  // a wave takes 64 consecutive gaussians per iteration (the SH rows of a wave leave as coalesced stores)
  for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * 64; base < a.n; base += (int64_t)gridDim.x * 256) {
    const int64_t i = base + lane;
    const int count = (a.n - base) < 64 ? (int)(a.n - base) : 64;
    const bool valid = lane < count;
    const bool vis = valid && a.depth[i] > T(0);

    T gp[7] = {T(0), T(0), T(0), T(0), T(0), T(0), T(0)};
    T gcov[3] = {T(0), T(0), T(0)};          // MOM: dL/d(a, b, c) of the 2D covariance straight from the moments
    T gf[GB_MAX_F] = {T(0), T(0), T(0), T(0)};
    T dp[3] = {T(0), T(0), T(0)}, dls[3] = {T(0), T(0), T(0)}, dq[4] = {T(0), T(0), T(0), T(0)}, dal = T(0);
    T heur0 = T(0), heur1 = T(0), vis_sum = T(0);

    if constexpr (MOM && FIXED) {
      struct alignas(16) Pair { long long lo, hi; };
      Pair* blk = reinterpret_cast<Pair*>(reinterpret_cast<long long*>(a.moments) + base * MS_MOMENT_ROW);
      const double s_main = ldexp(1.0, -a.fixed_exp[0]), s_h0 = ldexp(1.0, -a.fixed_exp[1]);
      float* flat = reinterpret_cast<float*>(&s_mom[wave][0]);
      wave_lds_fence();
      // (two rounds of four loads in flight: eight kept 32 registers busy and the kernel at 172 VGPRs = two waves per
      // SIMD; with four it fits three like the float-row instantiation)
#pragma unroll 4
      for (int j = 0; j < 8; ++j) {
        const int idx = j * 64 + lane, row = idx >> 3, pair = idx & 7;      // values 2 pair, 2 pair + 1 of row `row`
        if (row < count) {
          const Pair v = blk[idx];
          blk[idx] = Pair{0, 0};                                             // whole lines go back as zeros
          if (pair < 6) {
            flat[row * 12 + 2 * pair] = (float)((double)v.lo * s_main);
            // value 9: its own unit; value 11 (blend-weight sum, heuristics rows only): units of 2^-32
            flat[row * 12 + 2 * pair + 1] = (float)((double)v.hi * (pair == 4 ? s_h0 : pair == 5 ? 0x1p-32 : s_main));
          }
        }
      }
      wave_lds_fence();
    }
    if (vis) {
      const T p[3] = {a.position[i * 3 + 0], a.position[i * 3 + 1], a.position[i * 3 + 2]};
      const T ls[3] = {a.log_scaling[i * 3 + 0], a.log_scaling[i * 3 + 1], a.log_scaling[i * 3 + 2]};
      const T q[4] = {a.rotation[i * 4 + 0], a.rotation[i * 4 + 1], a.rotation[i * 4 + 2], a.rotation[i * 4 + 3]};
      ProjState<T> st;
      project_forward(p, ls, q, a.alpha_logit[i], cam, a.pp, st);

      if constexpr (MOM) {
        // moments -> gradients of the packed 2D gaussian and its colour (raster_bwd_scan.hip, generic.py:321-336);
        // the row is re-zeroed for the next frame
        float4 r0, r1, r2;
        if constexpr (FIXED) {
          r0 = s_mom[wave][lane * 3 + 0]; r1 = s_mom[wave][lane * 3 + 1]; r2 = s_mom[wave][lane * 3 + 2];
        } else {
          float4* row = reinterpret_cast<float4*>(reinterpret_cast<float*>(a.moments) + i * MS_MOMENT_ROW);
          r0 = row[0]; r1 = row[1]; r2 = row[2];
          const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
          row[0] = z; row[1] = z; row[2] = z;
        }
        const float ax = (float)st.axis[0], ay = (float)st.axis[1], sx = (float)st.sigma[0], sy = (float)st.sigma[1];
        const float alpha = (float)st.alpha;
        const float isx = 1.0f / sx, isy = 1.0f / sy;
        const float A = ax * isx, B = ay * isx, C = -ay * isy, Dm = ax * isy;
        constexpr float IS = 1.0f / EXP2_BASIS_SCALE, IS2 = IS * IS;
        const float S = r0.x, Sx = r0.y * IS, Sy = r0.z * IS, Sxx = r0.w * IS2, Sxy = r1.x * IS2, Syy = r1.y * IS2;
        const float det = A * Dm - B * C;
        const float idet = det != 0.0f ? 1.0f / det : 0.0f;
        gp[0] = (T)(Sx * A + Sy * C);
        gp[1] = (T)(Sx * B + Sy * Dm);
        gp[2] = (T)(-(isx * (Dm * Sxx - B * Sxy) + isy * (A * Syy - C * Sxy)) * idet);
        gp[3] = (T)((isy * (Dm * Sxy - B * Syy) + isx * (C * Sxx - A * Sxy)) * idet);
        gp[4] = (T)(isx * Sxx);
        gp[5] = (T)(isy * Syy);
        gp[6] = (T)(S / alpha);
        // The same moments as dL/d(covariance): for the plain pdf g = exp(-d^T S^-1 d / 2), dg/dS = g S^-1 d d^T S^-1 / 2,
        // and in the eigenbasis U = [axis, perp(axis)] that is U N U^T / 2 with N = [[Sxx / sx^2, Sxy / (sx sy)], [., Syy / sy^2]]:
        // smooth in the covariance.  The projection backward gets THIS instead of (d axis, d sigma), whose chain through
        // the eigen-decomposition divides by l1 - l2 (splat_math.h project_backward): float32 gradients of nearly
        // isotropic splats no longer lose their digits.  gp[2..5] are still what a caller of viewspace_gradient() reads.
        {
          const float N00 = Sxx * isx * isx, N01 = Sxy * isx * isy, N11 = Syy * isy * isy;
          const float uu = ax * ax, ww = ay * ay, uw = ax * ay;
          gcov[0] = (T)(0.5f * (N00 * uu - 2.0f * N01 * uw + N11 * ww));
          gcov[1] = (T)((N00 - N11) * uw + N01 * (uu - ww));            // b sits off the diagonal twice
          gcov[2] = (T)(0.5f * (N00 * ww + 2.0f * N01 * uw + N11 * uu));
        }
        gf[0] = (T)r1.z; gf[1] = (T)r1.w; gf[2] = (T)r2.x;
        heur0 = (T)(alpha * alpha * r2.y);          // backward.py:190-194
        heur1 = (T)(r2.z * IS2);
        vis_sum = (T)r2.w;                          // sum of the blend weights the raster backward visited (heuristics rows)
      } else if (a.gather_world > 0) {
        // the copies of this splat came back in the rows the pack kernel sent them out in: summed in copy order
        const int copies = a.gather_route[i] >> 16;
        for (int c = 0; c < copies; ++c) {
          const int slot = a.gather_slots[i * a.gather_world + c];
          if (slot < 0) continue;
          const T* row = a.gather_rows + (int64_t)slot * a.gp_stride;
#pragma unroll
          for (int k = 0; k < 7; ++k) gp[k] += row[k];
          if (DEG >= 0)
            _Pragma("unroll") for (int ch = 0; ch < GB_MAX_F; ++ch) if (ch < a.f) gf[ch] += row[7 + ch];
        }
      } else {
        if (a.grad_points7) {
#pragma unroll
          for (int k = 0; k < 7; ++k) gp[k] = a.grad_points7[i * a.gp_stride + k];
        }
        if (DEG >= 0 && a.grad_colours)
          _Pragma("unroll") for (int c = 0; c < GB_MAX_F; ++c) if (c < a.f) gf[c] = a.grad_colours[i * a.gc_stride + c];
      }
      T gx[7] = {T(0), T(0), T(0), T(0), T(0), T(0), T(0)};      // gradients arriving at the frame's own gaussians2d output
      if (a.extra_points7) {
#pragma unroll
        for (int k = 0; k < 7; ++k) { gx[k] = a.extra_points7[i * 7 + k]; gp[k] += gx[k]; }
      }
      if ((MOM || DEG >= 0) && a.extra_colours)
        _Pragma("unroll") for (int c = 0; c < GB_MAX_F; ++c) if (c < a.f) gf[c] += a.extra_colours[i * a.f + c];

      if constexpr (MOM) {
        // rasterizer part of (axis, sigma) as a covariance gradient; only the caller's extras go through the eigen chain
        const T gq[7] = {gp[0], gp[1], gx[2], gx[3], gx[4], gx[5], gp[6]};
        project_backward(p, cam, st, gq, a.extra_depth ? a.extra_depth[i] : T(0), dp, dls, dq, dal, cam_grad, gcov);
      } else if (a.boundary_cov) {
        // rows from a rank step's strips: the rasterizer's share arrives as a covariance gradient (columns 2..4)
        const T gq[7] = {gp[0], gp[1], gx[2], gx[3], gx[4], gx[5], gp[6]};
        const T gc[3] = {gp[2] - gx[2], gp[3] - gx[3], gp[4] - gx[4]};
        project_backward(p, cam, st, gq, a.extra_depth ? a.extra_depth[i] : T(0), dp, dls, dq, dal, cam_grad, gc);
      } else {
        project_backward(p, cam, st, gp, a.extra_depth ? a.extra_depth[i] : T(0), dp, dls, dq, dal, cam_grad);
      }

      if constexpr (DEG >= 0) {
        const T dx = p[0] - a.camera_position[0], dy = p[1] - a.camera_position[1], dz = p[2] - a.camera_position[2];
        const T len = t_sqrt(dx * dx + dy * dy + dz * dz);
        T Y[D];
        sh_basis<T, DEG>(dx / len, dy / len, dz / len, Y);
#pragma unroll
        for (int d = 0; d < D; ++d) s_Y[wave][lane * YS + d] = Y[d];
        // the clamp passes the gradient strictly inside (0, 1) (sh.hip)
        _Pragma("unroll") for (int c = 0; c < GB_MAX_F; ++c) if (c < a.f) {
          const T o = a.colours[i * a.f + c];
          s_g[wave][lane * GB_MAX_F + c] = (o > T(0) && o < T(1)) ? gf[c] : T(0);
        }
      }
    } else if (valid) {
      if constexpr (DEG >= 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) s_Y[wave][lane * YS + d] = T(0);
        _Pragma("unroll") for (int c = 0; c < GB_MAX_F; ++c) if (c < a.f) s_g[wave][lane * GB_MAX_F + c] = T(0);
      }
    }

    if (valid) {
      if (a.grad_position) {
#pragma unroll
        for (int k = 0; k < 3; ++k) stream_store(&a.grad_position[i * 3 + k], dp[k]);
      }
      if (a.grad_log_scaling) {
#pragma unroll
        for (int k = 0; k < 3; ++k) stream_store(&a.grad_log_scaling[i * 3 + k], dls[k]);
      }
      if (a.grad_rotation) {
#pragma unroll
        for (int k = 0; k < 4; ++k) stream_store(&a.grad_rotation[i * 4 + k], dq[k]);
      }
      if (a.grad_alpha_logit) stream_store(&a.grad_alpha_logit[i], dal);
      if (a.store_points7) {
#pragma unroll
        for (int k = 0; k < 7; ++k) a.store_points7[i * 7 + k] = gp[k];
      }
      if ((MOM || DEG >= 0) && a.store_colours)
        _Pragma("unroll") for (int c = 0; c < GB_MAX_F; ++c) if (c < a.f) a.store_colours[i * a.f + c] = gf[c];
      if (MOM && a.point_heuristic) {
        a.point_heuristic[i * 2 + 0] = heur0;
        a.point_heuristic[i * 2 + 1] = heur1;
      }
      if (MOM && a.point_visibility) a.point_visibility[i] = vis_sum;
      if (DEG < 0 && MOM && a.grad_feature)
        _Pragma("unroll") for (int c = 0; c < GB_MAX_F; ++c) if (c < a.f) a.grad_feature[i * a.f + c] = gf[c];
    }

    if constexpr (DEG >= 0) {
      if (a.grad_feature) {
        // LDS traffic stays inside the wave; the fences keep the compiler from forwarding per-thread values
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int row = a.f * D;
        T* dst0 = a.grad_feature + base * row;
        if (D % 4 == 0 && a.f == 3 && sizeof(T) == 4 && (reinterpret_cast<uintptr_t>(a.grad_feature) & 15) == 0) {
          // RGB, degree 1 / 3, float: the wave's 64 x 3 x D values leave as 128-bit stores of consecutive addresses
          constexpr int PIECES = 3 * D / 4;
          for (int qi = lane; qi < count * PIECES; qi += 64) {
            const int j = qi / PIECES, k = qi - j * PIECES;
            const int c = (4 * k) / D, d0 = 4 * k - c * D;
            const T g = s_g[wave][j * GB_MAX_F + c];
            const T* y = &s_Y[wave][j * YS + d0];
            typedef float vec4 __attribute__((ext_vector_type(4)));
            const vec4 val = {(float)(g * y[0]), (float)(g * y[1]), (float)(g * y[2]), (float)(g * y[3])};
            stream_store(reinterpret_cast<vec4*>(dst0) + qi, val);
          }
        } else {
          for (int e = lane; e < count * row; e += 64) {
            const int j = e / row, r = e - j * row;
            dst0[e] = s_g[wave][j * GB_MAX_F + r / D] * s_Y[wave][j * YS + r % D];
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
  }

  if (a.grad_camera) block_sum_commit<T, 16>(cam_grad, a.grad_camera, s_cam);
}

template <typename T>
static int launch_typed(const GaussianBwdArgs& g, hipStream_t s) {
  GaussBwdDev<T> a;
  a.n = g.n;
  a.position = (const T*)g.position; a.log_scaling = (const T*)g.log_scaling; a.rotation = (const T*)g.rotation;
  a.alpha_logit = (const T*)g.alpha_logit; a.Tcw = (const T*)g.T_camera_world; a.proj = (const T*)g.projection;
  a.pp.width = (T)g.image_w; a.pp.height = (T)g.image_h;
  a.pp.near_plane = T(1); a.pp.far_plane = T(2);           // culling is not re-decided here (depth says)
  a.pp.blur_cov = (T)g.blur_cov; a.pp.clamp_margin = (T)g.clamp_margin; a.pp.alpha_threshold = (T)(1.0 / 255.0);
  a.depth = (const T*)g.depth;
  a.moments = g.moments; a.fixed_exp = g.fixed_exp;
  a.grad_points7 = (const T*)g.grad_points7; a.grad_colours = (const T*)g.grad_colours;
  a.gather_world = g.gather_world; a.gather_rows = (const T*)g.gather_rows;
  a.gather_slots = g.gather_slots; a.gather_route = g.gather_route;
  a.boundary_cov = g.boundary_cov;
  a.gp_stride = g.boundary_stride > 0 ? g.boundary_stride : 7;
  a.gc_stride = g.boundary_stride > 0 ? g.boundary_stride : g.f;
  a.extra_points7 = (const T*)g.extra_points7; a.extra_depth = (const T*)g.extra_depth; a.extra_colours = (const T*)g.extra_colours;
  a.f = g.f;
  a.camera_position = (const T*)g.camera_position; a.colours = (const T*)g.colours;
  a.grad_position = (T*)g.grad_position; a.grad_log_scaling = (T*)g.grad_log_scaling; a.grad_rotation = (T*)g.grad_rotation;
  a.grad_alpha_logit = (T*)g.grad_alpha_logit; a.grad_feature = (T*)g.grad_feature; a.grad_camera = (T*)g.grad_camera;
  a.store_points7 = (T*)g.store_points7; a.store_colours = (T*)g.store_colours; a.point_heuristic = (T*)g.point_heuristic;
  a.point_visibility = (T*)g.point_visibility;

  int64_t blocks = div_up(g.n, 256);
  if (g.grad_camera && blocks > 2048) blocks = 2048;      // bounded atomic count for the 16 camera sums
  const dim3 grid((unsigned)blocks), block(256);
  const bool mom = g.moments != nullptr;
#define MS_GO(DEG, MOM, FIXED) gaussian_bwd_kernel<T, DEG, MOM, FIXED><<<grid, block, 0, s>>>(a)
#define MS_GO_DEG(MOM, FIXED)                                        \
  switch (g.sh_degree) {                                             \
    case -1: MS_GO(-1, MOM, FIXED); break;                           \
    case 0: MS_GO(0, MOM, FIXED); break;                             \
    case 1: MS_GO(1, MOM, FIXED); break;                             \
    case 2: MS_GO(2, MOM, FIXED); break;                             \
    default: MS_GO(3, MOM, FIXED); break;                            \
  }
  if constexpr (sizeof(T) == 4) {
    if (mom && g.deterministic) { MS_GO_DEG(true, true) }
    else if (mom) { MS_GO_DEG(true, false) }
    else { MS_GO_DEG(false, false) }
  } else {
    MS_GO_DEG(false, false)
  }
#undef MS_GO_DEG
#undef MS_GO
  return 0;
}

int gaussian_bwd_launch(const GaussianBwdArgs& g, hipStream_t s) {
  if (g.n <= 0) return 0;
  if (g.dtype != MS_F32 && g.dtype != MS_F64) { set_error("gaussian backward: dtype must be MS_F32 or MS_F64"); return MS_ERR_BAD_ARG; }
  if (g.sh_degree < -1 || g.sh_degree > 3) { set_error("gaussian backward: SH degree must be in [0, 3]"); return MS_ERR_BAD_ARG; }
  if (g.moments && (g.dtype != MS_F32 || g.f != 3)) { set_error("gaussian backward: moment rows are float32 RGB"); return MS_ERR_BAD_ARG; }
  if (g.sh_degree >= 0 && (g.f < 1 || g.f > GB_MAX_F)) { set_error("gaussian backward: 1..4 SH colour channels"); return MS_ERR_UNSUPPORTED; }
  if (!g.position || !g.log_scaling || !g.rotation || !g.alpha_logit || !g.T_camera_world || !g.projection || !g.depth) {
    set_error("gaussian backward: null input"); return MS_ERR_BAD_ARG;
  }
  if (g.sh_degree >= 0 && (!g.camera_position || !g.colours)) { set_error("gaussian backward: SH needs camera position and forward colours"); return MS_ERR_BAD_ARG; }
  if (g.moments && g.deterministic && !g.fixed_exp) { set_error("gaussian backward: deterministic rows need their scale exponents"); return MS_ERR_BAD_ARG; }
  if (g.dtype == MS_F32) launch_typed<float>(g, s);
  else launch_typed<double>(g, s);
  MS_CHECK_LAUNCH();
  return 0;
}

}  // namespace ms
