// optim.hip — the optimiser step that consumes the render path's gradients and visibilities each training iteration
// (SURVEY.md section 8f, rank N3): per-visible-point moment updates of the fractional (visibility-weighted) Adam /
// LaProp optimisers.  Replaces the Taichi kernels of optim/fractional_adam.py:8-86 and optim/fractional_laprop.py:8-86
// and the host logic around them (optim/fractional.py:108-156,176-195, optim/visibility_aware.py:25-41,86-104).
//
// HBM-streaming work: per element of a scalar group 4 loads (gradient, two moments, parameter) and 3 stores, 28 bytes;
// 6 M gaussians x 59 parameters = 9.9 GB per step.  Organisation (round 6):
//   * LPP lanes cooperate on one point, and a lane moves 16-BYTE pieces of the rows (VEC = 4: every row length that is a
//     multiple of 4 floats — rotation 4, SH colours 48) so that each load / store instruction covers whole 64-byte
//     sectors and carries 1 KB per wave; rows of other lengths (3, 1) keep 4-byte pieces;
//   * the per-point scalars (beta^w, bias corrections: four v_exp_f32) are paid once per LANE = once per 4 elements on
//     the wide rows; per-point reductions (squared gradient norm of vector groups, the basis products of local_vector
//     groups) stay inside the LPP-lane group (DPP / ds_bpermute shuffles, fixed order: deterministic);
//   * ALL parameter groups of an optimiser step run in ONE launch (ms_optim_step_groups): a workgroup takes 256
//     consecutive visible points through every group, so the index / weight / total-weight words are fetched from HBM
//     once per point instead of once per (point, group);
//   * the visibility-aware weights (running power mean, total weight, gradient scale: ~12 elementwise torch launches in
//     the host formulation) are ONE pass (ms_optim_visibility_weights), which can also skip invisible rows itself
//     (no torch.nonzero, no host synchronisation in the training loop).
#include <cstdlib>
#include "common.h"

namespace ms {

__device__ __forceinline__ float lerp_t(float t, float a, float b) { return a * t + b * (1.0f - t); }   // taichi_lib/generic.py:488-490

// KIND 0: Adam (fractional_adam.py), KIND 1: LaProp (fractional_laprop.py).
// VECTOR: one second-moment value per point (the squared gradient norm) instead of one per element.
// (thread-per-point fallback of ms_fractional_step for rows wider than 256 floats)
template <int KIND, bool VECTOR>
__global__ void __launch_bounds__(256)
fractional_step_kernel(float* __restrict__ lr_step, const int64_t* __restrict__ indexes,
                       const float* __restrict__ weight, float* __restrict__ m_arr, float* __restrict__ v_arr,
                       const float* __restrict__ total_weight, const float* __restrict__ grad, int64_t m_count,
                       int d, float lr, float beta1, float beta2, float eps, int bias_correction) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m_count) return;
  const int64_t idx = indexes[i];
  const float w = weight[i];
  const float tw = total_weight[idx];
  const float b1w = powf(beta1, w), b2w = powf(beta2, w);
  const float bias1 = bias_correction ? 1.0f - powf(beta1, tw) : 1.0f;
  const float bias2 = bias_correction ? 1.0f - powf(beta2, tw) : 1.0f;
  const float* g = grad + idx * d;
  float* mp = m_arr + idx * d;
  float* out = lr_step + i * d;

  if (VECTOR) {
    float norm = 0.f;
    for (int j = 0; j < d; ++j) norm += g[j] * g[j];
    const float v = lerp_t(b2w, v_arr[idx], norm);
    v_arr[idx] = v;
    if (KIND == 0) {
      const float scale = (bias_correction ? sqrtf(bias2) / bias1 : 1.0f) * lr / fmaxf(sqrtf(v), eps);
      for (int j = 0; j < d; ++j) {
        const float m = lerp_t(b1w, mp[j], g[j]);
        out[j] = m * scale;
        mp[j] = m;
      }
    } else {
      const float inv = 1.0f / fmaxf(sqrtf(v / bias2), eps);
      for (int j = 0; j < d; ++j) {
        const float m = lerp_t(b1w, mp[j], g[j] * inv);
        out[j] = m * lr / bias1;
        mp[j] = m;
      }
    }
  } else {
    float* vp = v_arr + idx * d;
    const float bias_factor = bias_correction ? sqrtf(bias2) / bias1 : 1.0f;
    for (int j = 0; j < d; ++j) {
      const float gj = g[j];
      const float v = lerp_t(b2w, vp[j], gj * gj);
      float m;
      if (KIND == 0) {
        m = lerp_t(b1w, mp[j], gj);
        out[j] = m / fmaxf(sqrtf(v), eps) * bias_factor * lr;
      } else {
        m = lerp_t(b1w, mp[j], gj / fmaxf(sqrtf(v / bias2), eps));
        out[j] = m * lr / bias1;
      }
      mp[j] = m;
      vp[j] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Group update: everything the reference does per parameter group and step on the host side of
// optim/fractional.py:108-156,176-195 (and the gradient pre-scaling of visibility_aware.py:95-104) in ONE
// HBM-bound pass over the visible rows:
//   g = grad[idx] * grad_scale[i]                  (visibility-aware: 1 / (visibility + vis_smooth))
//   local_vector: g = basis[i]^-1 g                (D = 2 or 3, closed-form inverse)
//   moment update of ms_fractional_step            (vector types: one second moment per point)
//   step = clamp(step, +-lr * clip); local_vector: step = basis[i] step; *= mask_lr[j]; *= point_lr[idx]
//   non-finite -> 0;  param[idx] -= step * (1 - exp(-2 weight[i]))
// A row with weight[i] < 0 is SKIPPED (nothing read or written): ms_optim_visibility_weights marks invisible rows
// that way in its dense mode.
// The first version (torch.linalg.inv + two batched einsum = rocBLAS batched 3x3 GEMMs, gathers, index_put and a
// thread-per-point kernel with D-strided accesses) took 166 ms per step for 6 M gaussians (59 floats each).
// ------------------------------------------------------------------------------------------------
struct GroupArgs {
  float* param; const float* grad; float* m; float* v;
  const float* basis; const float* mask_lr; const float* point_lr;
  int d; float lr, eps, clip; int bias_correction;
  float log2_beta1, log2_beta2;      // beta^w = exp2(w log2 beta): v_exp_f32 instead of powf
  float* out_step;                   // ms_fractional_step: write the raw step (M, D) instead of updating param
};

struct CommonArgs {
  const int64_t* indexes;            // NULL: rows 0 .. m_count - 1
  const float* weight; const float* total_weight; const float* grad_scale;
  int64_t m_count;
};

template <int LPP>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
  for (int off = LPP / 2; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

// -DMS_OPTIM_NT=1 (tools/build_variant.sh): the gradient rows — read once, never again — are loaded non-temporally
#ifndef MS_OPTIM_NT
#define MS_OPTIM_NT 0
#endif
template <int VEC> struct Piece;
template <> struct Piece<1> {
  static __device__ __forceinline__ void load(const float* p, float* out) { out[0] = *p; }
  static __device__ __forceinline__ void load_once(const float* p, float* out) {
#if MS_OPTIM_NT
    out[0] = __builtin_nontemporal_load(p);
#else
    out[0] = *p;
#endif
  }
  static __device__ __forceinline__ void store(float* p, const float* v) { *p = v[0]; }
};
template <> struct Piece<4> {
  static __device__ __forceinline__ void load(const float* p, float* out) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w;
  }
  static __device__ __forceinline__ void load_once(const float* p, float* out) {
#if MS_OPTIM_NT
    typedef float vec4 __attribute__((ext_vector_type(4)));
    const vec4 q = __builtin_nontemporal_load(reinterpret_cast<const vec4*>(p));
    out[0] = q.x; out[1] = q.y; out[2] = q.z; out[3] = q.w;
#else
    load(p, out);
#endif
  }
  static __device__ __forceinline__ void store(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

// beta^w with beta^0 = 1 also for beta = 0 (log2 beta = -inf would give 0 * -inf = NaN and poison the
// persistent moments of a point with zero weight, e.g. zero visibility; the reference's beta ** w gives 1)
__device__ __forceinline__ float pow_beta(float e, float log2_beta) { return e == 0.0f ? 1.0f : exp2f(e * log2_beta); }

// One visible point i, lane `sub` of its LPP-lane group.  TYPE: 0 scalar, 1 vector, 2 local_vector.  A lane owns the
// pieces sub, sub + LPP, ... (KMAX of them) of VEC floats each: D <= LPP * KMAX * VEC.  Whole LPP groups are live or not,
// so the shuffles stay uniform.
// the per-point words every group of a step shares
struct PointWords {
  int64_t idx;      // row of the parameter arrays
  float w;          // step weight (< 0: skip)
  float tw;         // total weight (after this step's weight was added)
  float gscale;     // gradient scale
};

__device__ __forceinline__ PointWords load_point_words(const CommonArgs& c, int64_t i, bool in_range) {
  PointWords p;
  p.w = in_range ? c.weight[i] : -1.0f;
  const bool live = p.w >= 0.0f;                  // (NaN weights skip as well)
  p.idx = live ? (c.indexes ? c.indexes[i] : i) : 0;
  p.tw = live ? c.total_weight[p.idx] : 1.f;
  p.gscale = (live && c.grad_scale) ? c.grad_scale[i] : 1.f;
  return p;
}

template <int KIND, int TYPE, int LPP, int KMAX, int VEC>
__device__ __forceinline__ void update_point(const GroupArgs& a, const PointWords& pw, int64_t i, int sub) {
  static_assert(TYPE != 2 || (VEC == 1 && LPP == 4 && KMAX == 1), "local_vector rows are 2 or 3 floats");
  constexpr int E = KMAX * VEC;
  const float w = pw.w;
  const bool live = w >= 0.0f;
  const int64_t idx = pw.idx;
  const int d = a.d;
  const float tw = pw.tw;
  const float gscale = pw.gscale;
  const int64_t row = idx * d;

  float g[E], mo[E], pa[E], vo[TYPE == 0 ? E : 1];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int j = (sub + k * LPP) * VEC;
    const bool ok = live && j < d;
    if (ok) {
      Piece<VEC>::load_once(a.grad + row + j, g + k * VEC);
      Piece<VEC>::load(a.m + row + j, mo + k * VEC);
      if (TYPE == 0) Piece<VEC>::load(a.v + row + j, vo + k * VEC);
      if (!a.out_step) Piece<VEC>::load(a.param + row + j, pa + k * VEC);
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      g[k * VEC + e] = ok ? g[k * VEC + e] * gscale : 0.f;
      if (!ok) { mo[k * VEC + e] = 0.f; pa[k * VEC + e] = 0.f; if (TYPE == 0) vo[k * VEC + e] = 0.f; }
    }
  }

  float B[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
  if constexpr (TYPE == 2) {
    // basis (M, D, D) row-major, D in {2, 3}; g_local = B^-1 g (adjugate / determinant)
    if (live) {
      for (int r = 0; r < d; ++r)
        for (int col = 0; col < d; ++col) B[r][col] = a.basis[(i * d + r) * d + col];
    }
    const float c00 = B[1][1] * B[2][2] - B[1][2] * B[2][1];
    const float c01 = B[1][2] * B[2][0] - B[1][0] * B[2][2];
    const float c02 = B[1][0] * B[2][1] - B[1][1] * B[2][0];
    const float inv_det = 1.0f / (B[0][0] * c00 + B[0][1] * c01 + B[0][2] * c02);
    float inv_row[3];     // row `sub` of the inverse
    if (sub == 0) { inv_row[0] = c00; inv_row[1] = B[0][2] * B[2][1] - B[0][1] * B[2][2]; inv_row[2] = B[0][1] * B[1][2] - B[0][2] * B[1][1]; }
    else if (sub == 1) { inv_row[0] = c01; inv_row[1] = B[0][0] * B[2][2] - B[0][2] * B[2][0]; inv_row[2] = B[0][2] * B[1][0] - B[0][0] * B[1][2]; }
    else { inv_row[0] = c02; inv_row[1] = B[0][1] * B[2][0] - B[0][0] * B[2][1]; inv_row[2] = B[0][0] * B[1][1] - B[0][1] * B[1][0]; }
    const int base = (int)(threadIdx.x & 63) - sub;
    const float g0 = __shfl(g[0], base + 0, 64), g1 = __shfl(g[0], base + 1, 64), g2 = __shfl(g[0], base + 2, 64);
    const float gl = (inv_row[0] * g0 + inv_row[1] * g1 + inv_row[2] * g2) * inv_det;
    g[0] = sub < d ? gl : 0.f;
  }

  const float b1w = pow_beta(w, a.log2_beta1), b2w = pow_beta(w, a.log2_beta2);
  const float bias1 = a.bias_correction ? 1.0f - pow_beta(tw, a.log2_beta1) : 1.0f;
  const float bias2 = a.bias_correction ? 1.0f - pow_beta(tw, a.log2_beta2) : 1.0f;

  float step[E];
  if constexpr (TYPE != 0) {
    float norm = 0.f;
#pragma unroll
    for (int e = 0; e < E; ++e) norm += g[e] * g[e];
    norm = group_sum<LPP>(norm);
    float v = 0.f;
    if (live) {
      v = lerp_t(b2w, a.v[idx], norm);
      if (sub == 0) a.v[idx] = v;
    }
    const float scale = KIND == 0 ? (a.bias_correction ? sqrtf(bias2) / bias1 : 1.0f) * a.lr / fmaxf(sqrtf(v), a.eps)
                                  : 1.0f / fmaxf(sqrtf(v / bias2), a.eps);
#pragma unroll
    for (int e = 0; e < E; ++e) {
      if (KIND == 0) {
        mo[e] = lerp_t(b1w, mo[e], g[e]);
        step[e] = mo[e] * scale;
      } else {
        mo[e] = lerp_t(b1w, mo[e], g[e] * scale);
        step[e] = mo[e] * a.lr / bias1;
      }
    }
  } else {
    const float bias_factor = a.bias_correction ? sqrtf(bias2) / bias1 : 1.0f;
#pragma unroll
    for (int e = 0; e < E; ++e) {
      vo[e] = lerp_t(b2w, vo[e], g[e] * g[e]);
      if (KIND == 0) {
        mo[e] = lerp_t(b1w, mo[e], g[e]);
        step[e] = mo[e] / fmaxf(sqrtf(vo[e]), a.eps) * bias_factor * a.lr;
      } else {
        mo[e] = lerp_t(b1w, mo[e], g[e] / fmaxf(sqrtf(vo[e] / bias2), a.eps));
        step[e] = mo[e] * a.lr / bias1;
      }
    }
  }

  if (a.clip >= 0.f) {
    const float max_step = a.lr * a.clip;
#pragma unroll
    for (int e = 0; e < E; ++e)
      if (step[e] == step[e]) step[e] = fminf(fmaxf(step[e], -max_step), max_step);   // NaN stays NaN -> zeroed below
  }
  if constexpr (TYPE == 2) {
    const int base = (int)(threadIdx.x & 63) - sub;
    const float s0 = __shfl(step[0], base + 0, 64), s1 = __shfl(step[0], base + 1, 64), s2 = __shfl(step[0], base + 2, 64);
    const int r = sub < 3 ? sub : 0;
    step[0] = B[r][0] * s0 + B[r][1] * s1 + (d > 2 ? B[r][2] * s2 : 0.f);
  }
  const float sat = 1.0f - expf(-2.0f * w);
  const float plr = (live && a.point_lr) ? a.point_lr[idx] : 1.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int j = (sub + k * LPP) * VEC;
    if (!(live && j < d)) continue;
    float out[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float sj = step[k * VEC + e] * plr;
      if (a.mask_lr) sj *= a.mask_lr[j + e];
      if (a.out_step) { out[e] = sj; continue; }
      if (!(fabsf(sj) < 3.0e38f)) sj = 0.f;          // non-finite -> 0 (fractional.py:153)
      out[e] = pa[k * VEC + e] - sj * sat;
    }
    Piece<VEC>::store(a.m + row + j, mo + k * VEC);
    if (TYPE == 0) Piece<VEC>::store(a.v + row + j, vo + k * VEC);
    if (a.out_step) Piece<VEC>::store(a.out_step + i * d + j, out);
    else Piece<VEC>::store(a.param + row + j, out);
  }
}

// one group per launch: lane t -> (point t / LPP, piece t % LPP)
template <int KIND, int TYPE, int LPP, int KMAX, int VEC>
__global__ void __launch_bounds__(256)
fractional_update_kernel(GroupArgs a, CommonArgs c) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t / LPP;
  update_point<KIND, TYPE, LPP, KMAX, VEC>(a, load_point_words(c, i, i < c.m_count), i, (int)(t % LPP));
}

// Row shapes.  Rows whose length is a multiple of 4 floats (and whose arrays are 16-byte aligned) move as 16-byte
// pieces; wide rows use 16 lanes x KMAX pieces rather than more lanes: the per-point scalar work is paid once per lane.
enum Shape { S_V1_L1 = 0, S_V1_L4, S_V1_L16, S_V1_L16_K4, S_V1_L16_K16, S_V4_L1, S_V4_L4, S_V4_L16, S_V4_L16_K4, S_COUNT };

__host__ __device__ inline int shape_lanes(int shape) {
  return shape == S_V1_L1 || shape == S_V4_L1 ? 1 : shape == S_V1_L4 || shape == S_V4_L4 ? 4 : 16;
}

// ALL groups of a step in one launch: a workgroup takes FUSED_POINTS consecutive visible points through every group
constexpr int MAX_FUSED_GROUPS = 8;
constexpr int FUSED_POINTS = 256;
struct FusedArgs {
  CommonArgs c;
  int num_groups;
  GroupArgs g[MAX_FUSED_GROUPS];
  int type[MAX_FUSED_GROUPS];
  int shape[MAX_FUSED_GROUPS];
};

// the workgroup's FUSED_POINTS points: their shared words are fetched ONCE into LDS (round 6, second step: with the index
// list in global memory every pass of every group began with a dependent load — index, then rows — and the indexed step ran
// 7 % behind the dense one)
struct FusedShared {
  int64_t idx[FUSED_POINTS];
  float w[FUSED_POINTS], tw[FUSED_POINTS], gscale[FUSED_POINTS];
};

template <int KIND, int TYPE, int LPP, int KMAX, int VEC>
__device__ __forceinline__ void fused_group(const GroupArgs& a, const CommonArgs& c, const FusedShared& sh, int64_t first) {
  constexpr int PER_PASS = 256 / LPP;
  const int sub = (int)(threadIdx.x % LPP), lane_point = (int)(threadIdx.x / LPP);
#pragma unroll 1
  for (int p = 0; p < FUSED_POINTS; p += PER_PASS) {
    if (first + p >= c.m_count) break;            // uniform over the workgroup
    const int q = p + lane_point;
    PointWords pw;
    pw.idx = sh.idx[q]; pw.w = sh.w[q]; pw.tw = sh.tw[q]; pw.gscale = sh.gscale[q];
    update_point<KIND, TYPE, LPP, KMAX, VEC>(a, pw, first + q, sub);
  }
}

template <int KIND>
__global__ void __launch_bounds__(256)
optim_fused_kernel(FusedArgs f) {
  const int64_t first = (int64_t)blockIdx.x * FUSED_POINTS;
  __shared__ FusedShared sh;
  {
    const int64_t i = first + threadIdx.x;
    const PointWords pw = load_point_words(f.c, i, i < f.c.m_count);
    sh.idx[threadIdx.x] = pw.idx; sh.w[threadIdx.x] = pw.w; sh.tw[threadIdx.x] = pw.tw; sh.gscale[threadIdx.x] = pw.gscale;
  }
  __syncthreads();
#pragma unroll 1
  for (int gi = 0; gi < f.num_groups; ++gi) {
    const GroupArgs& a = f.g[gi];
    const int type = f.type[gi];
#define MS_FG(T, L, K, V) fused_group<KIND, T, L, K, V>(a, f.c, sh, first)
#define MS_FT(L, K, V) do { if (type == 0) MS_FG(0, L, K, V); else MS_FG(1, L, K, V); } while (0)
    switch (f.shape[gi]) {
      case S_V1_L1: MS_FT(1, 1, 1); break;
      case S_V1_L4: MS_FT(4, 1, 1); break;
      case S_V1_L16: MS_FT(16, 1, 1); break;
      case S_V4_L1: MS_FT(1, 1, 4); break;
      case S_V4_L4: MS_FT(4, 1, 4); break;
      case S_V4_L16: MS_FT(16, 1, 4); break;
      default: break;                             // (wide shapes are launched on their own: launch_groups)
    }
#undef MS_FT
#undef MS_FG
  }
}

// Step weights of the visibility-aware optimisers (optim/visibility_aware.py:25-41 update_visibility, :86-104):
//   r <- ((1 - beta) v^4 + beta r^4)^(1/4) at the listed rows (running power mean, order 4);  w = v / max(r, floor);
//   total_weight += w;  grad_scale = 1 / (v + smooth).
// indexes == NULL (dense mode): row i is point i, and a point with visibility <= threshold is skipped — nothing of it is
// touched and out_weight[i] = -1, which the group kernels read as "skip".
__global__ void __launch_bounds__(256)
visibility_weights_kernel(const int64_t* __restrict__ indexes, const float* __restrict__ visibility, int64_t m_count,
                          float beta, float smooth, float floor_, float threshold, float* __restrict__ running,
                          float* __restrict__ total_weight, float* __restrict__ out_weight, float* __restrict__ out_scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m_count) return;
  const float v = visibility[i];
  if (!indexes && !(v > threshold)) {
    out_weight[i] = -1.0f;
    if (out_scale) out_scale[i] = 0.0f;
    return;
  }
  const int64_t idx = indexes ? indexes[i] : i;
  const float r = running[idx];
  const float v2 = v * v, r2 = r * r, v4 = v2 * v2, r4 = r2 * r2;
  const float mixed = sqrtf(sqrtf(v4 + (r4 - v4) * beta));
  running[idx] = mixed;
  const float w = v / fmaxf(mixed, floor_);
  total_weight[idx] += w;
  out_weight[i] = w;
  if (out_scale) out_scale[i] = 1.0f / (v + smooth);
}

}  // namespace ms

using namespace ms;

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// scalar groups read and write v row-wise like m; vector groups keep one float per point there
static int pick_shape(int group_type, const GroupArgs& a) {
  const int d = a.d;
  if (group_type == 2) return S_V1_L4;
  // MS_OPTIM_VEC=0 in the environment (read once): 4-byte pieces everywhere, for A/B measurements of the 16-byte pieces
  static const bool allow_vec = [] { const char* e = getenv("MS_OPTIM_VEC"); return !(e && e[0] == '0'); }();
  const bool vec = allow_vec && d % 4 == 0 && aligned16(a.param) && aligned16(a.grad) && aligned16(a.m) && aligned16(a.out_step) &&
                   (group_type != 0 || aligned16(a.v));
  if (vec) return d == 4 ? S_V4_L1 : d <= 16 ? S_V4_L4 : d <= 64 ? S_V4_L16 : S_V4_L16_K4;
  return d == 1 ? S_V1_L1 : d <= 4 ? S_V1_L4 : d <= 16 ? S_V1_L16 : d <= 64 ? S_V1_L16_K4 : S_V1_L16_K16;
}

static int launch_update(int kind, int group_type, const GroupArgs& a, const CommonArgs& c, hipStream_t s) {
  const int shape = pick_shape(group_type, a);
  const int lpp = shape_lanes(shape);
  const dim3 block(256), grid((unsigned)div_up(c.m_count * lpp, 256));
#define MS_GO(K, T, L, KM, V) fractional_update_kernel<K, T, L, KM, V><<<grid, block, 0, s>>>(a, c)
#define MS_SHAPES(K, T)                                                                                  \
  switch (shape) {                                                                                      \
    case S_V1_L1: MS_GO(K, T, 1, 1, 1); break; case S_V1_L4: MS_GO(K, T, 4, 1, 1); break;               \
    case S_V1_L16: MS_GO(K, T, 16, 1, 1); break; case S_V1_L16_K4: MS_GO(K, T, 16, 4, 1); break;        \
    case S_V1_L16_K16: MS_GO(K, T, 16, 16, 1); break; case S_V4_L1: MS_GO(K, T, 1, 1, 4); break;        \
    case S_V4_L4: MS_GO(K, T, 4, 1, 4); break; case S_V4_L16: MS_GO(K, T, 16, 1, 4); break;             \
    default: MS_GO(K, T, 16, 4, 4); break; }
  if (kind == 0) {
    if (group_type == 0) { MS_SHAPES(0, 0) } else if (group_type == 1) { MS_SHAPES(0, 1) } else { MS_GO(0, 2, 4, 1, 1); }
  } else {
    if (group_type == 0) { MS_SHAPES(1, 0) } else if (group_type == 1) { MS_SHAPES(1, 1) } else { MS_GO(1, 2, 4, 1, 1); }
  }
#undef MS_SHAPES
#undef MS_GO
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_fractional_step(int kind, int vector, float* lr_step, const int64_t* indexes,
                                  const float* weight, float* m, float* v, const float* total_weight,
                                  const float* grad, int64_t m_count, int d, float lr, float beta1,
                                  float beta2, float eps, int bias_correction, void* stream) {
  MS_CHECK_ARG(kind == 0 || kind == 1, "kind must be 0 (adam) or 1 (laprop)");
  MS_CHECK_ARG(m_count >= 0 && d >= 1, "bad sizes");
  if (m_count == 0) return 0;
  MS_CHECK_ARG(lr_step && indexes && weight && m && v && total_weight && grad, "null pointer");
  if (d <= 256) {       // the cooperative (coalesced) kernel of ms_fractional_update, writing the raw step
    GroupArgs a{nullptr, grad, m, v, nullptr, nullptr, nullptr, d, lr, eps, -1.0f, bias_correction,
                (float)log2((double)beta1), (float)log2((double)beta2), lr_step};
    CommonArgs c{indexes, weight, total_weight, nullptr, m_count};
    return launch_update(kind, vector ? 1 : 0, a, c, (hipStream_t)stream);
  }
  const dim3 block(256), grid((unsigned)div_up(m_count, 256));
  hipStream_t s = (hipStream_t)stream;
#define MS_GO(K, V) fractional_step_kernel<K, V><<<grid, block, 0, s>>>(lr_step, indexes, weight, m, v, total_weight, grad, m_count, d, lr, beta1, beta2, eps, bias_correction)
  if (kind == 0) { if (vector) MS_GO(0, true); else MS_GO(0, false); }
  else { if (vector) MS_GO(1, true); else MS_GO(1, false); }
#undef MS_GO
  MS_CHECK_LAUNCH();
  return 0;
}

static int check_group(int group_type, int d, const void* param, const void* grad, const void* m, const void* v,
                       const void* basis, const char* who) {
  if (group_type < 0 || group_type > 2) { set_error("%s: group_type must be 0 (scalar), 1 (vector) or 2 (local_vector)", who); return MS_ERR_BAD_ARG; }
  if (d < 1 || d > 256) { set_error("%s: 1 <= d <= 256 expected (got %d)", who, d); return MS_ERR_BAD_ARG; }
  if (!(param && grad && m && v)) { set_error("%s: null pointer", who); return MS_ERR_BAD_ARG; }
  if (group_type == 2) {
    if (!basis) { set_error("%s: local_vector groups need the basis", who); return MS_ERR_BAD_ARG; }
    if (d != 2 && d != 3) { set_error("%s: local_vector groups are 2 or 3 dimensional", who); return MS_ERR_BAD_ARG; }
  }
  return 0;
}

extern "C" int ms_fractional_update(int kind, int group_type, float* param, const float* grad, float* m, float* v,
                                    const int64_t* indexes, const float* weight, const float* total_weight,
                                    const float* grad_scale, const float* basis, const float* mask_lr,
                                    const float* point_lr, int64_t m_count, int d, float lr, float beta1,
                                    float beta2, float eps, float clip, int bias_correction, void* stream) {
  MS_CHECK_ARG(kind == 0 || kind == 1, "kind must be 0 (Adam) or 1 (LaProp)");
  MS_CHECK_ARG(m_count >= 0, "m_count >= 0 expected");
  if (m_count == 0) return 0;
  const int rc = check_group(group_type, d, param, grad, m, v, basis, "ms_fractional_update");
  if (rc) return rc;
  MS_CHECK_ARG(indexes && weight && total_weight, "null pointer");
  GroupArgs a{param, grad, m, v, basis, mask_lr, point_lr, d, lr, eps, clip, bias_correction,
              (float)log2((double)beta1), (float)log2((double)beta2), nullptr};
  CommonArgs c{indexes, weight, total_weight, grad_scale, m_count};
  return launch_update(kind, group_type, a, c, (hipStream_t)stream);
}

extern "C" int ms_optim_step_groups(int kind, const ms_optim_group* groups, int num_groups, const int64_t* indexes,
                                    const float* weight, const float* total_weight, const float* grad_scale,
                                    int64_t m_count, void* stream) {
  MS_CHECK_ARG(kind == 0 || kind == 1, "kind must be 0 (Adam) or 1 (LaProp)");
  MS_CHECK_ARG(num_groups >= 0 && (groups || num_groups == 0), "bad group list");
  MS_CHECK_ARG(m_count >= 0, "m_count >= 0 expected");
  if (m_count == 0 || num_groups == 0) return 0;
  MS_CHECK_ARG(weight && total_weight, "null pointer");
  hipStream_t s = (hipStream_t)stream;
  const CommonArgs c{indexes, weight, total_weight, grad_scale, m_count};
  FusedArgs f{};
  f.c = c;
  auto flush = [&]() -> int {
    if (f.num_groups == 0) return 0;
    const dim3 grid((unsigned)div_up(m_count, FUSED_POINTS));
    if (kind == 0) optim_fused_kernel<0><<<grid, dim3(256), 0, s>>>(f);
    else optim_fused_kernel<1><<<grid, dim3(256), 0, s>>>(f);
    f.num_groups = 0;
    MS_CHECK_LAUNCH();
    return 0;
  };
  for (int gi = 0; gi < num_groups; ++gi) {
    const ms_optim_group& g = groups[gi];
    if (g.struct_size != sizeof(ms_optim_group)) {
      set_error("ms_optim_step_groups: ms_optim_group of another ABI (struct_size %u, this library: %u)", g.struct_size,
                (unsigned)sizeof(ms_optim_group));
      return MS_ERR_ABI;
    }
    const int rc = check_group(g.group_type, g.d, g.param, g.grad, g.m, g.v, g.basis, "ms_optim_step_groups");
    if (rc) return rc;
    const GroupArgs a{g.param, g.grad, g.m, g.v, g.basis, g.mask_lr, g.point_lr, g.d, g.lr, g.eps, g.clip, g.bias_correction,
                      (float)log2((double)g.beta1), (float)log2((double)g.beta2), nullptr};
    const int shape = pick_shape(g.group_type, a);
    // MS_OPTIM_FUSED=0 (read once): one launch per group, for A/B measurements of the fused launch
    static const bool allow_fused = [] { const char* e = getenv("MS_OPTIM_FUSED"); return !(e && e[0] == '0'); }();
    // (local_vector groups keep a launch of their own as well: their 3 x 3 basis products cost the fused kernel 16 VGPRs
    // and 9 KB of LDS — a wave per SIMD — for a group type a 3D model does not have)
    const bool fusable = allow_fused && g.group_type != 2 && shape != S_V1_L16_K4 && shape != S_V1_L16_K16 && shape != S_V4_L16_K4;
    if (!fusable) {                                  // rows wider than 64 floats (or 16 unaligned): a launch of their own
      const int rc2 = launch_update(kind, g.group_type, a, c, s);
      if (rc2) return rc2;
      continue;
    }
    if (f.num_groups == MAX_FUSED_GROUPS) { const int rc3 = flush(); if (rc3) return rc3; }
    f.g[f.num_groups] = a; f.type[f.num_groups] = g.group_type; f.shape[f.num_groups] = shape;
    ++f.num_groups;
  }
  return flush();
}

extern "C" int ms_optim_visibility_weights(const int64_t* indexes, const float* visibility, int64_t m_count, float vis_beta,
                                           float vis_smooth, float floor_eps, float skip_threshold, float* running_vis,
                                           float* total_weight, float* out_weight, float* out_grad_scale, void* stream) {
  MS_CHECK_ARG(m_count >= 0, "m_count >= 0 expected");
  if (m_count == 0) return 0;
  MS_CHECK_ARG(visibility && running_vis && total_weight && out_weight, "null pointer");
  visibility_weights_kernel<<<dim3((unsigned)div_up(m_count, 256)), dim3(256), 0, (hipStream_t)stream>>>(
      indexes, visibility, m_count, vis_beta, vis_smooth, floor_eps, skip_threshold, running_vis, total_weight, out_weight,
      out_grad_scale);
  MS_CHECK_LAUNCH();
  return 0;
}
