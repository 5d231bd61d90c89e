// optim.hip — per-visible-point moment updates of the fractional (visibility-weighted) Adam / LaProp
// optimisers: the step that consumes the render path's gradients and visibility each iteration
// (SURVEY.md section 8f, rank N3).  Replaces the Taichi kernels of optim/fractional_adam.py:8-86 and
// optim/fractional_laprop.py:8-86.  One thread per visible point, rows of D contiguous floats:
// HBM-streaming (gather by the int64 visible-index list).
#include "common.h"

namespace ms {

__device__ __forceinline__ float lerp_t(float t, float a, float b) { return a * t + b * (1.0f - t); }   // taichi_lib/generic.py:488-490

// KIND 0: Adam (fractional_adam.py), KIND 1: LaProp (fractional_laprop.py).
// VECTOR: one second-moment value per point (the squared gradient norm) instead of one per element.
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
// Fused group update: everything the reference does per parameter group and step on the host side of
// optim/fractional.py:108-156,176-195 (and the gradient pre-scaling of visibility_aware.py:95-104) in ONE
// HBM-bound pass over the visible rows:
//   g = grad[idx] * grad_scale[i]                  (visibility-aware: 1 / (visibility + vis_smooth))
//   local_vector: g = basis[i]^-1 g                (D = 2 or 3, closed-form inverse)
//   moment update of ms_fractional_step            (vector types: one second moment per point)
//   step = clamp(step, +-lr * clip); local_vector: step = basis[i] step; *= mask_lr[j]; *= point_lr[idx]
//   non-finite -> 0;  param[idx] -= step * (1 - exp(-2 weight[i]))
// LPP lanes cooperate on one point (element j = sub, sub + LPP, ...), so rows are read and written with
// coalesced accesses for any D; the per-point norm and the D x D products go through ds_bpermute inside the
// LPP-lane group.  The first version (torch.linalg.inv + two batched einsum = rocBLAS batched 3x3 GEMMs, gathers,
// index_put and a thread-per-point kernel with D-strided accesses) took 166 ms per step for 6 M gaussians
// (59 floats each); this pass moves ~1.3 KB per point.
// ------------------------------------------------------------------------------------------------
struct UpdateArgs {
  float* param; const float* grad; float* m; float* v; const int64_t* indexes; const float* weight;
  const float* total_weight; const float* grad_scale; const float* basis; const float* mask_lr;
  const float* point_lr;
  int64_t m_count; int d; float lr, beta1, beta2, eps, clip; int bias_correction;
  float log2_beta1, log2_beta2;      // beta^w = exp2(w log2 beta): v_exp_f32 instead of powf
  float* out_step;                   // ms_fractional_step: write the raw step (M, D) instead of updating param
};

template <int LPP>
__device__ __forceinline__ float group_sum(float x) {
#pragma unroll
  for (int off = LPP / 2; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

// TYPE: 0 scalar, 1 vector, 2 local_vector
// KMAX = elements per lane (D <= LPP * KMAX).  Wide rows use 16 lanes x KMAX elements rather than more lanes:
// the per-point scalar work (four exponentials, bias terms) is paid once per LPP lanes.
template <int KIND, int TYPE, int LPP, int KMAX>
__global__ void __launch_bounds__(256)
fractional_update_kernel(UpdateArgs a) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t i = t / LPP;
  const int sub = (int)(t % LPP);
  const bool live = i < a.m_count;               // whole LPP groups are live or not: shuffles stay uniform
  const int64_t idx = live ? a.indexes[i] : 0;
  const int d = a.d;
  const float w = live ? a.weight[i] : 0.f;
  const float tw = live ? a.total_weight[idx] : 1.f;
  const float gscale = (live && a.grad_scale) ? a.grad_scale[i] : 1.f;

  float g[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int j = sub + k * LPP;
    g[k] = (live && j < d) ? a.grad[idx * d + j] * gscale : 0.f;
  }

  float B[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
  if (TYPE == 2) {
    // basis (M, D, D) row-major, D in {2, 3}; g_local = B^-1 g (adjugate / determinant)
    if (live) {
      for (int r = 0; r < d; ++r)
        for (int c = 0; c < d; ++c) B[r][c] = a.basis[(i * d + r) * d + c];
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

  // beta^w with beta^0 = 1 also for beta = 0 (log2 beta = -inf would give 0 * -inf = NaN and poison the
  // persistent moments of a point with zero weight, e.g. zero visibility; the reference's beta ** w gives 1)
  auto pow_beta = [](float e, float log2_beta) { return e == 0.0f ? 1.0f : exp2f(e * log2_beta); };
  const float b1w = pow_beta(w, a.log2_beta1), b2w = pow_beta(w, a.log2_beta2);
  const float bias1 = a.bias_correction ? 1.0f - pow_beta(tw, a.log2_beta1) : 1.0f;
  const float bias2 = a.bias_correction ? 1.0f - pow_beta(tw, a.log2_beta2) : 1.0f;

  float step[KMAX];
  if (TYPE != 0) {
    float norm = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) norm += g[k] * g[k];
    norm = group_sum<LPP>(norm);
    float v = 0.f;
    if (live) {
      v = lerp_t(b2w, a.v[idx], norm);
      if (sub == 0) a.v[idx] = v;
    }
    const float scale = KIND == 0 ? (a.bias_correction ? sqrtf(bias2) / bias1 : 1.0f) * a.lr / fmaxf(sqrtf(v), a.eps)
                                  : 1.0f / fmaxf(sqrtf(v / bias2), a.eps);
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int j = sub + k * LPP;
      step[k] = 0.f;
      if (live && j < d) {
        float* mp = a.m + idx * d + j;
        if (KIND == 0) {
          const float m = lerp_t(b1w, *mp, g[k]);
          step[k] = m * scale;
          *mp = m;
        } else {
          const float m = lerp_t(b1w, *mp, g[k] * scale);
          step[k] = m * a.lr / bias1;
          *mp = m;
        }
      }
    }
  } else {
    const float bias_factor = a.bias_correction ? sqrtf(bias2) / bias1 : 1.0f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int j = sub + k * LPP;
      step[k] = 0.f;
      if (live && j < d) {
        float* mp = a.m + idx * d + j;
        float* vp = a.v + idx * d + j;
        const float v = lerp_t(b2w, *vp, g[k] * g[k]);
        float m;
        if (KIND == 0) {
          m = lerp_t(b1w, *mp, g[k]);
          step[k] = m / fmaxf(sqrtf(v), a.eps) * bias_factor * a.lr;
        } else {
          m = lerp_t(b1w, *mp, g[k] / fmaxf(sqrtf(v / bias2), a.eps));
          step[k] = m * a.lr / bias1;
        }
        *mp = m;
        *vp = v;
      }
    }
  }

  if (a.clip >= 0.f) {
    const float max_step = a.lr * a.clip;
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
      if (step[k] == step[k]) step[k] = fminf(fmaxf(step[k], -max_step), max_step);   // NaN stays NaN -> zeroed below
  }
  if (TYPE == 2) {
    const int base = (int)(threadIdx.x & 63) - sub;
    const float s0 = __shfl(step[0], base + 0, 64), s1 = __shfl(step[0], base + 1, 64), s2 = __shfl(step[0], base + 2, 64);
    const int r = sub < 3 ? sub : 0;
    step[0] = B[r][0] * s0 + B[r][1] * s1 + (d > 2 ? B[r][2] * s2 : 0.f);
  }
  const float sat = 1.0f - expf(-2.0f * w);
  const float plr = (live && a.point_lr) ? a.point_lr[idx] : 1.f;
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    const int j = sub + k * LPP;
    if (live && j < d) {
      float sj = step[k] * plr;
      if (a.mask_lr) sj *= a.mask_lr[j];
      if (a.out_step) { a.out_step[i * d + j] = sj; continue; }
      if (!(fabsf(sj) < 3.0e38f)) sj = 0.f;          // non-finite -> 0 (fractional.py:153)
      a.param[idx * d + j] -= sj * sat;
    }
  }
}

}  // namespace ms

using namespace ms;

static int launch_update(int kind, int group_type, const UpdateArgs& a, hipStream_t s) {
  const int d = a.d;
  const int64_t m_count = a.m_count;
  const int shape = group_type == 2 ? 1 : (d == 1 ? 0 : d <= 4 ? 1 : d <= 16 ? 2 : d <= 64 ? 3 : 4);
  const int lpp = shape == 0 ? 1 : shape == 1 ? 4 : 16;
  const dim3 block(256), grid((unsigned)div_up(m_count * lpp, 256));
#define MS_GO(K, T, L, KM) fractional_update_kernel<K, T, L, KM><<<grid, block, 0, s>>>(a)
#define MS_LPP(K, T)                                                                                    \
  switch (shape) { case 0: MS_GO(K, T, 1, 1); break; case 1: MS_GO(K, T, 4, 1); break;                  \
                   case 2: MS_GO(K, T, 16, 1); break; case 3: MS_GO(K, T, 16, 4); break;                \
                   default: MS_GO(K, T, 16, 16); break; }
  if (kind == 0) {
    if (group_type == 0) { MS_LPP(0, 0) } else if (group_type == 1) { MS_LPP(0, 1) } else { MS_GO(0, 2, 4, 1); }
  } else {
    if (group_type == 0) { MS_LPP(1, 0) } else if (group_type == 1) { MS_LPP(1, 1) } else { MS_GO(1, 2, 4, 1); }
  }
#undef MS_LPP
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
    UpdateArgs a{nullptr, grad, m, v, indexes, weight, total_weight, nullptr, nullptr, nullptr, nullptr,
                 m_count, d, lr, beta1, beta2, eps, -1.0f, bias_correction,
                 (float)log2((double)beta1), (float)log2((double)beta2), lr_step};
    return launch_update(kind, vector ? 1 : 0, a, (hipStream_t)stream);
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

extern "C" int ms_fractional_update(int kind, int group_type, float* param, const float* grad, float* m, float* v,
                                    const int64_t* indexes, const float* weight, const float* total_weight,
                                    const float* grad_scale, const float* basis, const float* mask_lr,
                                    const float* point_lr, int64_t m_count, int d, float lr, float beta1,
                                    float beta2, float eps, float clip, int bias_correction, void* stream) {
  MS_CHECK_ARG(kind == 0 || kind == 1, "kind must be 0 (Adam) or 1 (LaProp)");
  MS_CHECK_ARG(group_type >= 0 && group_type <= 2, "group_type must be 0 (scalar), 1 (vector) or 2 (local_vector)");
  MS_CHECK_ARG(m_count >= 0 && d >= 1 && d <= 256, "m_count >= 0 and 1 <= d <= 256 expected");
  if (m_count == 0) return 0;
  MS_CHECK_ARG(param && grad && m && v && indexes && weight && total_weight, "null pointer");
  if (group_type == 2) {
    MS_CHECK_ARG(basis, "local_vector groups need the basis");
    MS_CHECK_ARG(d == 2 || d == 3, "local_vector groups are 2 or 3 dimensional");
  }
  UpdateArgs a{param, grad, m, v, indexes, weight, total_weight, grad_scale, basis, mask_lr, point_lr,
               m_count, d, lr, beta1, beta2, eps, clip, bias_correction,
               (float)log2((double)beta1), (float)log2((double)beta2), nullptr};
  return launch_update(kind, group_type, a, (hipStream_t)stream);
}
