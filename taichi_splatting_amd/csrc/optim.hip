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

}  // namespace ms

using namespace ms;

extern "C" int ms_fractional_step(int kind, int vector, float* lr_step, const int64_t* indexes,
                                  const float* weight, float* m, float* v, const float* total_weight,
                                  const float* grad, int64_t m_count, int d, float lr, float beta1,
                                  float beta2, float eps, int bias_correction, void* stream) {
  MS_CHECK_ARG(kind == 0 || kind == 1, "kind must be 0 (adam) or 1 (laprop)");
  MS_CHECK_ARG(m_count >= 0 && d >= 1, "bad sizes");
  if (m_count == 0) return 0;
  MS_CHECK_ARG(lr_step && indexes && weight && m && v && total_weight && grad, "null pointer");
  const dim3 block(256), grid((unsigned)div_up(m_count, 256));
  hipStream_t s = (hipStream_t)stream;
#define MS_GO(K, V) fractional_step_kernel<K, V><<<grid, block, 0, s>>>(lr_step, indexes, weight, m, v, total_weight, grad, m_count, d, lr, beta1, beta2, eps, bias_correction)
  if (kind == 0) { if (vector) MS_GO(0, true); else MS_GO(0, false); }
  else { if (vector) MS_GO(1, true); else MS_GO(1, false); }
#undef MS_GO
  MS_CHECK_LAUNCH();
  return 0;
}
