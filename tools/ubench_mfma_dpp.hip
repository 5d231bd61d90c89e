// Does MFMA issue hide behind a DPP scan chain?  (the raster backward's per-splat moment sums
// sum_p q {1, x, y, x^2, xy, y^2} and sum_p w G_c are a (splats x pixels) . (pixels x 9) contraction; float32 MFMA has
// the vector FP32 rate, so the only possible gain is issue-port relief IF v_mfma_f32_4x4x1 issues beside the scans.)
// One iteration models one pixel step of raster_bwd_scan_kernel: two 6-level DPP scans (v_mul / v_add) + ~20 plain VALU
// instructions, then EITHER the 11 accumulation instructions of today (2 mul, 3 add, 6 fma) OR three
// v_mfma_f32_4x4x1_16b_f32 (A = per-pixel constants, B = per-lane q / w: D[block][i][j] = A[i] B[j], lane = 4 block + j:
// four columns per instruction).  Launch: 4 waves per SIMD, as the kernel runs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_dpp.hip -o tools/ubench_mfma_dpp.bin && tools/ubench_mfma_dpp.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float vec4 __attribute__((ext_vector_type(4)));

#define SCAN_STEP(OP, CTRL) OP " %0, %0, %0 " CTRL "\n\ts_nop 1\n\t"
#define SCAN(OP, v) asm volatile("s_nop 1\n\t" SCAN_STEP(OP, "row_shr:1 row_mask:0xf bank_mask:0xf") SCAN_STEP(OP, "row_shr:2 row_mask:0xf bank_mask:0xf") \
  SCAN_STEP(OP, "row_shr:4 row_mask:0xf bank_mask:0xf") SCAN_STEP(OP, "row_shr:8 row_mask:0xf bank_mask:0xf") \
  SCAN_STEP(OP, "row_bcast:15 row_mask:0xa bank_mask:0xf") OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v))

template <int MODE>   // 0: scans + common VALU only; 1: + 11 VALU accumulation; 2: + 3 MFMA accumulation
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  float x = threadIdx.x * 1e-3f + 0.5f, y = 1.0f - x * 0.25f;
  float m0 = 0, m1 = 0, m2 = 0, m3 = 0, m4 = 0, m5 = 0, a0 = 0, a1 = 0, a2 = 0;
  vec4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  const float c0 = (threadIdx.x & 3) * 0.5f, c1 = 1.0f + (threadIdx.x & 3), c2 = 0.25f;
  for (int i = 0; i < iters; ++i) {
    // "common" part of a pixel step: exp, gate, products (values kept in a sane range)
    float e = __builtin_amdgcn_exp2f(-(x * x + y * y));
    float a = e > 0.004f ? e : 0.0f;
    a = __builtin_fminf(a, 0.99f);
    float om = 1.0f - a, T = om;
    SCAN("v_mul_f32_dpp", T);
    float w = a * T, fG = x * 0.3f + y * 0.2f + 0.1f, S = w * fG;
    SCAN("v_add_f32_dpp", S);
    float ag = T * fG - S * __builtin_amdgcn_rcpf(om), q = ag * a;
    if (MODE == 1) {
      const float qX = q * x, qY = q * y;
      m0 += q; m1 += qX; m2 += qY;
      m3 = __builtin_fmaf(qX, x, m3); m4 = __builtin_fmaf(qX, y, m4); m5 = __builtin_fmaf(qY, y, m5);
      a0 = __builtin_fmaf(w, 0.3f, a0); a1 = __builtin_fmaf(w, 0.2f, a1); a2 = __builtin_fmaf(w, 0.1f, a2);
    } else if (MODE == 2) {
      acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(c0, q, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(c1, q, acc1, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(c2, w, acc2, 0, 0, 0);
    }
    x = x * 0.999f + 0.0007f; y = y * 0.998f + 0.0011f;
  }
  out[blockIdx.x * 256 + threadIdx.x] = m0 + m1 + m2 + m3 + m4 + m5 + a0 + a1 + a2 + acc0.x + acc0.y + acc0.z + acc0.w +
                                        acc1.x + acc1.y + acc1.z + acc1.w + acc2.x + acc2.y + acc2.z + acc2.w + x + y;
}

template <int MODE> static float run(float* out, int iters) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  const dim3 g(256 * 4), t(256);          // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  k<MODE><<<g, t>>>(out, iters);
  hipEventRecord(s);
  for (int r = 0; r < 5; ++r) k<MODE><<<g, t>>>(out, iters);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  return ms / 5;
}

int main() {
  float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
  const int iters = 20000;
  const float t0 = run<0>(out, iters), t1 = run<1>(out, iters), t2 = run<2>(out, iters);
  // SIMD cycles per iteration per wave slot: time x clock / iters / (waves per SIMD)
  printf("per pixel step (ns per iteration, 4 waves per SIMD): scans + common %.2f   + 11 VALU accumulate %.2f   + 3 MFMA 4x4x1 accumulate %.2f\n",
         t0 * 1e6 / iters, t1 * 1e6 / iters, t2 * 1e6 / iters);
  printf("accumulation costs %.2f ns as VALU, %.2f ns as MFMA: %+.1f %% of the step\n", (t1 - t0) * 1e6 / iters, (t2 - t0) * 1e6 / iters,
         100.0 * (t2 - t1) / t1);
  return 0;
}
