// What does the blend phase of raster_bwd_scan_kernel cost when NOTHING else runs, and how does that depend on the waves
// a SIMD holds?  ("recover the idle issue slots" — are they idle because waves are parked in
// the other phases, or does the instruction stream of the blend itself not issue any faster?)
//
// One iteration = one chunk of the product kernel's grid-moment form: 8 pixel-pair steps (X, Y, v_exp_f32, gate, clamp,
// two interleaved 6-level DPP scans, d(alpha), row sums), the row folds, the change of variables, and the pixel state
// read from / written back to LDS exactly as the kernel does (broadcast reads one step ahead, last lane writes).
// No staging, no culling, no barriers, no global memory: the time per chunk is the issue-bound floor of the phase.
// Occupancy is set by the dynamic LDS size (160 KB / waves per SIMD per workgroup of 4 waves).
//   MODE 0: as the kernel   1: scans replaced by one plain multiply / add each (what the DPP chains cost)
//   MODE 2: no LDS traffic for the pixel state (registers)   3: v_exp_f32 / v_rcp_f32 replaced by an FMA / a multiply
//   MODE 4: no gate / clamp / saturation test (v_cmp, v_cndmask, v_min gone)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Itaichi_splatting_amd/csrc -Iinclude tools/ubench_blend.hip -o tools/ubench_blend.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "raster_bwd_shared.h"

using namespace ms;

template <int MODE>
__global__ void __launch_bounds__(256) blend_only(float* out, int iters, float alpha_threshold, float clamp_max_alpha,
                                                  float oms, int* sink, unsigned long long* cycles) {
  const unsigned long long c_start = __builtin_readcyclecounter();
  extern __shared__ float4 smem[];
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  float4* s_pix = smem + wave * 80;                                   // 64 float4 + 64 floats per wave
  float* s_rg = reinterpret_cast<float*>(s_pix + 64);
  s_pix[lane] = make_float4(0.3f + 0.01f * lane, 0.5f, 0.7f - 0.005f * lane, 1.0f);
  s_rg[lane] = 0.4f + 0.001f * lane;
  const bool last_lane = lane == 63;
  // one splat per lane: a config-D-like basis (sigma ~ 2 px), means scattered around the sub-patch
  const float sig = 1.5f + 0.03f * lane, th = 0.1f * lane;
  const float s = EXP2_BASIS_SCALE / sig;
  const float A = __cosf(th) * s, B = __sinf(th) * s, C = -__sinf(th) * s * 0.8f, D = __cosf(th) * s * 0.8f;
  const float mx = 1.5f + 0.11f * (lane % 9) - 2.0f, my = 1.5f + 0.13f * (lane % 7) - 1.0f;
  const float nl2a = 1.0f + 0.02f * lane, f0 = 0.3f, f1 = 0.5f + 0.001f * lane, f2 = 0.2f;
  const uint32_t oms_bits = __float_as_uint(oms);
  float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    wave_lds_fence();
    const float fx = 0.5f + (float)(it & 3), fy = 0.5f;
    const float dx0 = fx - mx, dy0 = fy - my;
    const float X00 = A * dx0 + B * dy0, Y00 = C * dx0 + D * dy0;
    const float Xr[4] = {X00, X00 + B, __builtin_fmaf(B, 2.0f, X00), __builtin_fmaf(B, 3.0f, X00)};
    const float Yr[4] = {Y00, Y00 + D, __builtin_fmaf(D, 2.0f, Y00), __builtin_fmaf(D, 3.0f, Y00)};
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f, n4 = 0.f, n5 = 0.f, r0 = 0.f, r1 = 0.f, r2 = 0.f;
    const int pbase = 16 * (it & 3);
    constexpr int U = 2;
    float4 pg[U];
    float prg[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { pg[u] = s_pix[pbase + u]; prg[u] = s_rg[pbase + u]; }
#pragma unroll
    for (int i = 0; i < 16; i += U) {
      const int p = pbase + i;
      float4 cur[U];
      float RGin[U];
      bool any_alive = false;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        cur[u] = pg[u]; RGin[u] = prg[u];
        any_alive |= __float_as_uint(cur[u].w) > oms_bits;
        if (MODE != 2 && i + U < 16) { pg[u] = s_pix[p + U + u]; prg[u] = s_rg[p + U + u]; }
      }
      if (__ballot(any_alive) != 0) {
        float X[U], Y[U], a_gated[U], a[U], om[U], Tk[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int x = (i + u) & 3, y = i >> 2;
          X[u] = x == 0 ? Xr[y] : __builtin_fmaf(A, (float)x, Xr[y]);
          Y[u] = x == 0 ? Yr[y] : __builtin_fmaf(C, (float)x, Yr[y]);
          const float e_ = __builtin_fmaf(X[u], X[u], __builtin_fmaf(Y[u], Y[u], nl2a));
          const float a_raw = MODE == 3 ? __builtin_fmaf(e_, -0.01f, 0.5f) : __builtin_amdgcn_exp2f(-e_);
          a_gated[u] = MODE == 4 ? a_raw : (a_raw > alpha_threshold ? a_raw : 0.0f);
          a[u] = MODE == 4 ? a_gated[u] : min_f32_uniform(a_gated[u], clamp_max_alpha);
          om[u] = 1.0f - a[u];
          Tk[u] = MODE == 1 ? cur[u].w * om[u] : dpp_f32<0x138>(cur[u].w, om[u]);
        }
        if (MODE != 1) wave_scan_mul2(Tk[0], Tk[1]);
        float a_st[U];
        bool any_sat = false;
#pragma unroll
        for (int u = 0; u < U; ++u) { a_st[u] = a_gated[u]; if (MODE != 4) any_sat |= !(Tk[u] > oms); }
        if (MODE != 4 && __ballot(any_sat) != 0) {
          asm volatile("; saturation inside the chunk" ::: "memory");
#pragma unroll
          for (int u = 0; u < U; ++u) {
            const bool live = Tk[u] > oms;
            a[u] = live ? a[u] : 0.0f;
            a_st[u] = live ? a_gated[u] : 0.0f;
          }
        }
        float w[U], fG[U], S[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          w[u] = a[u] * Tk[u];
          fG[u] = __builtin_fmaf(f2, cur[u].z, __builtin_fmaf(f1, cur[u].y, f0 * cur[u].x));
          S[u] = w[u] * fG[u];
        }
        if (MODE != 1) wave_scan_add2(S[0], S[1]);
        float RGout[U];
#pragma unroll
        for (int u = 0; u < U; ++u) RGout[u] = RGin[u] - S[u];
        if (MODE != 2 && last_lane) {
          // (the benchmark keeps T at 1 so that every iteration blends: the write goes to a dummy slot of the wave)
          s_pix[64 + (p & 7)].w = Tk[0] * om[0];
          s_pix[64 + ((p + 1) & 7)].w = Tk[1] * om[1];
          *reinterpret_cast<float2*>(&s_rg[p]) = make_float2(RGin[0], RGin[1]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const float RGk = RGout[u];
          const float ag = __builtin_fmaf(Tk[u], fG[u], -(RGk * (MODE == 3 ? om[u] * 1.01f : __builtin_amdgcn_rcpf(om[u]))));
          const float q_ = ag * a_st[u];
          const int x = (i + u) & 3;
          r0 += q_;
          if (x == 1) { r1 += q_; r2 += q_; }
          if (x > 1) { r1 = __builtin_fmaf(q_, (float)x, r1); r2 = __builtin_fmaf(q_, (float)(x * x), r2); }
          a0 = __builtin_fmaf(w[u], cur[u].x, a0); a1 = __builtin_fmaf(w[u], cur[u].y, a1); a2 = __builtin_fmaf(w[u], cur[u].z, a2);
        }
      }
      if ((i & 3) == 2) {
        const int y = i >> 2;
        n0 += r0; n1 += r1; n3 += r2;
        if (y == 1) { n2 += r0; n4 += r1; n5 += r0; }
        if (y > 1) {
          n2 = __builtin_fmaf(r0, (float)y, n2); n4 = __builtin_fmaf(r1, (float)y, n4);
          n5 = __builtin_fmaf(r0, (float)(y * y), n5);
        }
        r0 = 0.f; r1 = 0.f; r2 = 0.f;
      }
    }
    const float P = __builtin_fmaf(A, n1, B * n2), Q = __builtin_fmaf(C, n1, D * n2);
    const float s1 = __builtin_fmaf(A, n3, B * n4), s2 = __builtin_fmaf(A, n4, B * n5);
    const float t1 = __builtin_fmaf(C, n3, D * n4), t2 = __builtin_fmaf(C, n4, D * n5);
    const float m0 = n0;
    const float m1 = __builtin_fmaf(X00, n0, P);
    const float m2 = __builtin_fmaf(Y00, n0, Q);
    const float m3 = __builtin_fmaf(X00, m1 + P, __builtin_fmaf(A, s1, B * s2));
    const float m5 = __builtin_fmaf(Y00, m2 + Q, __builtin_fmaf(C, t1, D * t2));
    const float m4 = __builtin_fmaf(X00, m2, __builtin_fmaf(Y00, P, __builtin_fmaf(A, t1, B * t2)));
    acc[0] += m0; acc[1] += m1; acc[2] += m2; acc[3] += m3; acc[4] += m4; acc[5] += m5; acc[6] += a0; acc[7] += a1; acc[8] += a2;
  }
  float sum = 0.f;
  for (int k = 0; k < 9; ++k) sum += acc[k];
  out[blockIdx.x * 256 + t] = sum;
  if (sum == 123.456f) *sink = 1;
  if (blockIdx.x == 0 && t == 0) *cycles = __builtin_readcyclecounter() - c_start;
}

static unsigned long long* g_cycles;
static double g_ghz;
template <int MODE> static float run(float* out, int* sink, int iters, int waves_per_simd) {
  hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
  // dynamic LDS so that exactly `waves_per_simd` workgroups (4 waves each, one per SIMD) fit a CU
  size_t lds = (160 * 1024) / waves_per_simd;
  lds = lds / 1024 * 1024;
  if (lds < 4 * 80 * 16) lds = 4 * 80 * 16;
  if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)blend_only<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const dim3 g(256 * waves_per_simd), t(256);
  blend_only<MODE><<<g, t, lds>>>(out, iters, 1.0f / 255.0f, 0.99f, 1e-4f, sink, g_cycles);
  hipEventRecord(s);
  for (int r = 0; r < 3; ++r) blend_only<MODE><<<g, t, lds>>>(out, iters, 1.0f / 255.0f, 0.99f, 1e-4f, sink, g_cycles);
  hipEventRecord(e); hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  unsigned long long c = 0; hipMemcpy(&c, g_cycles, 8, hipMemcpyDeviceToHost);
  g_ghz = (double)c / (ms / 3 * 1e6);        // shader cycles of one workgroup / kernel time (one resident set of workgroups)
  return ms / 3;
}

int main(int argc, char** argv) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  int* sink; hipMalloc(&sink, 4);
  hipMalloc(&g_cycles, 8);
  const int iters = 4000;
  printf("blend phase alone: ns per chunk (16 pixel steps) per wave slot = time / iters / waves_per_simd; cycles at the clock rocm-smi shows\n");
  printf("%-8s %14s %14s %14s %14s %14s\n", "waves", "kernel form", "no DPP scans", "no LDS state", "no exp / rcp", "no gate/clamp");
  for (int w : {1, 2, 3, 4, 5, 6, 8}) {
    const float t1 = run<1>(out, sink, iters, w), t2 = run<2>(out, sink, iters, w), tk = run<0>(out, sink, iters, w);
    const float t3 = run<3>(out, sink, iters, w), t4 = run<4>(out, sink, iters, w);
    (void)run<0>(out, sink, iters, w);
    printf("%-8d %11.1f ns %11.1f ns %11.1f ns %11.1f ns %11.1f ns   (per SIMD: a chunk every %.1f ns = %.0f cycles at the %.2f GHz s_memtime shows)\n", w,
           tk * 1e6 / iters, t1 * 1e6 / iters, t2 * 1e6 / iters, t3 * 1e6 / iters, t4 * 1e6 / iters, tk * 1e6 / iters / w, tk * 1e6 / iters / w * g_ghz, g_ghz);
    const float t0 = 0; (void)t0;
  }
  return 0;
}
