// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction for the op classes the
// raster kernels are made of.  hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float float2_ __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
  float a[8];
  float2_ p[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x * 0.001f + i; p[i] = float2_{a[i], a[i] * 0.5f}; }
  const float c1 = seed * 0.999f, c2 = seed * 0.0001f;
  const unsigned long long mask = 0xAAAAAAAAAAAAAAAAull + (unsigned long long)(seed > 2.f);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = __builtin_fmaf(a[i], c1, c2);
      if (OP == 1) p[i] = __builtin_elementwise_fma(p[i], float2_{c1, c1}, float2_{c2, c2});
      if (OP == 2) a[i] = a[i] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0xB1, 0xf, 0xf, true));
      if (OP == 3) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(a[i]) : "v"(a[i]), "v"(c1), "s"(mask));
      if (OP == 19) asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(c1) : );
      if (OP == 20) asm volatile("v_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(a[i]));
      if (OP == 21) { asm volatile("v_cndmask_b32_e32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(a[i]), "v"(c1) : ); asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[(i+4)&7]) : "v"(c1), "v"(c2)); }
      if (OP == 10) asm volatile("v_add_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c1));
      if (OP == 11) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(a[(i + 3) & 7]));
      if (OP == 12) a[i] = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a[i]), 0x041F));
      if (OP == 13) a[i] = __int_as_float(__builtin_amdgcn_ds_bpermute((threadIdx.x ^ 16) << 2, __float_as_int(a[i])));
      if (OP == 14) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c1));
      if (OP == 15) asm volatile("v_add_f32_dpp %0, %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(a[(i + 3) & 7]), "v"(a[(i + 5) & 7]));
      if (OP == 16) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(c1), "v"(c2));
      if (OP == 17) asm volatile("v_add_f32_dpp %0, %1, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(a[(i + 3) & 7]), "v"(a[(i + 5) & 7]));
      if (OP == 18) asm volatile("v_max_f32 %0, %1, %2" : "=v"(a[i]) : "v"(a[i]), "v"(c1));
      if (OP == 4) a[i] = __builtin_amdgcn_exp2f(a[i]);
      if (OP == 5) a[i] = a[i] + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x114, 0xf, 0xf, true));
      if (OP == 6) { auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[i]), __float_as_uint(a[(i + 1) & 7]), false, false); a[i] = __uint_as_float(r[0]); }
      if (OP == 7) a[i] = __builtin_amdgcn_rcpf(a[i]);
      if (OP == 8) a[i] = a[i] * c1;
      if (OP == 9) p[i] = p[i] * float2_{c1, c1};
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP> void run(const char* name, float* out) {
  const int iters = 4000, blocks = 256 * 8;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(out, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(out, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)blocks * 4 /*waves*/ * iters * 8;
  const double per_simd_s = insts / 1024.0 / (ms * 1e-3);
  printf("%-28s %8.3f ms  %.3f ns/inst/SIMD  (= %.2f cycles @2.4GHz, %.2f @2.0GHz)\n", name, ms, 1e9 / per_simd_s,
         2.4e9 / per_simd_s, 2.0e9 / per_simd_s);
}

int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  run<0>("v_fma_f32", out); run<1>("v_pk_fma_f32", out); run<8>("v_mul_f32", out); run<9>("v_pk_mul_f32", out);
  run<2>("v_add_f32_dpp quad_perm", out); run<5>("v_add_f32_dpp row_shr", out); run<3>("v_cndmask", out);
  run<10>("v_add_f32 (asm)", out); run<16>("v_fmac_f32 (asm)", out); run<18>("v_max_f32 (asm)", out); run<15>("v_add_f32_dpp quad (asm,indep)", out); run<17>("v_add_f32_dpp shr4 (asm,indep)", out);
  run<19>("v_cndmask_e32 vcc (asm)", out); run<20>("v_add_dpp self-dependent", out); run<21>("cndmask_e32 + fmac pair", out);
  run<11>("v_mov_b32_dpp (asm)", out); run<12>("ds_swizzle", out); run<13>("ds_bpermute", out); run<14>("v_mul_lo_u32", out);
  run<4>("v_exp_f32", out); run<7>("v_rcp_f32", out); run<6>("v_permlane32_swap", out);
  return 0;
}
