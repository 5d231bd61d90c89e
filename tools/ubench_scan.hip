// Microbenchmark of the instruction classes raster_bwd_scan.hip is made of, on gfx950: DPP scan steps
// (row_shr / row_bcast / wave_shr), lane <-> SGPR moves, and LDS accumulation variants (ds_add_f32 vs plain
// read-modify-write, conflict-free vs row-indexed).  hipcc --offload-arch=gfx950 -O3 tools/ubench_scan.hip -o /tmp/ubs
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, const int* __restrict__ slots, int iters, float seed) {
  __shared__ float s_acc[4][256 * 12];
  float a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 0.001f + i;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = slots[lane];                 // 64 increasing pseudo-random slots in [0, 256)
  for (int i = threadIdx.x; i < 256 * 12; i += 256) { s_acc[0][i] = 0.f; s_acc[1][i] = 0.f; s_acc[2][i] = 0.f; s_acc[3][i] = 0.f; }
  __syncthreads();
  int sl = (int)seed + 5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_mul_f32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 3) & 7]));
      if (OP == 1) asm volatile("v_mul_f32_dpp %0, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 3) & 7]));
      if (OP == 2) asm volatile("v_mul_f32_dpp %0, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 3) & 7]));
      if (OP == 3) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[(i + 3) & 7]));
      if (OP == 4) {   // full dependent 6-step scan on one register
        asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                     "s_nop 1\n\tv_mul_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(a[i]));
      }
      if (OP == 5) { int s; asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s) : "v"(a[i]), "s"(sl)); asm volatile("v_add_f32 %0, %1, %0" : "+v"(a[(i + 1) & 7]) : "s"(s)); }
      if (OP == 6) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tv_writelane_b32 %0, %1, m0" : "+v"(a[i]) : "s"(sl), "s"(sl & 63) : "m0");
      if (OP == 8) { int s_; asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s_) : "v"(a[i]), "s"(sl)); sl ^= (s_ & 1); }   // result consumed by SALU only
      if (OP == 9) { int s_; asm volatile("v_readlane_b32 %0, %1, 63" : "=s"(s_) : "v"(a[i])); asm volatile("s_nop 3\n\tv_add_f32 %0, %1, %0" : "+v"(a[(i + 1) & 7]) : "s"(s_)); }
      if (OP == 7) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 3) & 7]), "v"(a[(i + 5) & 7]));
      // LDS accumulation of one value per lane into a per-splat row
      if (OP == 10) __hip_atomic_fetch_add(&s_acc[0][lane * 1 + i * 64], a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // conflict free
      if (OP == 11) __hip_atomic_fetch_add(&s_acc[0][slot * 9 + i], a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);            // row of 9, shared by the 4 waves
      if (OP == 12) { float* p = &s_acc[wave][slot * 9 + i]; *p = *p + a[i]; }                                                       // per-wave rows, b32 RMW
      if (OP == 13 && i < 3) { float4* p = reinterpret_cast<float4*>(&s_acc[wave][slot * 12 + i * 4]); float4 v = *p; v.x += a[i]; v.y += a[i + 1]; v.z += a[i + 2]; v.w += a[i + 3]; *p = v; }  // b128 RMW
      if (OP == 14) __hip_atomic_fetch_add(&s_acc[0][i * 257 + slot], a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);          // SoA
      if (OP == 15) { float* p = &s_acc[wave][i * 257 + slot]; *p = *p + a[i]; }                                                     // SoA per-wave RMW
    }
    sl = ((sl * 5 + 1) & 63);
  }
  __syncthreads();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s + s_acc[0][threadIdx.x] + s_acc[wave][threadIdx.x + 256];
}

template <int OP> void run(const char* name, float* out, const int* slots, int per_iter = 8) {
  const int iters = 2000, blocks = 256 * 4;          // 4 workgroups of 4 waves per CU = 4 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(out, slots, 10, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(out, slots, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)blocks * 4 /*waves*/ * iters * per_iter;
  const double per_simd_s = insts / 1024.0 / (ms * 1e-3);     // per SIMD (1024 SIMDs); x4 for "per CU" (LDS is per CU)
  printf("%-44s %8.3f ms  %7.2f cycles/op/SIMD @2.0GHz  (%7.2f cycles/op/CU)\n", name, ms, 2.0e9 / per_simd_s, 2.0e9 / per_simd_s / 4);
}

int main() {
  float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
  int h[64]; unsigned r = 12345; int pos = 0;
  for (int i = 0; i < 64; ++i) { r = r * 1664525u + 1013904223u; pos += 1 + (r >> 24) % 7; h[i] = pos & 255; }
  int* slots; hipMalloc(&slots, sizeof(h)); hipMemcpy(slots, h, sizeof(h), hipMemcpyHostToDevice);
  run<7>("v_fmac_f32 (reference)", out, slots);
  run<0>("v_mul_f32_dpp row_shr:1 (independent)", out, slots);
  run<1>("v_mul_f32_dpp row_bcast:15 (independent)", out, slots);
  run<2>("v_mul_f32_dpp row_bcast:31 (independent)", out, slots);
  run<3>("v_mov_b32_dpp wave_shr:1 (independent)", out, slots);
  run<4>("6-step dependent mul scan (per scan)", out, slots);
  run<5>("v_readlane + v_add sgpr", out, slots);
  run<8>("v_readlane -> SALU use", out, slots);
  run<9>("v_readlane + s_nop 3 + v_add sgpr", out, slots);
  run<6>("s_mov m0 + v_writelane", out, slots);
  run<10>("ds_add_f32 conflict-free", out, slots);
  run<11>("ds_add_f32 row[slot*9+k] shared", out, slots);
  run<14>("ds_add_f32 SoA [k*257+slot] shared", out, slots);
  run<12>("b32 read+add+write per-wave row[slot*9+k]", out, slots);
  run<15>("b32 read+add+write per-wave SoA", out, slots);
  run<13>("b128 read+add+write per-wave row[slot*12] (x3)", out, slots, 3);
  return 0;
}
