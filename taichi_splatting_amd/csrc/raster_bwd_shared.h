// raster_bwd_shared.h — building blocks of the splat-per-lane raster backward (raster_bwd_scan.hip; also included by the
// measured-and-dropped round-4 variant tools/experiments/raster_bwd_rows.hip): the wave-wide DPP prefix scans, the
// 48-byte LDS record of a staged splat and the conservative rectangle test.
#pragma once
#include "raster_common.h"

namespace ms {

constexpr int MOMENT_ROW = MS_MOMENT_ROW;     // floats per point in the moments buffer (64 B, line aligned)

// Inclusive prefix product / sum over the 64 lanes: row_shr:1,2,4,8 build the 16-lane row prefixes, row_bcast:15
// (rows 1, 3) and row_bcast:31 (rows 2, 3) carry the row totals — six DPP instructions.  Lanes without a source
// (and rows masked off) keep their value, which is the identity of the scan: exactly what v_*_dpp without
// bound_ctrl does when it writes in place.  Written in assembly because LLVM's DPP combiner does not treat
// 1.0f / 0.0f as identities of v_mul_f32 / v_add_f32 (it emits v_mov_b32 + v_mov_b32_dpp + v_mul_f32 per step).
//
// The scan runs on TWO independent registers (two adjacent pixels) with the dependency chains interleaved: a DPP
// read needs two wait states after the VALU write of its source, and the other chain's instruction + s_nop 0
// provide them (a single chain needs s_nop 1 per level and leaves the SIMD idle for them).
// Measured on config D (MI355X), pixels per step: 1 -> 1.65 ms, 2 -> 1.52 ms, 4 (one row of the sub-patch, no wait
// states at all) -> 1.80 ms: the fourfold live state costs 154 VGPRs and a wave per SIMD.
#define MS_SCAN2_STEP(OP, CTRL)                                                                  \
  OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
#define MS_SCAN2_ASM(OP)                                                                          \
  asm("s_nop 1\n\t"                                                                               \
      MS_SCAN2_STEP(OP, "row_shr:1 row_mask:0xf bank_mask:0xf")                                   \
      MS_SCAN2_STEP(OP, "row_shr:2 row_mask:0xf bank_mask:0xf")                                   \
      MS_SCAN2_STEP(OP, "row_shr:4 row_mask:0xf bank_mask:0xf")                                   \
      MS_SCAN2_STEP(OP, "row_shr:8 row_mask:0xf bank_mask:0xf")                                   \
      MS_SCAN2_STEP(OP, "row_bcast:15 row_mask:0xa bank_mask:0xf")                                \
      OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                               \
      OP " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf"                                    \
      : "+v"(a), "+v"(b))
__device__ __forceinline__ void wave_scan_mul2(float& a, float& b) { MS_SCAN2_ASM("v_mul_f32_dpp"); }
__device__ __forceinline__ void wave_scan_add2(float& a, float& b) { MS_SCAN2_ASM("v_add_f32_dpp"); }
#undef MS_SCAN2_ASM

#undef MS_SCAN2_STEP

typedef __fp16 half2_t __attribute__((ext_vector_type(2)));

// 48-byte LDS record of a staged splat:
//   [mx my A' B'] [C' D' half2(ex, ey) R] [-log2(alpha) f0 f1 f2]
// A'..D' = basis * s, s = sqrt(log2(e) / 2), so alpha g = exp2(-(X'^2 + Y'^2 - log2 alpha)).  Cull data (same
// contribution region alpha g > alpha_threshold as write_records()): (ex, ey) = axis-aligned half extents of the
// ellipse, rounded UP to fp16 (+inf beyond the fp16 range: such a splat passes the rectangle-axis test, the
// ellipse-axis tests still apply); R = s * cutoff radius, i.e. |X'| - (|A'| + |B'|) h <= R on the ellipse axes.
// Round 5: everything the two cull levels test sits in the FIRST TWO words (they read 32 bytes per splat, and the
// level-1 loop can hold the next group's words while it tests the current ones without leaving the register budget);
// the blend reads word 0, half of word 1 and word 2.
struct ScanRecord { float4 r0, r1, r2; };
__device__ __forceinline__ ScanRecord make_scan_record(const Raw& r, float alpha_threshold) {
  const float mx = r.g[0], my = r.g[1], ax = r.g[2], ay = r.g[3], sx = r.g[4], sy = r.g[5], alpha = r.g[6];
  const float isx = rcp_newton(sx), isy = rcp_newton(sy);        // staging arithmetic: see raster_common.h
  const float s = EXP2_BASIS_SCALE;
  ScanRecord out;
  out.r0 = make_float4(mx, my, ax * isx * s, ay * isx * s);
  const float nl2a = -fast_log2(alpha);
  const float gs = cutoff_radius(alpha, alpha_threshold) * 1.001f;           // NaN below the threshold: culled
  const float v1x = ax * sx * gs, v1y = ay * sx * gs, v2x = -ay * sy * gs, v2y = ax * sy * gs;
  float ex = (fast_sqrt(v1x * v1x + v2x * v2x) + 0.01f) * 1.002f, ey = (fast_sqrt(v1y * v1y + v2y * v2y) + 0.01f) * 1.002f;
  ex = ex > 6.0e4f ? __builtin_inff() : ex;      // cvt_pkrtz rounds toward zero: pre-inflated by 2^-9
  ey = ey > 6.0e4f ? __builtin_inff() : ey;
  const half2_t e = __builtin_amdgcn_cvt_pkrtz(ex, ey);
  out.r1 = make_float4(-ay * isy * s, ax * isy * s, __builtin_bit_cast(float, e), gs * s * 1.002f);
  out.r2 = make_float4(nl2a, r.f[0], r.f[1], r.f[2]);
  return out;
}
__device__ __forceinline__ void write_scan_record(const Raw& r, float alpha_threshold, float4* rec) {
  const ScanRecord q = make_scan_record(r, alpha_threshold);
  rec[0] = q.r0; rec[1] = q.r1; rec[2] = q.r2;
}

// conservative test: can the contribution region touch the rectangle of pixel centres with centre (rcx, rcy)
// and half size h?  (rect_hit() of raster_common.h on the packed record)
__device__ __forceinline__ bool scan_rect_hit(const float4 q0, const float4 q1, float rcx, float rcy, float h) {
  const half2_t e = __builtin_bit_cast(half2_t, q1.z);
  const float dx = rcx - q0.x, dy = rcy - q0.y;
  bool hit = (fabsf(dx) <= (float)e[0] + h) && (fabsf(dy) <= (float)e[1] + h);
  const float p1 = q0.z * dx + q0.w * dy, e1 = (fabsf(q0.z) + fabsf(q0.w)) * h;
  const float p2 = q1.x * dx + q1.y * dy, e2 = (fabsf(q1.x) + fabsf(q1.y)) * h;
  return hit && (fabsf(p1) - e1 <= q1.w) && (fabsf(p2) - e2 <= q1.w);
}

}  // namespace ms
