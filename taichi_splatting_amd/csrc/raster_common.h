// raster_common.h — pieces shared by the raster kernels (raster_fast.hip, and the float commit of raster.hip):
// kernel parameters, the two-deep gather pipeline's raw record, the LDS blend / cull records, the
// conservative rectangle-vs-contribution-ellipse test and the wave64 halving butterfly.
#pragma once
#include <stdlib.h>
#include "common.h"

namespace ms {

struct FastParams {
  int width, height, tiles_wide, tile_begin;
  float clamp_max_alpha, alpha_threshold, one_minus_saturate;
  int deterministic;      // raster_bwd_scan.hip: order-independent (fixed-point integer) gradient commits
  int num_tiles;          // tiles of this launch (xcd_tile)
  int grad_broadcast;     // raster_bwd_scan.hip: dL/dimage is ONE pixel's f values, the same for every pixel (a sum /
                          // mean loss hands autograd an expanded scalar: no (H, W, f) copy is made or read)
  // long tile runs cut into segments (below): 0 = off; > 0: the per-tile launch leaves tiles with longer runs to the
  // per-segment launch, which reads the plan
  int split_min_run;
  const int4* split_items;        // (tile, first entry, last entry + 1, index of the tile's first item)
  const int32_t* split_counts;    // [0] items, [1] long tiles, [2] plan overflow (capacities are bounds for k_capacity >= K;
                                  // a caller that lied about K: every tile falls back to its per-tile workgroup)
  float4* split_state;            // (item, pixel in the forward kernel's thread order): forward (C, P) of the segment,
                                  // then (colour in front of the segment, transmittance at its start)
  int32_t* long_run_word;         // pinned host word (may be NULL): the length of a run above SPLIT_MIN_RUN is noted there
};

// Long tile runs (round 5).  The raster kernels run ONE workgroup per tile; a scene that piles its splats onto a few
// tiles (a zoomed-out view: 2.3 M overlaps on nine tiles in tools/sweep_scenes.py) then uses a few of the chip's 256
// CUs.  The blending forward is a composition of affine maps per pixel — T' = T (1 - a), C' = C + f a T
// (rasterizer/forward.py:99-110) — so a run may be cut anywhere: a workgroup blends segment s from (C, T) = (0, 1) and
// leaves (C_s, P_s) per pixel; one pass per long tile composes them front to back, C = sum_s (prod_{r<s} P_r) C_s,
// T = prod_s P_s, writes the pixel, and leaves in place of (C_s, P_s) what the BACKWARD needs to start its walk at
// segment s: the colour in front of it and the transmittance there (backward.py:131-136 keeps exactly this running
// state).  No gate depends on T in the forward; the backward's saturation test (T against 1 - saturate_threshold)
// sees a T that was rounded in a different order — the same class of deviation as a pair on the blend gate.
constexpr int SPLIT_MIN_RUN = 16384;   // DEFAULT: runs above this are cut (and reported to the host: long_run_word) ...
constexpr int SPLIT_MAX_SEG = 256;     // ... into at most this many segments per tile ...
// ... of at least seg_len entries (a multiple of 256: of every batch size).  Short segments = many workgroups: the pile-up's
// 2.35 M overlaps make 576 workgroups at 4096 entries (2 per CU: one wave per SIMD, latency-bound) and 2300 at 1024.
// A tile-32 workgroup has 16 waves and 16 KB of state per segment: four times the length.
// Both numbers are RUN-TIME parameters of a (forward, backward) pair since round 6 (ms_raster_fwd_split's
// split_min_run / split_seg_len, ms_frame_desc.split_long_runs / split_seg_len; 0 = the defaults here), so that the
// oracle can reach the segment kernels on scenes it finishes in seconds.  MS_SPLIT_SEG in the environment (read once)
// overrides the default segment length, for measurements.
static inline int split_default_seg_len(int tile_size) {
  static const int forced = [] { const char* e = getenv("MS_SPLIT_SEG"); const int v = e ? atoi(e) : 0; return v >= 256 ? (v + 255) & ~255 : 0; }();
  if (forced) return forced;
  return tile_size == 32 ? 4096 : 1024;
}
struct SplitParams { int min_run, seg_len; };
// min_run_req <= 1 / seg_req <= 0: the defaults.  A run is cut only if it is longer than min_run, and never below 256
static inline SplitParams split_params(int tile_size, int min_run_req, int seg_req) {
  SplitParams sp;
  sp.min_run = min_run_req > 1 ? (min_run_req < 256 ? 256 : min_run_req) : SPLIT_MIN_RUN;
  sp.seg_len = seg_req > 0 ? ((seg_req + 255) & ~255) : split_default_seg_len(tile_size);
  return sp;
}
// the scratch block of one (forward, backward) pair: plan + per-(item, pixel) state, carved from caller memory
struct SplitScratch {
  int32_t* counts;      // 4 words: [0] items, [1] long tiles, [2] plan overflow, [3] unused
  int4* long_tiles;     // long_cap x (tile, first item, segments, 0)
  int4* items;          // item_cap
  float4* state;        // item_cap x tile_size^2
  int64_t long_cap, item_cap;
  int min_run, seg_len;
};
static inline int64_t split_long_capacity(int64_t k_capacity, SplitParams sp) { return k_capacity / sp.min_run + 1; }
static inline int64_t split_item_capacity(int64_t k_capacity, SplitParams sp) { return k_capacity / sp.seg_len + split_long_capacity(k_capacity, sp) + 1; }
static inline size_t split_scratch_bytes(int64_t k_capacity, int tile_size, SplitParams sp) {
  const size_t lc = (size_t)split_long_capacity(k_capacity, sp), ic = (size_t)split_item_capacity(k_capacity, sp);
  return 256 + ((lc * 16 + 255) & ~(size_t)255) + ((ic * 16 + 255) & ~(size_t)255) + ic * (size_t)tile_size * tile_size * 16;
}
static inline SplitScratch split_scratch_carve(void* base, int64_t k_capacity, int tile_size, SplitParams sp) {
  SplitScratch sc;
  char* p = (char*)base;
  sc.min_run = sp.min_run; sc.seg_len = sp.seg_len;
  sc.long_cap = split_long_capacity(k_capacity, sp); sc.item_cap = split_item_capacity(k_capacity, sp);
  sc.counts = (int32_t*)p; p += 256;
  sc.long_tiles = (int4*)p; p += ((size_t)sc.long_cap * 16 + 255) & ~(size_t)255;
  sc.items = (int4*)p; p += ((size_t)sc.item_cap * 16 + 255) & ~(size_t)255;
  sc.state = (float4*)p;
  return sc;
}

// Workgroups are dealt round-robin to the 8 XCDs, each with its own L2.  Neighbouring tiles share splats (a gaussian
// overlaps 2.1 tiles on config D; the four quarter workgroups of a 32 x 32 tile share ALL of them), so an XCD may
// take runs of CHUNK consecutive tiles instead of every eighth tile: block b -> XCD b % 8, position b / 8 in that
// XCD's sequence; `parts` consecutive positions belong to one tile (*part = which); run r of the XCD is tile run
// r * 8 + xcd.  CHUNK = 0: plain order.  Measured on config D (tile 8 / 16 / 32, ms):
//   forward   plain 0.73 / 0.667 / 0.78   CHUNK 2: 0.68 / 0.658 / 0.78   CHUNK 8: 0.65 / 0.651 / 0.78   CHUNK 32: 0.66 / 0.655 / 0.79
//   backward  plain 1.69 / 1.525 / 3.09   CHUNK 2: 1.68 / 1.532 / 3.04   CHUNK 8: 1.70 / 1.564 / 3.05   CHUNK 32: 1.73 / 1.600 / 3.04
//   (one contiguous eighth of the image per XCD: backward 1.64 at tile 16 — the eight bands are not equally heavy)
// The forward (bound by VALU, LDS and the gathers together) gains from the shared L2 lines; the backward does not
// (its gathers hide behind the blend phases, and tiles that share splats commit to the same moments rows).
// Launch xcd_grid<CHUNK>() blocks; xcd_tile() returns -1 for the padding blocks.
constexpr unsigned NUM_XCD = 8;
template <unsigned CHUNK>
__device__ __forceinline__ int xcd_tile(int num_tiles, unsigned block, unsigned parts, unsigned* part) {
  if constexpr (CHUNK == 0) {
    *part = block % parts;
    return block / parts < (unsigned)num_tiles ? (int)(block / parts) : -1;
  } else {
    const unsigned xcd = block % NUM_XCD, pos = block / NUM_XCD;
    const unsigned tpos = pos / parts, run = tpos / CHUNK, within = tpos % CHUNK;
    const unsigned tile = (run * NUM_XCD + xcd) * CHUNK + within;
    *part = pos % parts;
    return tile < (unsigned)num_tiles ? (int)tile : -1;
  }
}
template <unsigned CHUNK>
static inline unsigned xcd_grid(int num_tiles, unsigned parts) {
  if (CHUNK == 0) return (unsigned)num_tiles * parts;
  const unsigned span = NUM_XCD * CHUNK;
  return ((unsigned)num_tiles + span - 1) / span * span * parts;
}

// raw per-splat data in flight between the gather and the LDS write (one batch ahead)
struct Raw {
  float g[7];
  float f[3];
  int id;
};

// (ROWS: the frame executor's splat-row table, common.h)
template <bool ROWS = false>
__device__ __forceinline__ Raw load_raw(const float* __restrict__ points, const float* __restrict__ feats, int id) {
  Raw r;
  if constexpr (ROWS) {
    const float4* row = reinterpret_cast<const float4*>(points) + (int64_t)id * (SPLAT_ROW / 4);
    const float4 a = row[0], b = row[1], c = row[2];
    r.g[0] = a.x; r.g[1] = a.y; r.g[2] = a.z; r.g[3] = a.w; r.g[4] = b.x; r.g[5] = b.y; r.g[6] = b.z;
    r.f[0] = c.x; r.f[1] = c.y; r.f[2] = c.z;
  } else {
    const float* g = points + (int64_t)id * 7;
    const float* f = feats + (int64_t)id * 3;
#pragma unroll
    for (int k = 0; k < 7; ++k) r.g[k] = g[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) r.f[k] = f[k];
  }
  r.id = id;
  return r;
}

// Staging arithmetic (once per (tile, splat) overlap and kernel: ~250 instructions per staged splat with the
// correctly rounded division / sqrtf / logf sequences of -fno-fast-math, 6 % of the backward's VALU instructions).
// The hardware forms are within 1 ulp: v_rcp_f32 with one Newton step for the basis (0.5-1 ulp, the basis enters
// every alpha), plain v_rcp_f32 / v_sqrt_f32 / v_log_f32 for the cull data, whose margins (x 1.001, + 0.01 px,
// x 1.002) are four orders of magnitude wider.
#ifndef MS_SLOW_STAGING
#define MS_SLOW_STAGING 0            // 1: the library forms (A/B builds, tools/build_variant.sh)
#endif
#if MS_SLOW_STAGING
__device__ __forceinline__ float rcp_newton(float x) { return 1.0f / x; }
__device__ __forceinline__ float stage_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float fast_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ float fast_log2(float x) { return log2f(x); }
#else
__device__ __forceinline__ float rcp_newton(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, r, 1.0f);          // NaN for x = 0 / inf (r = inf / 0): the step is skipped
  return fabsf(e) < 0.5f ? __builtin_fmaf(r, e, r) : r;
}
__device__ __forceinline__ float stage_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }
#endif
// cutoff radius of the contribution region alpha g > threshold in units of sigma: sqrt(2 ln(alpha / threshold)),
// NaN when alpha < threshold (every comparison of the hit tests then fails: culled)
__device__ __forceinline__ float cutoff_radius(float alpha, float alpha_threshold) {
  constexpr float TWO_LN2 = 1.38629436111989061883f;
  return fast_sqrt(TWO_LN2 * (fast_log2(alpha) - fast_log2(alpha_threshold)));
}

constexpr float EXP2_SCALE = -0.72134752044448170368f;   // -0.5 * log2(e)
constexpr float EXP2_BASIS_SCALE = 0.84932180028801904272f;   // sqrt(0.5 * log2(e))

// LDS records of one splat: blend record [mx my A B] [C D alpha f0] [f1 f2 isx isy] and cull record
// [cx cy ex ey] [A' B' C' D'] (inverse basis divided by the cutoff radius)
// FWD_FORM (forward kernel): A..D are stored pre-scaled by sqrt(log2(e) / 2) and the alpha slot holds
// -log2(alpha), so that alpha * g = exp2(-(X^2 + Y^2 - log2 alpha)) is two FMAs and one v_exp_f32; the
// backward needs the true X, Y and g and keeps the plain form.
// In FWD_FORM the mean is replaced by (cX, cY) = basis * (mean - origin), origin = the tile centre, so that
// X = fma(px - ox, A, fma(py - oy, B, -cX)): two FMAs per axis instead of two subtractions + MUL + FMA (FMA /
// MUL issue at 2.5 cycles, ADD / SUB at 3+).  Tile-centre-relative coordinates keep every product below
// ~16 / sigma, so the rounding of the expanded form stays ~1e-6 in X.
template <bool FWD_FORM = false>
__device__ __forceinline__ void write_records(const Raw& r, float alpha_threshold, float4* rec, float4* cull,
                                              float origin_x = 0.0f, float origin_y = 0.0f, float log2_alpha_bias = 0.0f) {
  const float basis_scale = FWD_FORM ? EXP2_BASIS_SCALE : 1.0f;
  const float mx = r.g[0], my = r.g[1], ax = r.g[2], ay = r.g[3], sx = r.g[4], sy = r.g[5], alpha = r.g[6];
  const float isx = rcp_newton(sx), isy = rcp_newton(sy);
  const float A = ax * isx, B = ay * isx, C = -ay * isy, D = ax * isy;
  if (FWD_FORM) {
    const float rx = mx - origin_x, ry = my - origin_y;
    rec[0] = make_float4((rx * A + ry * B) * basis_scale, (rx * C + ry * D) * basis_scale, A * basis_scale, B * basis_scale);
  } else {
    rec[0] = make_float4(mx, my, A, B);
  }
  // (FWD_FORM, log2_alpha_bias = log2(clamp_max_alpha): the exponential then yields alpha g / clamp_max, raster_fast.hip)
  rec[1] = make_float4(C * basis_scale, D * basis_scale, FWD_FORM ? log2_alpha_bias - fast_log2(alpha) : alpha, r.f[0]);
  // (the forward reads 40 of the record's 48 bytes; it keeps its spent-wave flags in the last word of the first records,
  // so its stagers leave words 10 and 11 alone)
  if (FWD_FORM) *reinterpret_cast<float2*>(&rec[2]) = make_float2(r.f[1], r.f[2]);
  else rec[2] = make_float4(r.f[1], r.f[2], isx, isy);
  // contribution ellipse  alpha * g > threshold  <=>  X^2 + Y^2 < gs^2, gs = sqrt(2 ln(alpha/thr))
  // (NaN when alpha < threshold: every comparison of the hit test fails and the splat is culled)
  const float gs = cutoff_radius(alpha, alpha_threshold) * 1.001f;
  const float v1x = ax * sx * gs, v1y = ay * sx * gs, v2x = -ay * sy * gs, v2y = ax * sy * gs;
  cull[0] = make_float4(mx, my, fast_sqrt(v1x * v1x + v2x * v2x) + 0.01f, fast_sqrt(v1y * v1y + v2y * v2y) + 0.01f);
  const float igs = stage_rcp(gs);
  cull[1] = make_float4(A * igs, B * igs, C * igs, D * igs);
}

// does the splat's contribution region possibly touch the rectangle of pixel centres with centre
// (rcx, rcy) and half size h (3.5 for an 8x8 patch)?  Conservative: false
// only if provably no pixel of the rectangle passes alpha * g > threshold.
__device__ __forceinline__ bool rect_hit(const float4 c0, const float4 c1, float rcx, float rcy, float h) {
  const float dx = rcx - c0.x, dy = rcy - c0.y;   // rectangle centre relative to the mean
  // rectangle axes: |d| <= extent + h
  bool hit = (fabsf(dx) <= c0.z + h) && (fabsf(dy) <= c0.w + h);
  // ellipse axes (unit circle in the normalised frame): |c| - e <= 1 (+ margin)
  const float p1 = c1.x * dx + c1.y * dy, e1 = (fabsf(c1.x) + fabsf(c1.y)) * h;
  const float p2 = c1.z * dx + c1.w * dy, e2 = (fabsf(c1.z) + fabsf(c1.w)) * h;
  hit = hit && (fabsf(p1) - e1 <= 1.002f) && (fabsf(p2) - e2 <= 1.002f);
  return hit;
}

__device__ __forceinline__ bool patch_hit(const float4 c0, const float4 c1, float rcx, float rcy) {
  return rect_hit(c0, c1, rcx, rcy, 3.5f);
}

template <int CTRL>
__device__ __forceinline__ float add_dpp(float keep, float send) {
  return keep + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), CTRL, 0xf, 0xf, true));
}

// index of the value whose total lane `lane` holds after wave_reduce16 (valid when (lane & 15) >= 12)
__device__ __forceinline__ int butterfly_slot(int lane) {
  return 4 * (2 * (lane >> 5) + ((lane >> 4) & 1)) + (lane & 3);
}

// Halving butterfly: v[0..15] -> total of value butterfly_slot(lane) in lanes with (lane & 15) >= 12.
// Quad stages: each lane keeps half of its values and hands the rest to its partner (2 v_cndmask +
// 1 v_add_dpp quad_perm per value pair); row_shr:4/8 finish the 16-lane rows; v_permlane16_swap /
// v_permlane32_swap (gfx950) halve again across rows and wave halves.
__device__ __forceinline__ float wave_reduce16(const float (&v)[16], bool b0, bool b1) {
  float r1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float keep = b0 ? v[2 * i + 1] : v[2 * i];
    const float send = b0 ? v[2 * i] : v[2 * i + 1];
    r1[i] = add_dpp<0xB1>(keep, send);                      // quad_perm:[1,0,3,2]
  }
  float r2[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float keep = b1 ? r1[2 * j + 1] : r1[2 * j];
    const float send = b1 ? r1[2 * j] : r1[2 * j + 1];
    r2[j] = add_dpp<0x4E>(keep, send);                      // quad_perm:[2,3,0,1]
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r2[j] = add_dpp<0x114>(r2[j], r2[j]);                   // row_shr:4
    r2[j] = add_dpp<0x118>(r2[j], r2[j]);                   // row_shr:8
  }
  const auto p0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r2[0]), __float_as_uint(r2[1]), false, false);
  const auto p1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(r2[2]), __float_as_uint(r2[3]), false, false);
  const float s0 = __uint_as_float(p0[0]) + __uint_as_float(p0[1]);
  const float s1 = __uint_as_float(p1[0]) + __uint_as_float(p1[1]);
  const auto p2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(s0), __float_as_uint(s1), false, false);
  return __uint_as_float(p2[0]) + __uint_as_float(p2[1]);
}

// Four values per lane -> their wave totals: lanes with (lane & 15) >= 12 of EVERY row end with the total of
// v[lane & 3].  Two quad stages as above (6 v_cndmask + 3 v_add_dpp), row_shr:4/8, then the rows and wave halves are
// summed with v_permlane16_swap / v_permlane32_swap of the value with itself: 17 VALU instructions for four sums
// (the forward's per-splat visibility: four hits share one reduction instead of 6 DPP adds each).
__device__ __forceinline__ float wave_reduce4(const float (&v)[4], bool b0, bool b1) {
  const float a = add_dpp<0xB1>(b0 ? v[1] : v[0], b0 ? v[0] : v[1]);          // quad_perm:[1,0,3,2]
  const float b = add_dpp<0xB1>(b0 ? v[3] : v[2], b0 ? v[2] : v[3]);
  float c = add_dpp<0x4E>(b1 ? b : a, b1 ? a : b);                            // quad_perm:[2,3,0,1]
  c = add_dpp<0x114>(c, c);                                                   // row_shr:4
  c = add_dpp<0x118>(c, c);                                                   // row_shr:8
  const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(c), __float_as_uint(c), false, false);
  const float s = __uint_as_float(p[0]) + __uint_as_float(p[1]);
  const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}

// v_min_f32 without the canonicalising v_max hipcc puts in front of fminf (inputs are never sNaN here).
// NOT directly after the transcendental instruction that produces an operand: see clamp_alpha().
__device__ __forceinline__ float min_f32(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}


// v_min_f32 with the wave-uniform operand read straight from its SGPR (min_f32() takes two
// VGPRs and costs a v_mov_b32 per use when one side is a kernel argument)
__device__ __forceinline__ float min_f32_uniform(float a, float uniform_b) {
  float r;
  asm("v_min_f32 %0, %2, %1" : "=v"(r) : "v"(a), "s"(uniform_b));
  return r;
}

// min(e, clamp) for e >= 0 as ONE compiler-visible instruction (v_med3_f32 e, 0, clamp).  The inline-assembly
// v_min_f32 above must not directly follow the v_exp_f32 that produces its operand: gfx950 needs a wait state between
// a transcendental result and its VALU use, the hazard recogniser does not look into inline assembly, and half of the
// lanes then read the stale register (seen in the forward's hit loop once nothing else was scheduled in between).
__device__ __forceinline__ float clamp_alpha(float e, float clamp_max_alpha) {
  return __builtin_amdgcn_fmed3f(e, 0.0f, clamp_max_alpha);
}

// Lanes of one wave hand data to each other through LDS (hit lists, accumulator rows).  LDS operations of a wave
// execute in order, but the COMPILER reasons per thread: without a fence it may keep a value this thread loaded earlier
// instead of re-reading what another lane stored.  Release + acquire at wavefront scope costs no instruction.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int TS> struct TileGeom {
  static constexpr int THREADS = TS * TS;
  static constexpr int BATCH = THREADS < 256 ? THREADS : 256;
  static constexpr int WAVES_WIDE = TS / 8;
};

}  // namespace ms
