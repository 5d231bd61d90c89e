// raster_bwd_scan.hip — product-path raster backward (float32, RGB, plain gaussian pdf, alpha blending):
// semantics of rasterizer/backward.py:97-224, organised for wave64 so that NO per-splat cross-lane gradient
// reduction exists at all.
//
// The reference (and round 1 of this library) maps one thread to one pixel, walks the tile's splats in depth
// order and, for every splat, sums the 7 + 3 (+2) per-pixel gradient terms over the pixels of a warp / wave
// (backward.py:200-224; taichi_lib/concurrent.py:11-23,69-86).  With small splats only a few of the 64 pixels of
// a wave contribute to a given splat, so most lanes compute and reduce zeros.
//
// Here the roles are transposed:
//
//   * one workgroup per tile, one wave per 8x8 pixel patch = four 4x4 SUB-PATCHES;
//   * per staged batch (the tile's list in equal batches of <= 320 splats, LDS) every wave tests the splats against each of its sub-patches (one
//     splat per lane, same conservative oriented-box test as the tile mapper, grid_query.py:30-43) and compacts
//     the hits into per-sub-patch lists (ballot + mbcnt, order preserving => still depth sorted);
//   * a sub-patch list is consumed in chunks of 64 hits with ONE SPLAT PER LANE.  The 16 pixels of the
//     sub-patch are visited in pairs (two independent scan chains, interleaved); the pixel (its coordinates,
//     transmittance T, dL/dC and the colour still to come <R, G>) is wave-uniform: every lane reads it from the
//     same LDS address (broadcast), one step ahead of its use.  The front-to-back recurrence over the 64 splats of the chunk is two DPP prefix scans:
//         T_k  = T_in * prod_{j<k} (1 - a_j)            (multiplicative, exclusive: wave_shr:1 + 6 v_mul_f32_dpp)
//         S_k  = sum_{j<=k} w_j <f_j, G>                 (additive, inclusive: 6 v_add_f32_dpp)
//     so each lane knows the T and <R, G> its splat sees at this pixel, evaluates d(alpha) and accumulates ITS
//     splat's gradient in ITS OWN registers over the 16 pixels — no butterfly, no atomics in the loop;
//   * what a lane accumulates are the six moments  sum q, q X, q Y, q X^2, q X Y, q Y^2  (q = alpha g dL/dalpha,
//     (X, Y) = pixel in the splat's normalised frame) plus sum w G_c: every geometric gradient of
//     generic.py:321-336 is a per-splat LINEAR map of these sums, applied once per gaussian by
//     raster_moments_finalize_kernel (or by the fused projection/SH backward) instead of once per pixel.  Inside a
//     chunk the moments are first taken in the sub-patch's integer pixel grid (x, y in 0..3 are compile-time constants
//     of the unrolled steps: 75 instead of 128 instructions per chunk) and mapped to (X, Y) once per chunk (round 4);
//   * a chunk ends with a plain read-add-write of the lane's sums into the WAVE'S OWN LDS row of that splat (LDS
//     float atomics cost ~160 cycles per instruction on gfx950 and are avoided), and a pass over the staged batch
//     ends with ONE 64-byte, line-aligned row of global float atomics per (8x8 patch, splat): seven rows of nine
//     sums per wave instruction.
//
// VALU work per (sub-patch, splat) hit is ~16 pixel steps x ~46 instructions / (lanes filled) ~= 16 wave
// instructions, against ~126 per (8x8 patch, splat) hit of the pixel-per-lane kernel it replaces.
#include <stdlib.h>

#include <type_traits>

#include "raster_bwd_shared.h"
#include "frame_internal.h"

// Development builds (tools/abl): -DMS_SCAN_STATS counts chunks / filled lanes / executed pixel steps into
// g_scan_stats (read with ms_debug_scan_stats); -DMS_SCAN_ABLATE=1 skips the blend phase, =2 the cull + blend.
#ifndef MS_SCAN_STATS
#define MS_SCAN_STATS 0
#endif
#ifndef MS_SCAN_ABLATE
#define MS_SCAN_ABLATE 0
#endif
#ifndef MS_GRID_MOMENTS
#define MS_GRID_MOMENTS 1           // 0: per-pixel moment sums in the splat's frame (rounds 2-3), kept for A/B builds
#endif
// tile 32 (one 1024-thread workgroup per tile): staged splats per batch / accumulator rows per wave
// tile 8 (one wave per tile)
#ifndef MS_T8_BATCH
#define MS_T8_BATCH 128
#endif
#ifndef MS_T8_CAP
#define MS_T8_CAP 128
#endif
#ifndef MS_T32_BATCH
#define MS_T32_BATCH 896
#endif
#ifndef MS_T32_CAP
#define MS_T32_CAP 128
#endif

namespace ms {

#if MS_SCAN_STATS
__device__ unsigned long long g_scan_stats[12];
#endif

// Deterministic mode: the per-(patch, splat) sums are committed as 64-bit fixed-point integers with INTEGER atomics.
// Integer addition is associative, so the accumulated row does not depend on the order in which the patches of
// different tiles reach memory and the gradients are bitwise reproducible; everything before the commit (scans,
// per-lane sums, per-wave LDS rows) already runs in program order.  The unit is 2^-e with e chosen PER LAUNCH from
// max |dL/dimage| (ms_fixed_point_exponents; a fixed 2^-32 kept a few bits only of the gradients of a mean-reduced
// loss, dL/dC ~ 1e-7, and rounded the squared heuristic term to 0): e_main for the nine sums that are linear in
// dL/dimage and for split_score, e_h0 for prune_cost's sum of (dL/dalpha)^2.
constexpr int FIXED_POINT_BITS = 36;      // a commit of magnitude max|dL/dimage| is worth 2^36 units
__global__ void fixed_point_exponents_kernel(const float* __restrict__ amax, int32_t* __restrict__ out) {
  const float m = *amax;
  int e = 0;
  if (m > 0.0f && m < __builtin_inff()) (void)frexpf(m, &e);        // m = f * 2^e, f in [0.5, 1)
  int e_main = FIXED_POINT_BITS - e, e_h0 = FIXED_POINT_BITS - 2 * e;
  out[0] = e_main < -100 ? -100 : (e_main > 100 ? 100 : e_main);
  out[1] = e_h0 < -100 ? -100 : (e_h0 > 100 ? 100 : e_h0);
}

// SPLIT: the mapper's tile is (TS * SPLIT)^2 pixels and SPLIT^2 workgroups share its splat list, each taking one
// TS x TS quarter.  Tile 32 runs as <16, HEUR, 2>: with one 1024-thread workgroup per tile a staged splat touches
// few of the 16 patches and the per-wave lists stay short (4.0 ms on config D against 1.5 ms at tile 16); a
// quarter stages the whole list (2.8x the splats that touch it) and then works exactly like a 16 x 16 tile.
template <int TS, bool HEUR, int SPLIT = 1>
__global__ void __launch_bounds__(TS * TS)
raster_bwd_scan_kernel(const float* __restrict__ points, const float* __restrict__ feats,
                       const int32_t* __restrict__ ranges, const int32_t* __restrict__ o2p,
                       const float* __restrict__ image, const float* __restrict__ grad_image,
                       FastParams rp, float* __restrict__ moments, const int32_t* __restrict__ fixed_exp) {
  constexpr int THREADS = TS * TS, WAVES = THREADS / 64, WAVES_WIDE = TS / 8;
  // Splats staged per batch (shared by the tile's waves).  A tile's list is cut into EQUAL batches of about
  // BATCH_TARGET (see the batch loop): with fixed 256-splat batches config D's ~779 splats per tile end in an
  // 11-splat batch whose chunks run all 16 pixel steps for a couple of lanes — a quarter of all chunks.
  // TS == 32 (one 1024-thread workgroup per 32 x 32 tile, 16 waves): an 8x8 patch sees ~1/8 of the tile's splats, so
  // the batch is 896 splats for the per-wave lists to be as long as at tile 16 (~115 patch hits); 43 KB of records +
  // 16 x 6.6 KB per-wave state = 152 of the CU's 160 KB LDS: one workgroup per CU = the same 4 waves per SIMD
  constexpr int BATCH = TS == 8 ? MS_T8_BATCH : TS == 32 ? MS_T32_BATCH : (TS == 16 && !HEUR) ? 268 : 256;
  constexpr int BATCH_TARGET = TS == 8 ? MS_T8_BATCH - 16 : TS == 32 ? MS_T32_BATCH - 64 : 256;
  // Patch hits a wave takes on per pass (>= 64: a pass always advances) = rows of its accumulator.  A wave whose
  // patch list overflows runs the rest of the batch as a second pass with nearly empty chunks, so at tile 16 the
  // 40 KB a workgroup may use (four per CU) go to CAP first and to the staging batch second (config D, ms;
  // a batch of ~260 splats puts ~100 on an 8x8 patch):
  //   BATCH / CAP   320 / 112: 1.50    268 / 128: 1.43    256 / 132: 1.51    (3.27 / 3.04 / 3.04 at 4096^2)
  //   with heuristics (11 floats per row)   256 / 104: 1.73    256 / 110: 1.68    320 / 92: 2.23
  constexpr int CAP = TS == 16 ? (HEUR ? 110 : 128) : (TS == 32) ? (HEUR ? MS_T32_CAP - 24 : MS_T32_CAP) : MS_T8_CAP;
  constexpr int NACC = HEUR ? 11 : 9;
  constexpr bool GRID_MOMENTS = MS_GRID_MOMENTS != 0;            // see the blend loop
  // largest basis entry (A..D, in units of sqrt(log2 e / 2) / sigma per pixel) a chunk may hold and still use the grid
  // form: 2.5 <=> sigma 0.34 px, where the expansion costs ~3e-5 of the moments' scale (fuzz: tools/fuzz_raster_bwd.py)
  constexpr float GRID_MAX_BASIS = 2.5f;
  constexpr bool PIPELINED = THREADS >= 256;     // staged splats are gathered one batch ahead (slots t and PRIMARY + t)
  constexpr int PRIMARY = THREADS < BATCH ? THREADS : BATCH;      // slots filled by "thread t stages slot t"
  constexpr int SLOTS_B = PIPELINED ? BATCH - PRIMARY : 0;  // second slot of the first SLOTS_B threads
  static_assert(SLOTS_B >= 0 && SLOTS_B <= 64, "second staging slot: first wave only");
  // tile 16: 12.6 KB records + 1 KB ids + 4 x (4.5 KB accumulators + 0.75 KB lists + 1.25 KB pixels) = 39.9 KB: four
  // workgroups per CU
  __shared__ float4 s_rec[BATCH * 3];
  __shared__ int32_t s_id[BATCH];
  // PER-WAVE gradient accumulators, one row per splat of the wave's patch list: plain read-add-write, no LDS
  // atomics (ds_add_f32 costs ~160 LDS cycles per instruction on gfx950, tools/ubench_scan.hip)
  __shared__ float s_acc[WAVES][CAP][NACC];
  __shared__ uint16_t s_plist[WAVES][CAP];       // patch-list position -> staged index
  __shared__ uint8_t s_list[WAVES][4][CAP];      // per sub-patch: patch-list positions of its hits, depth ordered
  // per-pixel data, read by ALL lanes of the wave at the pixel's step (same address: LDS broadcast; v_readlane
  // from state registers costs ~12-16 cycles per value on gfx950, tools/ubench_scan.hip):
  // [dL/dC.rgb, T] and <R, G>; entry p = 16 * sub-patch + 4 * y + x
  __shared__ float4 s_pix[WAVES][64];
  __shared__ float s_rg[WAVES][64];

  unsigned quarter_u;
  // the quarter workgroups of a tile run on one XCD (they stage the same list); tiles themselves in plain order
  const int local_tile = xcd_tile<(SPLIT > 1 ? 1 : 0)>(rp.num_tiles, blockIdx.x, SPLIT * SPLIT, &quarter_u);
  if (local_tile < 0) return;
  const int tile_id = rp.tile_begin + local_tile;
  const int quarter = (int)quarter_u;
  const int tile_u = tile_id % rp.tiles_wide, tile_v = tile_id / rp.tiles_wide;
  const int t = threadIdx.x, wave = t >> 6, lane = lane_id();
  const int patch_x = (tile_u * SPLIT + quarter % SPLIT) * TS + (wave % WAVES_WIDE) * 8;
  const int patch_y = (tile_v * SPLIT + quarter / SPLIT) * TS + (wave / WAVES_WIDE) * 8;

  // pixel state: lane p = 16 * sub + 4 * y + x holds pixel (x, y) of sub-patch `sub` (sub-patches 2 x 2)
  const int sub = lane >> 4;
  const int pix_x = patch_x + (sub & 1) * 4 + (lane & 3), pix_y = patch_y + (sub >> 1) * 4 + ((lane >> 2) & 3);
  {
    float G0 = 0.f, G1 = 0.f, G2 = 0.f, RG = 0.f, T = 0.f;        // T = 0: out-of-image pixels never blend
    if (pix_x < rp.width && pix_y < rp.height) {
      const int64_t p = (int64_t)pix_y * rp.width + pix_x;
      const int64_t gp = rp.grad_broadcast ? 0 : p * 3;
      G0 = grad_image[gp + 0]; G1 = grad_image[gp + 1]; G2 = grad_image[gp + 2];
      RG = image[p * 3 + 0] * G0 + image[p * 3 + 1] * G1 + image[p * 3 + 2] * G2;   // <R, G>, R = forward image
      T = 1.0f;
    }
    s_pix[wave][lane] = make_float4(G0, G1, G2, T);
    s_rg[wave][lane] = RG;
  }
  const bool last_lane = lane == 63;
  float fixed_main = 0.f, fixed_h0 = 0.f;
  if (rp.deterministic) { fixed_main = ldexpf(1.0f, fixed_exp[0]); fixed_h0 = ldexpf(1.0f, fixed_exp[1]); }
  const float oms = rp.one_minus_saturate;
  const uint32_t oms_bits = __float_as_uint(oms);     // T >= 0: the float order is the order of the bit patterns

  // accumulators start at zero; the commit of a pass re-zeroes exactly what it read
  for (int i = lane; i < CAP * NACC; i += 64) (&s_acc[wave][0][0])[i] = 0.0f;

  const int start = ranges[tile_id * 2 + 0], end = ranges[tile_id * 2 + 1];

  // equal batches: as many as it takes to stay near BATCH_TARGET (rounded to nearest), never above BATCH
  const int total = end - start;
  int num_batches = (total + BATCH_TARGET / 2) / BATCH_TARGET;
  if (num_batches < 1) num_batches = 1;
  int bsz = (total + num_batches - 1) / num_batches;
  if (bsz > BATCH) { num_batches = (total + BATCH - 1) / BATCH; bsz = (total + num_batches - 1) / num_batches; }

  // two-deep gather pipeline per staging slot: `raw` = splat data of the batch about to be staged, `next_id` =
  // point index of the batch after it
  Raw raw, raw_b;
  int next_id = 0, next_id_b = 0;
  if (PIPELINED) {
    if (t < bsz && start + t < end) raw = load_raw(points, feats, o2p[start + t]);
    if (t < bsz && start + bsz + t < end) next_id = o2p[start + bsz + t];
    if (t < SLOTS_B) {
      if (PRIMARY + t < bsz && start + PRIMARY + t < end) raw_b = load_raw(points, feats, o2p[start + PRIMARY + t]);
      if (PRIMARY + t < bsz && start + bsz + PRIMARY + t < end) next_id_b = o2p[start + bsz + PRIMARY + t];
    }
  }

#if MS_SCAN_STATS
  __shared__ int s_bsum, s_bmax;
  if (t == 0) { s_bsum = 0; s_bmax = 0; }
#define MS_FLUSH_BALANCE()                                                                                      \
  if (t == 0) {                                                                                                 \
    atomicAdd(&g_scan_stats[8], (unsigned long long)s_bsum);                                                    \
    atomicAdd(&g_scan_stats[9], (unsigned long long)(s_bmax * WAVES));                                          \
    s_bsum = 0; s_bmax = 0;                                                                                     \
  }
#endif
  for (int begin = start; begin < end; begin += bsz) {
    const int count = (end - begin) < bsz ? (end - begin) : bsz;
    // all waves are done with the previous batch; tile-wide early out once every pixel is saturated
    // (backward.py:116)
    wave_lds_fence();
    const bool tile_done = __syncthreads_and(__float_as_uint(s_pix[wave][lane].w) <= oms_bits);
#if MS_SCAN_STATS
    MS_FLUSH_BALANCE()
#endif
    if (tile_done) break;

    if (PIPELINED) {
      if (t < count) {
        write_scan_record(raw, rp.alpha_threshold, &s_rec[t * 3]);
        s_id[t] = raw.id;
      }
      if (t < bsz && begin + bsz + t < end) raw = load_raw(points, feats, next_id);
      if (t < bsz && begin + 2 * bsz + t < end) next_id = o2p[begin + 2 * bsz + t];
      if (t < SLOTS_B) {
        const int sb = PRIMARY + t;
        if (sb < count) {
          write_scan_record(raw_b, rp.alpha_threshold, &s_rec[sb * 3]);
          s_id[sb] = raw_b.id;
        }
        if (sb < bsz && begin + bsz + sb < end) raw_b = load_raw(points, feats, next_id_b);
        if (sb < bsz && begin + 2 * bsz + sb < end) next_id_b = o2p[begin + 2 * bsz + sb];
      }
    } else {
      for (int s = t; s < count; s += THREADS) {
        const Raw r = load_raw(points, feats, o2p[begin + s]);
        write_scan_record(r, rp.alpha_threshold, &s_rec[s * 3]);
        s_id[s] = r.id;
      }
    }
    __syncthreads();

    // wave-wide early out (backward.py:142)
    if (__ballot(__float_as_uint(s_pix[wave][lane].w) > oms_bits) == 0) continue;
#if MS_SCAN_ABLATE == 2
    continue;
#endif

#if MS_SCAN_STATS
    // balance of the blend work between the waves of a workgroup: per batch, sum and WAVES x max of the chunks
    // the waves ran (a batch ends at a barrier, so the slowest wave sets its length)
    int batch_chunks = 0;
#endif
    // From here to the next barrier the wave works alone: it walks the staged batch in passes of at most CAP
    // patch hits (one pass per batch unless most staged splats touch this 8x8 patch).
    int r = 0;
    while (r < count) {
      // ---- cull, level 1: the staged splats that can touch this wave's 8x8 patch -> patch list ------------------
      int pcount = 0;
      const float pcx = (float)patch_x + 4.0f, pcy = (float)patch_y + 4.0f;
      while (r < count) {
        const int j = r + lane;
        const bool hit = j < count && scan_rect_hit(s_rec[j * 3 + 0], s_rec[j * 3 + 1], s_rec[j * 3 + 2], pcx, pcy, 3.5f);
        const unsigned long long m = __ballot(hit);
        const int nhit = __builtin_popcountll(m);
        if (pcount + nhit > CAP) break;          // next pass (nhit <= 64 <= CAP: an empty list always takes the group)
        const int ppos = pcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (hit) s_plist[wave][ppos] = (uint16_t)j;
        pcount += nhit;
        r += 64;
      }
      wave_lds_fence();
      // ---- cull, level 2: only the patch hits (about 40 % of a batch) are tested against the four 4x4 sub-patches;
      // both lists are appended in order, so they stay depth sorted
      int cnt[4] = {0, 0, 0, 0};
      for (int g = 0; g < pcount; g += 64) {
        const int ppos = g + lane;
        const bool in = ppos < pcount;
        const int j = in ? (int)s_plist[wave][ppos] : 0;
        const float4 q0 = s_rec[j * 3 + 0], q1 = s_rec[j * 3 + 1], q2 = s_rec[j * 3 + 2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float rcx = (float)(patch_x + (q & 1) * 4) + 2.0f, rcy = (float)(patch_y + (q >> 1) * 4) + 2.0f;
          const bool hit = in && scan_rect_hit(q0, q1, q2, rcx, rcy, 1.5f);
          const unsigned long long m = __ballot(hit);
          const int pos = cnt[q] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
          if (hit) s_list[wave][q][pos] = (uint8_t)ppos;
          cnt[q] += __builtin_popcountll(m);
        }
      }
      wave_lds_fence();
#if MS_SCAN_ABLATE == 1
      continue;
#endif
#if MS_SCAN_STATS
      if (lane == 0) {
        atomicAdd(&g_scan_stats[0], 1ull);                                                      // passes
        atomicAdd(&g_scan_stats[1], (unsigned long long)(cnt[0] + cnt[1] + cnt[2] + cnt[3]));   // (sub-patch, splat) hits
        atomicAdd(&g_scan_stats[6], (unsigned long long)pcount);                                // (patch, splat) hits
      }
#endif

      // ---- blend: lane = splat, 16 pixel steps per chunk ------------------------------------------------------
#pragma unroll 1
      for (int q = 0; q < 4; ++q) {
        const int n = q == 0 ? cnt[0] : q == 1 ? cnt[1] : q == 2 ? cnt[2] : cnt[3];
        if (n == 0) continue;
        wave_lds_fence();
        const unsigned long long alive = __ballot(__float_as_uint(s_pix[wave][lane].w) > oms_bits);
        if (((alive >> (16 * q)) & 0xffffull) == 0) continue;
        const int pbase = q * 16;
        const float fx = (float)(patch_x + (q & 1) * 4) + 0.5f, fy = (float)(patch_y + (q >> 1) * 4) + 0.5f;

#pragma unroll 1
        for (int c0 = 0; c0 < n; c0 += 64) {
          wave_lds_fence();                      // pixel state and accumulator rows written by other lanes in earlier chunks
          const bool valid = c0 + lane < n;
          const int pos = valid ? (int)s_list[wave][q][c0 + lane] : 0;
          const int idx = (int)s_plist[wave][pos];
          const float4 q0 = s_rec[idx * 3 + 0], q1 = s_rec[idx * 3 + 1], q2 = s_rec[idx * 3 + 2];
          const float A = q0.z, B = q0.w, C = q1.x, D = q1.y;
          const float nl2a = valid ? q1.z : __builtin_inff();       // idle lanes: alpha g = exp2(-inf) = 0
          const float f0 = q1.w, f1 = q2.x, f2 = q2.y;
          const float dx0 = fx - q0.x, dy0 = fy - q0.y;             // first pixel centre of the sub-patch - mean
          const float X00 = A * dx0 + B * dy0, Y00 = C * dx0 + D * dy0;
          // (X', Y') at the first pixel of each of the four pixel rows; a step adds x * (A, C)
          const float Xr[4] = {X00, X00 + B, __builtin_fmaf(B, 2.0f, X00), __builtin_fmaf(B, 3.0f, X00)};
          const float Yr[4] = {Y00, Y00 + D, __builtin_fmaf(D, 2.0f, Y00), __builtin_fmaf(D, 3.0f, Y00)};

          float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
          float h0 = 0.f, h1 = 0.f;
          // GRID_MOMENTS: the six moments are first taken in the sub-patch's own pixel grid,
          //   n = sum q {1, x, y, x^2, x y, y^2},  x, y in 0..3,
          // where x and y are compile-time constants of the unrolled steps: a pixel row keeps r = sum_x q {1, x, x^2}
          // (7 additions / FMAs for its four pixels), a row end folds r into n with the constants y, y^2 (3 to 6), and
          // the chunk ends with the affine change of variables X = X00 + A x + B y, Y = Y00 + C x + D y (26) — 75
          // instructions per chunk instead of 16 x 8.
          float n0 = 0.f, n1 = 0.f, n2 = 0.f, n3 = 0.f, n4 = 0.f, n5 = 0.f, r0 = 0.f, r1 = 0.f, r2 = 0.f;
          // heuristics: |d mu| = |q (X A + Y C)| + |q (X B + Y D)| (backward.py:190-194) is affine in the grid too:
          //   X A + Y C = gx0 + x gxx + y gxy,   X B + Y D = gy0 + x gxy + y gyy
          const float gxx = __builtin_fmaf(A, A, C * C), gxy = __builtin_fmaf(A, B, C * D), gyy = __builtin_fmaf(B, B, D * D);
          const float gx0 = __builtin_fmaf(X00, A, Y00 * C), gy0 = __builtin_fmaf(X00, B, Y00 * D);
#if MS_SCAN_STATS
          int steps_run = 0, lanes_contrib = 0;
#endif

          // Two instantiations of the 16 steps, chosen per chunk (wave-uniform): the grid form expands X, Y around the
          // sub-patch's first pixel, where |X00| ~ 3.3 + 4 |A|: with a steep basis (sigma well below a pixel: only the
          // 2D operators can hand those in, a projected gaussian has sigma >= sqrt(blur_cov) = 0.55 px) the products
          // X00^2 n0, A^2 n3 ... cancel to the sum they stand for and lose digits the per-pixel form keeps.
          auto blend_chunk = [&](auto grid_tag) {
            constexpr bool GRID = decltype(grid_tag)::value;
            // The 16 pixels are visited in PAIRS (x, x + 1 of one pixel row): the two pixels are independent, so each
            // pair runs two dependency chains side by side.  The data of the next pair is requested before the
            // current pair is evaluated.
            constexpr int U = 2;                     // pixels per step (see wave_scan_mul2)
            float4 pg[U];
            float prg[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { pg[u] = s_pix[wave][pbase + u]; prg[u] = s_rg[wave][pbase + u]; }
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
                if (i + U < 16) { pg[u] = s_pix[wave][p + U + u]; prg[u] = s_rg[wave][p + U + u]; }
              }
              // wave-uniform: a group of saturated / out-of-image pixels is skipped; if only some of them are dead, their
              // lanes all find T <= 1 - saturate_threshold below and contribute nothing
              if (__ballot(any_alive) != 0) {

              float X[U], Y[U], a_gated[U], a[U], om[U], Tk[U];
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const int x = (i + u) & 3, y = i >> 2;
                X[u] = x == 0 ? Xr[y] : __builtin_fmaf(A, (float)x, Xr[y]);
                Y[u] = x == 0 ? Yr[y] : __builtin_fmaf(C, (float)x, Yr[y]);
                const float a_raw = __builtin_amdgcn_exp2f(-__builtin_fmaf(X[u], X[u], __builtin_fmaf(Y[u], Y[u], nl2a)));
                // blend gate (forward.py:99-101): lanes below the threshold carry alpha = 0 from here on
                a_gated[u] = a_raw > rp.alpha_threshold ? a_raw : 0.0f;
                a[u] = min_f32_uniform(a_gated[u], rp.clamp_max_alpha);
                om[u] = 1.0f - a[u];
                // T before this splat: exclusive prefix product seeded with the pixel's T (lane 0 <- T of the pixel)
                Tk[u] = dpp_f32<0x138>(cur[u].w, om[u]);                              // wave_shr:1
              }
              wave_scan_mul2(Tk[0], Tk[1]);
              // saturation skip (backward.py:154): splats that find T <= 1 - saturate_threshold do not blend.
              // A pixel crosses that line inside at most one chunk of its life: wave-uniform slow path.
              float a_st[U];                                                          // straight-through alpha (below)
              bool any_sat = false;
#pragma unroll
              for (int u = 0; u < U; ++u) { a_st[u] = a_gated[u]; any_sat |= !(Tk[u] > oms); }
              if (__ballot(any_sat) != 0) {
                asm volatile("; saturation inside the chunk" ::: "memory");          // keep this a branch, not selects
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
              wave_scan_add2(S[0], S[1]);
              // <R, G> after this splat: R -= f w  (backward.py:171-174)
              float RGout[U];
#pragma unroll
              for (int u = 0; u < U; ++u) RGout[u] = RGin[u] - S[u];
              // the last lane holds the state of both pixels after the whole chunk: one masked block, three LDS writes
              if (last_lane) {
                s_pix[wave][p].w = Tk[0] * om[0];
                s_pix[wave][p + 1].w = Tk[1] * om[1];
                *reinterpret_cast<float2*>(&s_rg[wave][p]) = make_float2(RGout[0], RGout[1]);
              }
#pragma unroll
              for (int u = 0; u < U; ++u) {
                const float RGk = RGout[u];
                // d(alpha) = T <f, G> - <R, G> / (1 - alpha)
                const float ag = __builtin_fmaf(Tk[u], fG[u], -(RGk * __builtin_amdgcn_rcpf(om[u])));
                // straight-through clamp (backward.py:158-163): d(alpha_pt g) = d(alpha); q = alpha_pt g d(alpha)
                const float q_ = ag * a_st[u];
                const float qX = q_ * X[u], qY = q_ * Y[u];
                if (GRID) {
                  const int x = (i + u) & 3;                          // compile-time in the unrolled loop
                  r0 += q_;
                  if (x == 1) { r1 += q_; r2 += q_; }
                  if (x > 1) { r1 = __builtin_fmaf(q_, (float)x, r1); r2 = __builtin_fmaf(q_, (float)(x * x), r2); }
                } else {
                  m0 += q_; m1 += qX; m2 += qY;
                  m3 = __builtin_fmaf(qX, X[u], m3); m4 = __builtin_fmaf(qX, Y[u], m4); m5 = __builtin_fmaf(qY, Y[u], m5);
                }
                a0 = __builtin_fmaf(w[u], cur[u].x, a0); a1 = __builtin_fmaf(w[u], cur[u].y, a1); a2 = __builtin_fmaf(w[u], cur[u].z, a2);
                if (HEUR) {                                           // backward.py:190-194
                  const float agm = a_st[u] != 0.0f ? ag : 0.0f;
                  h0 = __builtin_fmaf(agm, agm, h0);
                  if (GRID) {
                    const int x = (i + u) & 3, y = i >> 2;
                    const float bx = y == 0 ? gx0 : __builtin_fmaf(gxy, (float)y, gx0), by = y == 0 ? gy0 : __builtin_fmaf(gyy, (float)y, gy0);
                    const float tx = x == 0 ? bx : __builtin_fmaf(gxx, (float)x, bx), ty = x == 0 ? by : __builtin_fmaf(gxy, (float)x, by);
                    h1 += fabsf(q_ * tx) + fabsf(q_ * ty);
                  } else {
                    h1 += fabsf(__builtin_fmaf(qX, A, qY * C)) + fabsf(__builtin_fmaf(qX, B, qY * D));
                  }
                }
#if MS_SCAN_STATS
                lanes_contrib += __builtin_popcountll(__ballot(w[u] != 0.0f));
#endif
              }
#if MS_SCAN_STATS
              steps_run += U;
#endif
              }
              if (GRID && (i & 3) == 2) {                     // end of pixel row y: fold it into the grid moments
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
            if (GRID) {
              // X = X00 + A x + B y,  Y = Y00 + C x + D y:  sums of q {1, X, Y, X^2, X Y, Y^2} from the grid moments
              const float P = __builtin_fmaf(A, n1, B * n2), Q = __builtin_fmaf(C, n1, D * n2);
              const float s1 = __builtin_fmaf(A, n3, B * n4), s2 = __builtin_fmaf(A, n4, B * n5);
              const float t1 = __builtin_fmaf(C, n3, D * n4), t2 = __builtin_fmaf(C, n4, D * n5);
              m0 = n0;
              m1 = __builtin_fmaf(X00, n0, P);
              m2 = __builtin_fmaf(Y00, n0, Q);
              m3 = __builtin_fmaf(X00, m1 + P, __builtin_fmaf(A, s1, B * s2));
              m5 = __builtin_fmaf(Y00, m2 + Q, __builtin_fmaf(C, t1, D * t2));
              m4 = __builtin_fmaf(X00, m2, __builtin_fmaf(Y00, P, __builtin_fmaf(A, t1, B * t2)));
            }
          };
          {
            const bool steep = fmaxf(fmaxf(fabsf(A), fabsf(B)), fmaxf(fabsf(C), fabsf(D))) > GRID_MAX_BASIS;
#if MS_GRID_MOMENTS == 2               // A/B builds: the grid form whatever the basis
            (void)steep;
            blend_chunk(std::true_type{});
#else
            if (GRID_MOMENTS && __ballot(valid && steep) == 0) blend_chunk(std::true_type{});
            else blend_chunk(std::false_type{});
#endif
          }
#if MS_SCAN_STATS
          ++batch_chunks;
          if (lane == 0) {
            atomicAdd(&g_scan_stats[2], 1ull);                                           // chunks
            atomicAdd(&g_scan_stats[3], (unsigned long long)min(64, n - c0));             // filled lanes
            if (n - c0 <= 16) atomicAdd(&g_scan_stats[10], 1ull);                         // chunks of <= 16 splats
            else if (n - c0 <= 32) atomicAdd(&g_scan_stats[11], 1ull);                    // chunks of 17..32 splats
            atomicAdd(&g_scan_stats[4], (unsigned long long)steps_run);                  // executed pixel steps
            atomicAdd(&g_scan_stats[5], (unsigned long long)lanes_contrib);              // contributing (pixel, splat) pairs
          }
#endif

          // this wave's row of the splat: the lanes of a chunk hold distinct splats, chunks run one after the other
          if (valid) {
            float* row = &s_acc[wave][pos][0];
            const float v[11] = {m0, m1, m2, m3, m4, m5, a0, a1, a2, h0, h1};
#pragma unroll
            for (int k = 0; k < NACC; ++k) row[k] += v[k];
          }
        }
      }

      // ---- commit the pass: ONE 64-byte, line-aligned row of global float atomics per (patch, splat) ------------
      wave_lds_fence();
      // ROWS_PER rows of NACC sums per instruction (7 x 9 = 63 lanes; 5 x 11 with heuristics): the LDS reads sweep
      // the wave's accumulator block linearly and the loop runs pcount / 7 times (16 lanes per row, 9 of them with
      // data, ran pcount / 4 times: 1.45 -> 1.37 ms on config D; the atomics themselves are 0.10 ms of instruction
      // rate + 0.04 ms of misses: 1.33 ms when every row lands in a 16 MB window, 1.21 ms without the commit)
      {
        constexpr int ROWS_PER = 64 / NACC;
        const int sub_row = lane / NACC, k = lane - sub_row * NACC;
        const bool lane_used = sub_row < ROWS_PER;
        for (int e0 = 0; e0 < pcount; e0 += ROWS_PER) {
          const int e = e0 + sub_row;
          if (lane_used && e < pcount) {
            const float v = s_acc[wave][e][k];
            if (v != 0.0f) {
              const size_t word = (size_t)(uint32_t)s_id[s_plist[wave][e]] * MOMENT_ROW + k;
              if (rp.deterministic)
                __hip_atomic_fetch_add(reinterpret_cast<long long*>(moments) + word,
                                       (long long)llrintf(v * (k == 9 ? fixed_h0 : fixed_main)),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              else
                atomic_add_noret(moments + word, v);
              s_acc[wave][e][k] = 0.0f;
            }
          }
        }
      }
      wave_lds_fence();
    }
#if MS_SCAN_STATS
    if (lane == 0) { atomicAdd(&s_bsum, batch_chunks); atomicMax(&s_bmax, batch_chunks); }
#endif
  }
#if MS_SCAN_STATS
  __syncthreads();
  MS_FLUSH_BALANCE()
#undef MS_FLUSH_BALANCE
#endif
}

// Moments -> gradients of the packed 2D gaussian and its colour (one thread per point; plain stores).
// With (X, Y) = M (pixel - mean), M = [[A, B], [C, D]] = diag(1/sx, 1/sy) [axis; perp(axis)] and the sums
// S = sum q, Sx = sum q X, ..., Syy = sum q Y^2 over all contributing pixels (q = alpha g dL/dalpha):
//   d mean  = M^T (Sx, Sy)                                  d sigma = (Sxx / sx, Syy / sy)
//   d axis  = sum q (X/sx (-d) + Y/sy perp(d)),  d = M^-1 (X, Y)     d alpha = S / alpha      (generic.py:321-336)
// The kernel stores the moments of (X', Y') = s (X, Y), s = sqrt(log2(e) / 2).
// REZERO (frame executor): the rows are cleared as they are read, so a persistent moments buffer needs no fill pass.
template <bool HEUR, bool FIXED, bool REZERO = false>
__global__ void __launch_bounds__(256)
raster_moments_finalize_kernel(const float* __restrict__ points, float* __restrict__ moments, int64_t n,
                               float* __restrict__ grad_points, float* __restrict__ grad_feats,
                               float* __restrict__ heuristic, const int32_t* __restrict__ fixed_exp,
                               int gp_stride = 7, int gf_stride = 3, int covariance_form = 0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float4 r0, r1, r2;
  if (FIXED) {       // deterministic mode: rows of 64-bit fixed-point integers
    long long* row = reinterpret_cast<long long*>(moments) + i * MOMENT_ROW;
    const int e_main = fixed_exp[0], e_h0 = fixed_exp[1];
    float v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      v[k] = (float)ldexp((double)row[k], k == 9 ? -e_h0 : -e_main);
      if (REZERO) row[k] = 0;
    }
    r0 = make_float4(v[0], v[1], v[2], v[3]); r1 = make_float4(v[4], v[5], v[6], v[7]); r2 = make_float4(v[8], v[9], v[10], v[11]);
  } else {
    float4* row = reinterpret_cast<float4*>(moments + i * MOMENT_ROW);
    r0 = row[0]; r1 = row[1]; r2 = row[2];
    if (REZERO) { const float4 z = make_float4(0.f, 0.f, 0.f, 0.f); row[0] = z; row[1] = z; row[2] = z; }
  }
  if (grad_points) {
    const float* g = points + i * 7;
    const float ax = g[2], ay = g[3], sx = g[4], sy = g[5], alpha = g[6];
    // a splat the rasterizer can never blend (alpha or a sigma of 0 or NaN: masked / underflowed parameters of a direct
    // rasterize() caller, padding rows, culled rows of the frame executor) has an all-zero row; its gradient is
    // exactly zero like the atomic path's (0 / 0 or 0 * NaN otherwise)
    const bool dead = !(alpha > 0.0f) || !(sx > 0.0f) || !(sy > 0.0f);
    float* o = grad_points + i * gp_stride;
    if (dead) {
#pragma unroll
      for (int k = 0; k < 7; ++k) o[k] = 0.0f;
    } else {
      const float isx = 1.0f / sx, isy = 1.0f / sy;
      const float A = ax * isx, B = ay * isx, C = -ay * isy, D = ax * isy;
      constexpr float IS = 1.0f / EXP2_BASIS_SCALE, IS2 = IS * IS;
      const float S = r0.x, Sx = r0.y * IS, Sy = r0.z * IS, Sxx = r0.w * IS2, Sxy = r1.x * IS2, Syy = r1.y * IS2;
      const float det = A * D - B * C;
      const float idet = det != 0.0f ? 1.0f / det : 0.0f;
      o[0] = Sx * A + Sy * C;
      o[1] = Sx * B + Sy * D;
      if (covariance_form) {
        // dL/d(a, b, c) of the covariance [[a, b], [b, c]] = U diag(sigma^2) U^T (gaussian_bwd.hip has the derivation):
        // what the per-gaussian pass of a multi-GPU rank step reads (MS_BOUNDARY_COVARIANCE)
        const float N00 = Sxx * isx * isx, N01 = Sxy * isx * isy, N11 = Syy * isy * isy;
        const float uu = ax * ax, ww = ay * ay, uw = ax * ay;
        o[2] = 0.5f * (N00 * uu - 2.0f * N01 * uw + N11 * ww);
        o[3] = (N00 - N11) * uw + N01 * (uu - ww);
        o[4] = 0.5f * (N00 * ww + 2.0f * N01 * uw + N11 * uu);
        o[5] = 0.0f;
      } else {
        o[2] = -(isx * (D * Sxx - B * Sxy) + isy * (A * Syy - C * Sxy)) * idet;
        o[3] = (isy * (D * Sxy - B * Syy) + isx * (C * Sxx - A * Sxy)) * idet;
        o[4] = isx * Sxx;
        o[5] = isy * Syy;
      }
      o[6] = S / alpha;
    }
  }
  if (HEUR && heuristic) {                       // backward.py:190-194: sum (alpha d_alpha)^2, sum |d mean|_1
    const float alpha = points[i * 7 + 6];
    constexpr float IS2 = 1.0f / (EXP2_BASIS_SCALE * EXP2_BASIS_SCALE);
    heuristic[i * 2 + 0] = alpha * alpha * r2.y;
    heuristic[i * 2 + 1] = r2.z * IS2;
  }
  if (grad_feats) {
    float* c = grad_feats + i * gf_stride;
    c[0] = r1.z; c[1] = r1.w; c[2] = r2.x;
  }
}

}  // namespace ms

using namespace ms;

static bool tile32_quarters() {
  static const bool q = [] { const char* e = getenv("MS_TILE32_BWD"); return e && e[0] == 'q'; }();
  return q;
}

static int launch_scan_backward(const float* points7, const float* features,
                                const int32_t* tile_ranges, const int32_t* overlap_to_point, const float* image,
                                const float* grad_image, int image_w, int image_h, const ms_raster_config* cfg,
                                float* moments, int deterministic, const int32_t* fixed_exp, int tile_row_begin,
                                int tile_row_end, hipStream_t s, const char* who, int grad_broadcast = 0) {
  if (deterministic && !fixed_exp) { set_error("%s: deterministic commits need fixed_exp (ms_fixed_point_exponents)", who); return MS_ERR_BAD_ARG; }
  if (image_w <= 0 || image_h <= 0) { set_error("%s: bad image size", who); return MS_ERR_BAD_ARG; }
  if (!cfg->use_alpha_blending) {
    set_error("%s: backward requires use_alpha_blending (reference: tests/test_rasterizer.py:92-94)", who);
    return MS_ERR_BAD_ARG;
  }
  if (cfg->antialias) { set_error("%s: antialiased pdf is served by ms_raster_bwd", who); return MS_ERR_UNSUPPORTED; }
  const int ts = cfg->tile_size;
  if (ts != 8 && ts != 16 && ts != 32) { set_error("%s: tile_size must be 8, 16 or 32 (got %d)", who, ts); return MS_ERR_UNSUPPORTED; }
  const int tiles_high = (image_h + ts - 1) / ts, tiles_wide = (image_w + ts - 1) / ts;
  if (tile_row_begin < 0) tile_row_begin = 0;
  if (tile_row_end > tiles_high) tile_row_end = tiles_high;
  if (tile_row_end <= tile_row_begin) return 0;

  FastParams rp;
  rp.width = image_w; rp.height = image_h; rp.tiles_wide = tiles_wide; rp.tile_begin = tile_row_begin * tiles_wide;
  rp.clamp_max_alpha = (float)cfg->clamp_max_alpha;
  rp.alpha_threshold = (float)cfg->alpha_threshold;
  rp.one_minus_saturate = (float)(1.0 - cfg->saturate_threshold);
  rp.deterministic = deterministic != 0;
  rp.grad_broadcast = grad_broadcast != 0;
  rp.num_tiles = (tile_row_end - tile_row_begin) * tiles_wide;
#define MS_GO(TS, HEUR, SPLIT) raster_bwd_scan_kernel<TS, HEUR, SPLIT>                                           \
      <<<dim3(xcd_grid<(SPLIT > 1 ? 1 : 0)>(rp.num_tiles, SPLIT * SPLIT)), dim3(TS * TS), 0, s>>>(              \
          points7, features, tile_ranges, overlap_to_point, image, grad_image, rp, moments, fixed_exp)
#define MS_GO_TILE(TS, SPLIT)                                                                                   \
  do { if (hf) MS_GO(TS, true, SPLIT); else MS_GO(TS, false, SPLIT); } while (0)
  const bool hf = cfg->compute_point_heuristic;
#ifdef MS_WITH_ROWS_KERNEL      // tools/experiments/raster_bwd_rows.hip (measured and dropped in round 4)
  if (launch_rows_backward(points7, features, tile_ranges, overlap_to_point, image, grad_image, rp, ts, hf, moments,
                           fixed_exp, s)) {
    MS_CHECK_LAUNCH();
    return 0;
  }
#endif
  switch (ts) {
    case 8: MS_GO_TILE(8, 1); break;
    case 16: MS_GO_TILE(16, 1); break;
    default:
      // tile 32: ONE 1024-thread workgroup per tile with 896-splat batches (152 KB LDS), or — MS_TILE32_BWD=quarters —
      // four 16 x 16 quarter workgroups per tile that each stage the whole tile list
      if (tile32_quarters()) MS_GO_TILE(16, 2); else MS_GO_TILE(32, 1);
      break;
  }
#undef MS_GO_TILE
#undef MS_GO
  MS_CHECK_LAUNCH();
  return 0;
}

extern "C" int ms_raster_bwd_moments(const void* points7, const void* features, const int32_t* tile_ranges,
                                     const int32_t* overlap_to_point, const void* image, const void* grad_image,
                                     int image_w, int image_h, const ms_raster_config* cfg, float* moments,
                                     int deterministic, const int32_t* fixed_exp, int tile_row_begin,
                                     int tile_row_end, void* stream) {
  MS_CHECK_ARG(cfg && points7 && features && tile_ranges && image && grad_image && moments, "null pointer");
  return launch_scan_backward((const float*)points7, (const float*)features, tile_ranges, overlap_to_point,
                              (const float*)image, (const float*)grad_image, image_w, image_h, cfg, moments,
                              deterministic, fixed_exp, tile_row_begin, tile_row_end, (hipStream_t)stream,
                              "ms_raster_bwd_moments");
}

extern "C" int ms_fixed_point_exponents(const float* amax_dev, int32_t* out_exp2, void* stream) {
  MS_CHECK_ARG(amax_dev && out_exp2, "null pointer");
  fixed_point_exponents_kernel<<<1, 1, 0, (hipStream_t)stream>>>(amax_dev, out_exp2);
  MS_CHECK_LAUNCH();
  return 0;
}

#if MS_SCAN_STATS
extern "C" int ms_debug_scan_stats(unsigned long long* out10, int reset) {
  if (out10) (void)hipMemcpyFromSymbol(out10, HIP_SYMBOL(g_scan_stats), 12 * sizeof(unsigned long long));
  if (reset) { unsigned long long z[12] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_scan_stats), z, sizeof(z)); }
  return 0;
}
#endif

extern "C" int ms_raster_moments_finalize(const void* points7, const float* moments, int deterministic,
                                          const int32_t* fixed_exp, int64_t n,
                                          float* grad_points7, float* grad_features, float* point_heuristic,
                                          void* stream) {
  MS_CHECK_ARG(n >= 0, "negative size");
  MS_CHECK_ARG(!deterministic || fixed_exp, "deterministic rows need fixed_exp");
  if (n == 0) return 0;
  MS_CHECK_ARG(points7 && moments, "null pointer");
  if (!grad_points7 && !grad_features && !point_heuristic) return 0;
  const dim3 grid((unsigned)div_up(n, 256));
  hipStream_t s = (hipStream_t)stream;
#define MS_GO(HEUR, FIXED) raster_moments_finalize_kernel<HEUR, FIXED><<<grid, 256, 0, s>>>(            \
      (const float*)points7, const_cast<float*>(moments), n, grad_points7, grad_features, point_heuristic, fixed_exp)
  if (point_heuristic) { if (deterministic) MS_GO(true, true); else MS_GO(true, false); }
  else { if (deterministic) MS_GO(false, true); else MS_GO(false, false); }
#undef MS_GO
  MS_CHECK_LAUNCH();
  return 0;
}

// frame executor (frame_internal.h): finalize + re-zero of the rows it read
namespace ms {
int raster_bwd_moments_launch(const void* points7, const void* features, const int32_t* tile_ranges,
                              const int32_t* overlap_to_point, const void* image, const void* grad_image, int image_w,
                              int image_h, const ms_raster_config* cfg, float* moments, int deterministic,
                              const int32_t* fixed_exp, int tile_row_begin, int tile_row_end, int grad_broadcast,
                              hipStream_t s) {
  return launch_scan_backward((const float*)points7, (const float*)features, tile_ranges, overlap_to_point,
                              (const float*)image, (const float*)grad_image, image_w, image_h, cfg, moments,
                              deterministic, fixed_exp, tile_row_begin, tile_row_end, s, "ms_frame_backward", grad_broadcast);
}

int moments_finalize_rezero_launch(const float* points7, float* moments, int deterministic, const int32_t* fixed_exp,
                                   int64_t n, float* grad_points7, float* grad_features, float* point_heuristic,
                                   hipStream_t s, int row_stride, int covariance_form) {
  if (n == 0) return 0;
  const dim3 grid((unsigned)div_up(n, 256));
  const int gp_stride = row_stride > 0 ? row_stride : 7, gf_stride = row_stride > 0 ? row_stride : 3;
#define MS_GO(HEUR, FIXED) raster_moments_finalize_kernel<HEUR, FIXED, true><<<grid, 256, 0, s>>>(      \
      points7, moments, n, grad_points7, grad_features, point_heuristic, fixed_exp, gp_stride, gf_stride, covariance_form)
  if (point_heuristic) { if (deterministic) MS_GO(true, true); else MS_GO(true, false); }
  else { if (deterministic) MS_GO(false, true); else MS_GO(false, false); }
#undef MS_GO
  MS_CHECK_LAUNCH();
  return 0;
}
}  // namespace ms
