// raster_bwd_rows.hip — EXPERIMENT of round 4, measured and NOT part of the library: the raster
// backward (float32, RGB, plain gaussian pdf, alpha blending, tile 8 and 16; semantics of rasterizer/backward.py:97-224)
// in the splat-per-lane organisation of raster_bwd_scan.hip with the hit lists cut FINER and packed TIGHTER.
// Gradients equal the product kernel's; on config D it executes 39 % fewer pixel steps (8.1 M against 13.3 M, 30.6
// instead of 18.7 contributing lanes per step) and still runs 1.75 ms against 1.35 ms: building sixteen lists per
// (wave, pass) costs ~1200 VALU + ~800 SALU instructions per pass (the sixteen counts and row offsets live in SGPRs,
// which spill through v_writelane) against ~2400 for the chunks of the pass, and a chunk of two pixel-pair steps
// carries 70 + 24 per list segment instructions of set-up for its 184 of blending.
// To build it:  tools/build_variant.sh rows -DMS_WITH_ROWS_KERNEL=1  (links it in place of the hook's stub).
//
// raster_bwd_scan.hip keeps one hit list per 4x4 sub-patch and walks it in 64-lane chunks of 16 pixel steps.  Measured
// on config D: a chunk is 46 / 64 full (every list ends in a partial chunk) and a splat that hits a 4x4 sub-patch
// reaches 6.4 of its 16 pixels — 18.7 of 64 lanes do useful work per pixel step.  Here
//
//   * a list belongs to a 2x2 pixel QUAD (16 per 8x8 patch).  A splat enters it only if one of the four pixel centres
//     can pass the blend gate: per pixel row of the patch the x-interval with  X^2 + Y^2 < 2 ln(alpha / threshold)  is
//     solved in closed form (one v_sqrt per row; the discriminant is  qa R^2 - (det dy)^2 — no cancellation — and is
//     inflated, so the test never rejects a pixel the blend would accept), which gives a 64-pixel mask per splat and,
//     from it, a 16-bit quad mask.  tools/model_bwd_chunks.py (CPU model, calibrated on the old kernel's counters):
//     6.97 quad hits per tile overlap with 73 % of their pixels contributing, against 3.06 sub-patch hits with 42 %;
//   * the lists of a wave are packed by 16-lane DPP ROWS into one pool: a list takes ceil(n / 16) consecutive rows, a
//     chunk is four consecutive rows whatever lists they belong to, and runs 2 pixel-pair steps (the quad's 4 pixels).
//     A lane reads ITS quad's pixel state (one LDS address per row: broadcast).  The two DPP prefix scans stay inside a
//     row for their row_shr levels; the row_bcast levels run under EXEC masks so that a row receives from the row
//     above only when it continues that row's list (raster_bwd_shared.h).  Chunk fill 48 / 64 instead of 45 of 64 on
//     lists a third as long;
//   * everything else is the old kernel: LDS staging of the tile's list in equal batches with the two-deep gather
//     pipeline, the level-1 cull to a per-wave patch list, per-lane moment sums, per-wave LDS accumulator rows
//     (plain read-add-write; rows of one chunk that belong to DIFFERENT quads may hold the same splat, so the
//     read-add-write runs once per list segment of the chunk), one 64-byte row of global atomics per (patch, splat).
#include <stdlib.h>

#include "raster_bwd_shared.h"
#include "frame_internal.h"

namespace ms {
#define MS_SCAN2_STEP(OP, CTRL)                                                                  \
  OP " %0, %0, %0 " CTRL "\n\t" OP " %1, %1, %1 " CTRL "\n\ts_nop 0\n\t"
// The same scans when the wave holds SEVERAL lists side by side, each made of whole 16-lane rows (raster_bwd_rows.hip):
// the four row_shr levels never leave a row; the two row_bcast levels, which carry a row's total into the rows
// after it, run under EXEC masks so that only rows CONTINUING the list of the row before them receive:
//   exec15 = rows 0 and 2 (the sources) + row 1 / row 3 where it continues row 0 / row 2
//   exec31 = rows 0 and 1 (lane 31 is the source) + row 2 where it continues row 1 + row 3 where rows 2 AND 3 continue
// A lane switched off in EXEC is neither written nor — DPP, bound_ctrl = 0 — a valid source, hence the sources stay on.
// s_mov_b64 exec needs no wait state before a DPP instruction (only a VALU write of EXEC does); the pair of
// row_bcast:15 writes is followed by the s_mov + s_nop 0 as the two wait states the dependent row_bcast:31 reads need.
// Must run in wave-uniform control flow (EXEC = all lanes on entry and on exit).
#define MS_SCAN2M_ASM(OP)                                                                         \
  asm volatile("s_nop 1\n\t"                                                                      \
      MS_SCAN2_STEP(OP, "row_shr:1 row_mask:0xf bank_mask:0xf")                                   \
      MS_SCAN2_STEP(OP, "row_shr:2 row_mask:0xf bank_mask:0xf")                                   \
      MS_SCAN2_STEP(OP, "row_shr:4 row_mask:0xf bank_mask:0xf")                                   \
      MS_SCAN2_STEP(OP, "row_shr:8 row_mask:0xf bank_mask:0xf")                                   \
      "s_mov_b64 exec, %2\n\t"                                                                    \
      OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                               \
      OP " %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                               \
      "s_mov_b64 exec, %3\n\t"                                                                    \
      "s_nop 0\n\t"                                                                               \
      OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                               \
      OP " %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"                               \
      "s_mov_b64 exec, -1"                                                                        \
      : "+v"(a), "+v"(b) : "s"(exec15), "s"(exec31))
__device__ __forceinline__ void wave_scan_mul2_rows(float& a, float& b, unsigned long long exec15, unsigned long long exec31) {
  MS_SCAN2M_ASM("v_mul_f32_dpp");
}
__device__ __forceinline__ void wave_scan_add2_rows(float& a, float& b, unsigned long long exec15, unsigned long long exec31) {
  MS_SCAN2M_ASM("v_add_f32_dpp");
}
#undef MS_SCAN2M_ASM
#undef MS_SCAN2_STEP
}  // namespace ms

#ifndef MS_SCAN_STATS
#define MS_SCAN_STATS 0
#endif
#ifndef MS_R16_BATCH
#define MS_R16_BATCH 216
#endif
#ifndef MS_R16_CAP
#define MS_R16_CAP 128
#endif
#ifndef MS_R_POOL_ROWS
#define MS_R_POOL_ROWS 64
#endif

namespace ms {

#if MS_SCAN_STATS
__device__ unsigned long long g_rows_stats[12];
#endif

// Quads of the 8x8 patch whose 2x2 pixel centres the splat can reach: bit 8 qy + 2 qx for quad (qx, qy).
// (ox, oy) = centre of the patch's first pixel; E0 = -log2(alpha_threshold).  The blend evaluates
//   alpha g = exp2(-(X'^2 + Y'^2 + nl2a)),  X' = A' dx + B' dy,  Y' = C' dx + D' dy       (write_scan_record)
// and gates on alpha g > threshold, i.e. X'^2 + Y'^2 < E0 - nl2a =: R2.  Along pixel row r (dy = dy0 + r), with x the
// pixel offset from the patch's first column, that is the quadratic  qa x^2 + 2 hb x + c < R2  with
//   qa = A'^2 + C'^2,   vertex  xc = -(A' X'r + C' Y'r) / qa  (linear in r),
//   discriminant / 4 = qa R2 - (A' Y'r - C' X'r)^2 = qa R2 - (det dy)^2        (det = A' D' - B' C')
// — the second form has no cancellation between large terms.  R2 is inflated by 5e-4 (+ 1e-4), the discriminant by
// 4e-6 of each term (more than the float32 rounding of either), the interval by 0.01 pixel: a pixel the blend accepts is
// never dropped; a pixel within those margins of the boundary may be kept although it fails the gate, which costs one
// idle lane and changes nothing (the gate itself decides in the blend).
__device__ __forceinline__ uint32_t quad_mask(const float4 q0, const float4 q1, float ox, float oy, float E0) {
  const float A = q0.z, B = q0.w, C = q1.x, D = q1.y;
  const float R2 = __builtin_fmaf(E0 - q1.z, 1.0005f, 1.0e-4f);
  const float qa = __builtin_fmaf(A, A, C * C);
  const float inv_qa = __builtin_amdgcn_rcpf(qa);
  const float det = __builtin_fmaf(A, D, -(B * C));
  const float dx0 = ox - q0.x, dy0 = oy - q0.y;
  const float Xr0 = __builtin_fmaf(A, dx0, B * dy0), Yr0 = __builtin_fmaf(C, dx0, D * dy0);
  const float xc0 = -__builtin_fmaf(A, Xr0, C * Yr0) * inv_qa;
  const float slope = -__builtin_fmaf(A, B, C * D) * inv_qa;
  const float cr0 = det * dy0;
  const float qaR = qa * R2 * 1.000004f;
  uint32_t m = 0;
  float cr = cr0, xc = xc0;
  // one quad row (two pixel rows) per iteration; NOT unrolled further: eight rows in flight cost ~45 registers at the
  // point where the staging pipeline's registers are live as well
#pragma unroll 1
  for (int qr = 0; qr < 4; ++qr) {
    uint32_t both = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float disc = __builtin_fmaf(cr * -0.999996f, cr, qaR);
      const float h = __builtin_amdgcn_sqrtf(__builtin_fmaxf(disc, 0.0f)) * inv_qa;
      const float lo = __builtin_amdgcn_fmed3f(__builtin_ceilf(xc - h - 0.01f), 0.0f, 8.0f);
      const float hi = __builtin_amdgcn_fmed3f(__builtin_floorf(xc + h + 1.01f), 0.0f, 8.0f);      // one past the last pixel
      const uint32_t width = (uint32_t)__builtin_fmaxf(hi - lo, 0.0f);
      const uint32_t pixels = ((1u << width) - 1u) << (uint32_t)lo;                          // 8 bits
      both |= pixels;
      cr += det;
      xc += slope;
    }
    m |= ((both | (both >> 1)) & 0x55u) << (8 * qr);
  }
  return m;
}

template <int TS, bool HEUR>
// tile 16: four workgroups of four waves per CU = four waves per SIMD, which needs <= 128 VGPRs; left alone the
// register allocator takes 151 (it does not see that LDS allows the fourth workgroup) — no spills at 119
__global__ void __launch_bounds__(TS * TS) __attribute__((amdgpu_waves_per_eu(TS == 16 ? 4 : 2, TS == 16 ? 4 : 8)))
raster_bwd_rows_kernel(const float* __restrict__ points, const float* __restrict__ feats,
                       const int32_t* __restrict__ ranges, const int32_t* __restrict__ o2p,
                       const float* __restrict__ image, const float* __restrict__ grad_image,
                       FastParams rp, float* __restrict__ moments, const int32_t* __restrict__ fixed_exp) {
  static_assert(TS == 8 || TS == 16, "tile 32 stays on raster_bwd_scan_kernel");
  constexpr int THREADS = TS * TS, WAVES = THREADS / 64, WAVES_WIDE = TS / 8;
  // LDS at tile 16: 216 x 52 B records + 4 x (4.5 KB accumulators + 0.25 KB patch list + 1 KB pool + 0.125 KB row table
  // + 1.25 KB pixels) = 39.9 KB: four workgroups per CU, as before
  constexpr int BATCH = TS == 8 ? 128 : (HEUR ? MS_R16_BATCH - 32 : MS_R16_BATCH);
  constexpr int BATCH_TARGET = TS == 8 ? 112 : BATCH - 12;
  constexpr int CAP = TS == 16 ? (HEUR ? MS_R16_CAP - 18 : MS_R16_CAP) : 128;
  static_assert(CAP <= 128, "quad masks are kept for two groups of 64 patch hits");
  constexpr int NACC = HEUR ? 11 : 9;
  constexpr int POOL_ROWS = MS_R_POOL_ROWS;           // 16-lane rows of list entries per round (>= 8: a list has <= 8 rows)
  static_assert(POOL_ROWS >= 8 && POOL_ROWS % 4 == 0, "pool");
  constexpr bool PIPELINED = THREADS >= 256;
  constexpr int PRIMARY = THREADS < BATCH ? THREADS : BATCH;
  constexpr int SLOTS_B = PIPELINED ? BATCH - PRIMARY : 0;
  static_assert(SLOTS_B >= 0 && SLOTS_B <= 64, "second staging slot: first wave only");
  __shared__ float4 s_rec[BATCH * 3];
  __shared__ int32_t s_id[BATCH];
  __shared__ float s_acc[WAVES][CAP][NACC];
  __shared__ uint16_t s_plist[WAVES][CAP];            // patch-list position -> staged index
  __shared__ uint8_t s_pool[WAVES][POOL_ROWS * 16];   // list entries (patch-list positions), packed by rows
  __shared__ uint16_t s_rowinfo[WAVES][POOL_ROWS];    // per row: quad | first row of its list << 4 | last << 5 | entries << 6
  // pixel state, entry p = 4 * quad + 2 * y + x (quad = 4 * qy + qx): [dL/dC.rgb, T] and <R, G>
  __shared__ float4 s_pix[WAVES][64];
  __shared__ float s_rg[WAVES][64];

  const int local_tile = (int)blockIdx.x;
  if (local_tile >= rp.num_tiles) return;
  const int tile_id = rp.tile_begin + local_tile;
  const int tile_u = tile_id % rp.tiles_wide, tile_v = tile_id / rp.tiles_wide;
  const int t = threadIdx.x, wave = t >> 6, lane = lane_id();
  const int patch_x = tile_u * TS + (wave % WAVES_WIDE) * 8;
  const int patch_y = tile_v * TS + (wave / WAVES_WIDE) * 8;

  {
    const int quad = lane >> 2;
    const int pix_x = patch_x + 2 * (quad & 3) + (lane & 1), pix_y = patch_y + 2 * (quad >> 2) + ((lane >> 1) & 1);
    float G0 = 0.f, G1 = 0.f, G2 = 0.f, RG = 0.f, T = 0.f;        // T = 0: out-of-image pixels never blend
    if (pix_x < rp.width && pix_y < rp.height) {
      const int64_t p = (int64_t)pix_y * rp.width + pix_x;
      G0 = grad_image[p * 3 + 0]; G1 = grad_image[p * 3 + 1]; G2 = grad_image[p * 3 + 2];
      RG = image[p * 3 + 0] * G0 + image[p * 3 + 1] * G1 + image[p * 3 + 2] * G2;   // <R, G>, R = forward image
      T = 1.0f;
    }
    s_pix[wave][lane] = make_float4(G0, G1, G2, T);
    s_rg[wave][lane] = RG;
  }
  float fixed_main = 0.f, fixed_h0 = 0.f;
  if (rp.deterministic) { fixed_main = ldexpf(1.0f, fixed_exp[0]); fixed_h0 = ldexpf(1.0f, fixed_exp[1]); }
  const float oms = rp.one_minus_saturate;
  const uint32_t oms_bits = __float_as_uint(oms);     // T >= 0: the float order is the order of the bit patterns
  const float E0 = -log2f(rp.alpha_threshold);
  const int crow = lane >> 4, col = lane & 15;

  for (int i = lane; i < CAP * NACC; i += 64) (&s_acc[wave][0][0])[i] = 0.0f;

  const int start = ranges[tile_id * 2 + 0], end = ranges[tile_id * 2 + 1];
  const int total = end - start;
  int num_batches = (total + BATCH_TARGET / 2) / BATCH_TARGET;
  if (num_batches < 1) num_batches = 1;
  int bsz = (total + num_batches - 1) / num_batches;
  if (bsz > BATCH) { num_batches = (total + BATCH - 1) / BATCH; bsz = (total + num_batches - 1) / num_batches; }

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

  for (int begin = start; begin < end; begin += bsz) {
    const int count = (end - begin) < bsz ? (end - begin) : bsz;
    wave_lds_fence();
    const bool tile_done = __syncthreads_and(__float_as_uint(s_pix[wave][lane].w) <= oms_bits);   // backward.py:116
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

    if (__ballot(__float_as_uint(s_pix[wave][lane].w) > oms_bits) == 0) continue;      // backward.py:142

    int r = 0;
    while (r < count) {
      // ---- cull, level 1: staged splats that can touch this wave's 8x8 patch -> patch list (conservative box test) ----
      int pcount = 0;
      const float pcx = (float)patch_x + 4.0f, pcy = (float)patch_y + 4.0f;
      while (r < count) {
        const int j = r + lane;
        const bool hit = j < count && scan_rect_hit(s_rec[j * 3 + 0], s_rec[j * 3 + 1], s_rec[j * 3 + 2], pcx, pcy, 3.5f);
        const unsigned long long m = __ballot(hit);
        const int nhit = __builtin_popcountll(m);
        if (pcount + nhit > CAP) break;
        const int ppos = pcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (hit) s_plist[wave][ppos] = (uint16_t)j;
        pcount += nhit;
        r += 64;
      }
      wave_lds_fence();

      // ---- cull, level 2: per patch hit the quads it can reach (lane = patch hit, two groups of 64) -------------------
      uint32_t qm[2] = {0u, 0u};
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        if (g * 64 < pcount) {
          const int ppos = g * 64 + lane;
          const bool in = ppos < pcount;
          const int j = in ? (int)s_plist[wave][ppos] : 0;
          const uint32_t m = quad_mask(s_rec[j * 3 + 0], s_rec[j * 3 + 1], (float)patch_x + 0.5f, (float)patch_y + 0.5f, E0);
          qm[g] = in ? m : 0u;
        }
      }
      int cnt[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int bit = 8 * (q >> 2) + 2 * (q & 3);
        cnt[q] = __builtin_popcountll(__ballot(((qm[0] >> bit) & 1u) != 0u)) +
                 __builtin_popcountll(__ballot(((qm[1] >> bit) & 1u) != 0u));
      }
#if MS_SCAN_STATS
      if (lane == 0) {
        int hits = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) hits += cnt[q];
        atomicAdd(&g_rows_stats[0], 1ull);                                   // passes
        atomicAdd(&g_rows_stats[1], (unsigned long long)hits);               // (quad, splat) hits
        atomicAdd(&g_rows_stats[6], (unsigned long long)pcount);             // (patch, splat) hits
      }
#endif

      // ---- rounds: as many whole lists as fit the pool (all sixteen, except when most patch hits reach most quads) ----
      int q_begin = 0;
      while (q_begin < 16) {
        int R = 0, q_end = q_begin;
        int rowoff[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          rowoff[q] = 0;
          if (q == q_end) {                                   // lists are taken in order while they fit
            const int rq = (cnt[q] + 15) >> 4;
            if (R + rq <= POOL_ROWS) { rowoff[q] = R; R += rq; q_end = q + 1; }
          }
        }
        wave_lds_fence();                                      // the previous round's chunks are done with pool / row table
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          if (q >= q_begin && q < q_end && cnt[q] > 0) {
            const int bit = 8 * (q >> 2) + 2 * (q & 3);
            const bool h0 = ((qm[0] >> bit) & 1u) != 0u;
            const unsigned long long m0 = __ballot(h0);
            const int base = rowoff[q] * 16;
            if (h0) s_pool[wave][base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m0, 0u))] = (uint8_t)lane;
            if (pcount > 64) {
              const bool h1 = ((qm[1] >> bit) & 1u) != 0u;
              const unsigned long long m1 = __ballot(h1);
              const int base1 = base + __builtin_popcountll(m0);
              if (h1) s_pool[wave][base1 + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u))] = (uint8_t)(64 + lane);
            }
            const int rq = (cnt[q] + 15) >> 4;
            if (lane < rq) {
              const int left = cnt[q] - lane * 16;
              s_rowinfo[wave][rowoff[q] + lane] =
                  (uint16_t)(q | (lane == 0 ? 16 : 0) | (lane == rq - 1 ? 32 : 0) | ((left < 16 ? left : 16) << 6));
            }
          }
        }
        q_begin = q_end;
        wave_lds_fence();

        // ---- blend: chunks of four rows, lane = splat, two pixel-pair steps ------------------------------------------
        const int nchunks = (R + 3) >> 2;
#pragma unroll 1
        for (int c = 0; c < nchunks; ++c) {
          wave_lds_fence();                  // pixel state and accumulator rows written by other lanes in earlier chunks
          const int row = 4 * c + crow;
          const uint32_t info = row < R ? (uint32_t)s_rowinfo[wave][row] : 0u;
          const bool valid = col < (int)(info >> 6);
          const int quad = (int)(info & 15u);
          // this row continues, inside this chunk, the list of the row above it
          const bool cont = row < R && (info & 16u) == 0u && crow > 0;
          const unsigned long long contmask = __ballot(cont);
          const unsigned long long exec15 = 0x0000ffff0000ffffull | (contmask & 0xffff0000ffff0000ull);
          const unsigned long long cont2 = contmask & 0x0000ffff00000000ull;
          const unsigned long long exec31 = 0x00000000ffffffffull | cont2 | (cont2 != 0ull ? (contmask & 0xffff000000000000ull) : 0ull);
          const bool seg_start = col == 0 && !cont;                                    // seeds T from its quad's pixel
          const bool seg_end = col == 15 && row < R && (crow == 3 || (info & 32u) != 0u);     // writes the pixel state back

          const int pos = valid ? (int)s_pool[wave][c * 64 + lane] : 0;
          const int idx = (int)s_plist[wave][pos];
          const float4 q0 = s_rec[idx * 3 + 0], q1 = s_rec[idx * 3 + 1], q2 = s_rec[idx * 3 + 2];
          const float A = q0.z, B = q0.w, C = q1.x, D = q1.y;
          const float nl2a = valid ? q1.z : __builtin_inff();       // idle lanes: alpha g = exp2(-inf) = 0
          const float f0 = q1.w, f1 = q2.x, f2 = q2.y;
          const int pbase = quad * 4;
          const float dx0 = ((float)(patch_x + 2 * (quad & 3)) + 0.5f) - q0.x;
          const float dy0 = ((float)(patch_y + 2 * (quad >> 2)) + 0.5f) - q0.y;
          const float X00 = __builtin_fmaf(A, dx0, B * dy0), Y00 = __builtin_fmaf(C, dx0, D * dy0);

          float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
          float h0 = 0.f, h1 = 0.f;
#if MS_SCAN_STATS
          int steps_run = 0, lanes_contrib = 0;
#endif
          float4 pg[2];
          float2 prg;
          pg[0] = s_pix[wave][pbase]; pg[1] = s_pix[wave][pbase + 1];
          prg = *reinterpret_cast<const float2*>(&s_rg[wave][pbase]);
#pragma unroll
          for (int y = 0; y < 2; ++y) {
            const int p = pbase + 2 * y;
            const float4 cur[2] = {pg[0], pg[1]};
            const float RGin[2] = {prg.x, prg.y};
            const bool any_alive = __float_as_uint(cur[0].w) > oms_bits || __float_as_uint(cur[1].w) > oms_bits;
            if (y == 0) {
              pg[0] = s_pix[wave][pbase + 2]; pg[1] = s_pix[wave][pbase + 3];
              prg = *reinterpret_cast<const float2*>(&s_rg[wave][pbase + 2]);
            }
            // wave-uniform: a step in which every row's pixel pair is saturated / out of the image is skipped
            if (__ballot(any_alive) == 0) continue;

            float X[2], Y[2], a_gated[2], a[2], om[2], Tk[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const float Xy = y == 0 ? X00 : X00 + B, Yy = y == 0 ? Y00 : Y00 + D;
              X[u] = u == 0 ? Xy : Xy + A;
              Y[u] = u == 0 ? Yy : Yy + C;
              const float a_raw = __builtin_amdgcn_exp2f(-__builtin_fmaf(X[u], X[u], __builtin_fmaf(Y[u], Y[u], nl2a)));
              a_gated[u] = a_raw > rp.alpha_threshold ? a_raw : 0.0f;                 // blend gate (forward.py:99-101)
              a[u] = min_f32_uniform(a_gated[u], rp.clamp_max_alpha);
              om[u] = 1.0f - a[u];
              // T before this splat: the neighbour's 1 - alpha shifted in; the first lane of a list segment takes its pixel's T
              const float shifted = dpp_f32<0x138>(cur[u].w, om[u]);                  // wave_shr:1
              Tk[u] = seg_start ? cur[u].w : shifted;
            }
            wave_scan_mul2_rows(Tk[0], Tk[1], exec15, exec31);
            // saturation skip (backward.py:154), wave-uniform slow path as in raster_bwd_scan.hip
            float a_st[2];
            bool any_sat = false;
#pragma unroll
            for (int u = 0; u < 2; ++u) { a_st[u] = a_gated[u]; any_sat |= !(Tk[u] > oms); }
            if (__ballot(any_sat) != 0) {
              asm volatile("; saturation inside the chunk" ::: "memory");
#pragma unroll
              for (int u = 0; u < 2; ++u) {
                const bool live = Tk[u] > oms;
                a[u] = live ? a[u] : 0.0f;
                a_st[u] = live ? a_gated[u] : 0.0f;
              }
            }
            float w[2], fG[2], S[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              w[u] = a[u] * Tk[u];
              fG[u] = __builtin_fmaf(f2, cur[u].z, __builtin_fmaf(f1, cur[u].y, f0 * cur[u].x));
              S[u] = w[u] * fG[u];
            }
            wave_scan_add2_rows(S[0], S[1], exec15, exec31);
            float RGout[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) RGout[u] = RGin[u] - S[u];                    // R -= f w  (backward.py:171-174)
            // the last lane of every list segment holds its quad's pixel pair after the segment's splats (idle lanes
            // behind the last entry carry alpha = 0, so lane 15 of the row is as good as the last valid lane)
            if (seg_end) {
              s_pix[wave][p].w = Tk[0] * om[0];
              s_pix[wave][p + 1].w = Tk[1] * om[1];
              *reinterpret_cast<float2*>(&s_rg[wave][p]) = make_float2(RGout[0], RGout[1]);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const float RGk = RGout[u];
              const float ag = __builtin_fmaf(Tk[u], fG[u], -(RGk * __builtin_amdgcn_rcpf(om[u])));    // d(alpha)
              const float q_ = ag * a_st[u];                          // straight-through clamp (backward.py:158-163)
              const float qX = q_ * X[u], qY = q_ * Y[u];
              m0 += q_; m1 += qX; m2 += qY;
              m3 = __builtin_fmaf(qX, X[u], m3); m4 = __builtin_fmaf(qX, Y[u], m4); m5 = __builtin_fmaf(qY, Y[u], m5);
              a0 = __builtin_fmaf(w[u], cur[u].x, a0); a1 = __builtin_fmaf(w[u], cur[u].y, a1); a2 = __builtin_fmaf(w[u], cur[u].z, a2);
              if (HEUR) {                                           // backward.py:190-194
                const float agm = a_st[u] != 0.0f ? ag : 0.0f;
                h0 = __builtin_fmaf(agm, agm, h0);
                h1 += fabsf(__builtin_fmaf(qX, A, qY * C)) + fabsf(__builtin_fmaf(qX, B, qY * D));
              }
#if MS_SCAN_STATS
              lanes_contrib += __builtin_popcountll(__ballot(w[u] != 0.0f));
#endif
            }
#if MS_SCAN_STATS
            steps_run += 2;
#endif
          }
#if MS_SCAN_STATS
          if (lane == 0) {
            atomicAdd(&g_rows_stats[2], 1ull);                                             // chunks
            atomicAdd(&g_rows_stats[3], (unsigned long long)__builtin_popcountll(__ballot(valid)));   // filled lanes (lane 0 only adds once)
            atomicAdd(&g_rows_stats[4], (unsigned long long)steps_run);                    // executed pixel steps
            atomicAdd(&g_rows_stats[5], (unsigned long long)lanes_contrib);                // contributing (pixel, splat) pairs
          }
#endif

          // The wave's accumulator row of the splat.  Within a list segment the lanes hold distinct splats; two
          // segments of one chunk (different quads) may hold the SAME splat, so the segments take turns.
          {
            const int rows_here = (R - 4 * c) < 4 ? (R - 4 * c) : 4;
            const unsigned controws = (unsigned)((contmask >> 15) & 2ull) | (unsigned)((contmask >> 30) & 4ull) | (unsigned)((contmask >> 45) & 8ull);
            const unsigned startrows = ~controws & ((1u << rows_here) - 1u);
            const int nseg = __builtin_popcount(startrows);
            const int seg = __builtin_popcount(startrows & ((2u << crow) - 1u)) - 1;
            float* accrow = &s_acc[wave][pos][0];
            const float v[11] = {m0, m1, m2, m3, m4, m5, a0, a1, a2, h0, h1};
            for (int sgi = 0; sgi < nseg; ++sgi) {
              if (valid && seg == sgi) {
#pragma unroll
                for (int k = 0; k < NACC; ++k) accrow[k] += v[k];
              }
              wave_lds_fence();
            }
          }
        }
      }

      // ---- commit the pass: ONE 64-byte, line-aligned row of global float atomics per (patch, splat) ------------------
      wave_lds_fence();
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
  }
}

}  // namespace ms

using namespace ms;

// launcher shared with raster_bwd_scan.hip (frame_internal.h): returns false when this kernel does not serve the
// configuration (tile 32) or MS_RASTER_BWD=scan asks for the round-2/3 kernel
namespace ms {
bool rows_backward_enabled() {
  static const bool on = [] { const char* e = getenv("MS_RASTER_BWD"); return !(e && e[0] == 's'); }();
  return on;
}

bool launch_rows_backward(const float* points7, const float* features, const int32_t* tile_ranges,
                          const int32_t* overlap_to_point, const float* image, const float* grad_image,
                          const FastParams& rp, int tile_size, bool heuristics, float* moments, const int32_t* fixed_exp,
                          hipStream_t s) {
  if (!rows_backward_enabled() || (tile_size != 8 && tile_size != 16)) return false;
#define MS_GO(TS, HEUR) raster_bwd_rows_kernel<TS, HEUR><<<dim3((unsigned)rp.num_tiles), dim3(TS * TS), 0, s>>>(        \
      points7, features, tile_ranges, overlap_to_point, image, grad_image, rp, moments, fixed_exp)
  if (tile_size == 8) { if (heuristics) MS_GO(8, true); else MS_GO(8, false); }
  else { if (heuristics) MS_GO(16, true); else MS_GO(16, false); }
#undef MS_GO
  return true;
}
}  // namespace ms

#if MS_SCAN_STATS
extern "C" int ms_debug_rows_stats(unsigned long long* out12, int reset) {
  if (out12) (void)hipMemcpyFromSymbol(out12, HIP_SYMBOL(g_rows_stats), 12 * sizeof(unsigned long long));
  if (reset) { unsigned long long z[12] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rows_stats), z, sizeof(z)); }
  return 0;
}
#endif
