// raster_fast.hip — the product-path raster kernels: float32, RGB (F = 3), plain gaussian pdf,
// alpha blending.  Same algorithm and work decomposition as the generic templates in raster.hip
// (one workgroup per tile, one wave64 per 8x8 pixel patch, one lane per pixel; semantics of
// rasterizer/forward.py:39-135 and rasterizer/backward.py:97-224), hand-tuned for gfx950:
//
//   * 48-byte LDS splat record read as three ds_read_b128 broadcasts, software-prefetched one hit
//     ahead so LDS latency overlaps the arithmetic of the current splat;
//   * exact-ish culling per wave: axis-aligned extent test AND the oriented-box separating-axis
//     test of the tile mapper (grid_query.py:30-43) against the patch's pixel-centre rectangle,
//     one staged splat per lane, ballot -> scalar walk over the hits;
//   * no exec-mask divergence in the hot loop: contributions are predicated (v_cndmask);
//   * transmittance T = 1 - W is carried instead of W (w = alpha * T; T -= w);
//   * backward: halving-butterfly wave reduction (quad_perm DPP + v_permlane16/32_swap) and ONE
//     global_atomic_add_f32 per (patch, splat) — see wave_reduce16;
//   * the next batch's gather (overlap_to_point -> packed gaussian + colour) is issued before the
//     current batch is consumed, hiding the dependent HBM/L2 gather latency behind the blend loop.
#include "raster_common.h"

#ifndef MS_ABLATE
#define MS_ABLATE 0
#endif
// Forward variants (round 6, tools/build_variant.sh; profiles/r06_raster_fwd_phase_split.txt):
//   MS_FWD_PHASES      every wave adds up the shader cycles it spends per phase (ms_debug_fwd_phases, tools/rbench.py)
//   MS_FWD_CLAMP_FOLD  the clamp of alpha to clamp_max_alpha as the free `clamp` output modifier of v_exp_f32: the
//                      exponent carries + log2(clamp_max), the pixel state is T' = clamp_max T (w = a' T', T' -= clamp_max w)
//   MS_FWD_SKIP_EMPTY  a hit none of whose 64 pixels passes the blend gate (the cull is conservative) leaves before
//                      the colour read and the blend arithmetic
#ifndef MS_FWD_PHASES
#define MS_FWD_PHASES 0
#endif
#ifndef MS_FWD_CLAMP_FOLD
#define MS_FWD_CLAMP_FOLD 1
#endif
#ifndef MS_FWD_SKIP_EMPTY
#define MS_FWD_SKIP_EMPTY 0
#endif
//   MS_FWD_PAIR        the hit walk takes two hits per iteration (both records requested before either is used)
#ifndef MS_FWD_PAIR
#define MS_FWD_PAIR 1
#endif
// Hit-loop flavour (both measured on config D): the forward walks the hit mask with the record of the next hit
// requested one iteration ahead; the backward body is long enough for the other waves of the SIMD to hide the
// LDS latency and uses the leaner walk (one s_ff1 + one bit clear + one v_readlane per hit).

namespace ms {

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// VIS: also accumulate each splat's visibility = the sum of its blend weights over all pixels
// (forward.py:126-131): one 6-step DPP sum per (patch, splat) hit with a contribution, summed over the
// tile's patches in LDS (single-lane ds_add_f32) and committed with ONE global atomic per (tile, splat) —
// the pass is bound by the global atomic rate, so the count is what matters.
#if MS_FWD_PHASES
// 0 barrier at the top of a batch (flags)   1 staging (records + the next gathers issued)   2 barrier after staging
// 3 cull (LDS reads, test, ballot)   4 hit walk   5 visibility flush   6 epilogue   7 start .. first batch   8 whole wave
// 9 hits   10 batches   11 waves
__device__ unsigned long long* g_fwd_phase_rows = nullptr;
#define MS_FPH(i) do { const uint64_t ph_now = __builtin_readcyclecounter(); ph_acc[i] += (uint32_t)(ph_now - ph_last); ph_last = ph_now; } while (0)
#else
#define MS_FPH(i) do {} while (0)
#endif
constexpr float FWD_SPENT_T = 1.3552527e-20f;   // 2^-66: transmittance below which the rest of a tile's list cannot change a pixel
constexpr unsigned FWD_XCD_CHUNK = 8;      // tiles per XCD run (xcd_tile, raster_common.h)
constexpr unsigned BWD_XCD_CHUNK = 8;      // pixel-per-lane backward: 3.12 -> 3.08 ms at tile 32, 2.84 -> 2.79 at tile 16

#if MS_FWD_CLAMP_FOLD
// exp2 of the biased exponent = alpha g / clamp_max; clamp(., 0, 1) is the v_exp_f32's own output modifier (no v_med3)
#define FWD_ALPHA_BIAS (__builtin_amdgcn_logf(rp.clamp_max_alpha))
__device__ __forceinline__ float fwd_alpha(float e, float) { return __builtin_amdgcn_fmed3f(e, 0.0f, 1.0f); }
#else
#define FWD_ALPHA_BIAS 0.0f
__device__ __forceinline__ float fwd_alpha(float e, float clamp_max_alpha) { return clamp_alpha(e, clamp_max_alpha); }
#endif

template <int TS, bool VIS, bool ROWS, bool SEGS = false>   // ROWS: `points` is a splat-row table (common.h), `feats` unused;
__global__ void __launch_bounds__(TS * TS)                  // SEGS: one workgroup per SEGMENT of a long tile run (raster_common.h)
raster_fwd_f32x3_kernel(const float* __restrict__ points, const float* __restrict__ feats,
                        const int32_t* __restrict__ ranges, const int32_t* __restrict__ o2p,
                        FastParams rp, float* __restrict__ image, float* __restrict__ image_alpha,
                        float* __restrict__ visibility) {
  using G = TileGeom<TS>;
  constexpr int BATCH = G::BATCH;
  __shared__ float4 s_rec[BATCH * 3];
  __shared__ float4 s_cull[BATCH * 2];
  __shared__ int32_t s_id[VIS ? BATCH : 1];
  __shared__ float s_vis[VIS ? BATCH : 1];
#if MS_FWD_PHASES
  const uint64_t ph_start = __builtin_readcyclecounter();
  uint64_t ph_last = ph_start;
  uint32_t ph_acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

  int tile_id, start, end;
  if constexpr (SEGS) {
    // (a plan that overflowed its capacities — a caller that passed k_capacity below the real overlap count — is not
    // executed: the per-tile launch has rendered every tile)
    if (rp.split_counts[2] != 0 || (int)blockIdx.x >= rp.split_counts[0]) return;
    const int4 item = rp.split_items[blockIdx.x];
    tile_id = item.x; start = item.y; end = item.z;
  } else {
    unsigned part_;
    const int local_tile = xcd_tile<FWD_XCD_CHUNK>(rp.num_tiles, blockIdx.x, 1, &part_);
    if (local_tile < 0) return;
    tile_id = rp.tile_begin + local_tile;
    start = ranges[tile_id * 2 + 0]; end = ranges[tile_id * 2 + 1];
    // a scene shape that shows a run above the default limit is rendered with the segment launches from its next frame
    // on (frame.py)
    if (end - start > SPLIT_MIN_RUN && rp.long_run_word && threadIdx.x == 0) *rp.long_run_word = end - start;
    // this frame already is: the segment launch has the tile (unless the plan overflowed, see above)
    if (rp.split_min_run > 0 && end - start > rp.split_min_run && rp.split_counts[2] == 0) return;
  }
  const int tile_u = tile_id % rp.tiles_wide, tile_v = tile_id / rp.tiles_wide;
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int patch_x = tile_u * TS + (wave % G::WAVES_WIDE) * 8;
  const int patch_y = tile_v * TS + (wave / G::WAVES_WIDE) * 8;
  const int pix_x = patch_x + (lane & 7), pix_y = patch_y + (lane >> 3);
  const float px = (float)pix_x + 0.5f, py = (float)pix_y + 0.5f;
  const float rcx = (float)patch_x + 4.0f, rcy = (float)patch_y + 4.0f;
  const bool in_bounds = pix_x < rp.width && pix_y < rp.height;
  const float origin_x = (float)(tile_u * TS) + 0.5f * TS, origin_y = (float)(tile_v * TS) + 0.5f * TS;
  const float pxr = px - origin_x, pyr = py - origin_y;

  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  // transmittance = 1 - accumulated weight, carried as T' = TSCALE T (MS_FWD_CLAMP_FOLD: TSCALE = clamp_max_alpha)
#if MS_FWD_CLAMP_FOLD
  const float TSCALE = rp.clamp_max_alpha, GATE = rp.alpha_threshold / rp.clamp_max_alpha;
#else
  const float TSCALE = 1.0f, GATE = rp.alpha_threshold;
#endif
  float T = in_bounds ? TSCALE : 0.0f;
  // SEGS + VIS = the SECOND walk of a segment, behind the composition pass: it starts from the true transmittance at
  // the segment's start (what the composition left in the state) and exists for the visibility sums only
  if constexpr (SEGS && VIS) T = TSCALE * rp.split_state[(int64_t)blockIdx.x * (TS * TS) + threadIdx.x].w;
  const float SPENT = FWD_SPENT_T * TSCALE;

  const int t = threadIdx.x;

  // two-deep gather pipeline: `raw` = splat data of the batch about to be staged, `next_id` = point
  // index of the batch after it, so neither dependent load is waited for inside the blend loop
  Raw raw;
  int next_id = 0;
  const bool stager = t < BATCH;
  if (stager && start + t < end) raw = load_raw<ROWS>(points, feats, o2p[start + t]);
  if (stager && start + BATCH + t < end) next_id = o2p[start + BATCH + t];
  // (the spent-tile exit below may leave the loop before the first batch is staged — the second walk of a segment that
  // sits behind an opaque surface: the flush behind the loop must then find zeros, not whatever LDS held)
  if (VIS && stager) s_vis[t] = 0.0f;

  for (int begin = start; begin < end; begin += BATCH) {
    const int count = (end - begin) < BATCH ? (end - begin) : BATCH;
    // Spent tiles (round 5).  The reference's blending forward never leaves its loop (forward.py:69-70 tests a flag
    // only the non-blending branch sets) and neither did rounds 1-4: a tile with 240 000 splats behind an opaque surface
    // cost 15 ms in ONE workgroup (tools/sweep_scenes.py, the pile-up scene).  Once every pixel of the tile has
    // T < 2^-66 whatever is left of the list adds less than 255 T max|f| ~ 3.5e-18 max|f| to a pixel — nothing a
    // float32 image of these features can hold, ten orders of magnitude inside the 1e-4 contract — and image_alpha =
    // 1 - T is exactly 1 either way: the workgroup stops, and a wave whose own 64 pixels are spent skips its share of
    // a batch.  (A visibility sum loses the same < 1e-17 per splat.)  One barrier: each wave posts its flag before it.
    // The flags live in the last word of records 0 .. waves - 1, which nothing else reads or writes (write_records<true>
    // stores 40 of a record's 48 bytes): an LDS array of their own cost the kernel its eighth workgroup per CU (20 496
    // bytes instead of 20 480) and, with this compiler, 21 VGPRs (65 instead of 44-46) — the forward of config D ran
    // 0.56-0.59 ms instead of 0.52-0.54 for most of round 5 (tests/test_kernel_budgets.py holds it now).
    if (begin == start) MS_FPH(7);
    const bool wave_spent = __ballot(T >= SPENT) == 0;
    if (lane == 0) reinterpret_cast<int*>(&s_rec[wave * 3 + 2])[3] = wave_spent ? 1 : 0;
    __syncthreads();                       // previous batch fully consumed; flags posted
    if (__ballot(reinterpret_cast<const int*>(&s_rec[(lane % (TS * TS / 64)) * 3 + 2])[3] != 0) == ~0ull) break;
    MS_FPH(0);
    if (VIS && stager) {
      if (begin > start && s_vis[t] != 0.0f) atomic_add_noret(visibility + s_id[t], s_vis[t]);   // previous batch
      s_vis[t] = 0.0f;
    }
    MS_FPH(5);
    if (stager && begin + t < end) {
      write_records<true>(raw, rp.alpha_threshold, &s_rec[t * 3], &s_cull[t * 2], origin_x, origin_y, FWD_ALPHA_BIAS);
      if (VIS) s_id[t] = raw.id;
    }
    if (stager && begin + BATCH + t < end) raw = load_raw<ROWS>(points, feats, next_id);
    if (stager && begin + 2 * BATCH + t < end) next_id = o2p[begin + 2 * BATCH + t];
    MS_FPH(1);
    __syncthreads();
    MS_FPH(2);
#if MS_FWD_PHASES
    ph_acc[10] += 1;
#endif
    if (wave_spent) continue;

    for (int r = 0; r < count; r += 64) {
      const int j = r + lane;
      bool hit = false;
      if (j < count) hit = patch_hit(s_cull[j * 2], s_cull[j * 2 + 1], rcx, rcy);
      unsigned long long m = __ballot(hit);
      MS_FPH(3);
      if (m == 0) continue;
#if MS_FWD_PHASES
      ph_acc[9] += (uint32_t)__popcll(m);
#endif
      if constexpr (VIS) {
        // visibility = per-splat sum of the blend weights (forward.py:126-131): the hits are taken four at a time
        // and ONE transposing reduction (wave_reduce4, 17 VALU) yields the four sums — 6 DPP adds + a ballot per hit
        // before; lanes 60..63 add them to the tile's LDS row, the stager commits that row once per (tile, splat)
        while (m != 0) {
          float wq[4] = {0.f, 0.f, 0.f, 0.f};
          int bq[4] = {0, 0, 0, 0};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (m != 0) {                                   // wave-uniform
              const int b = __builtin_ctzll(m);
              asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(b));
              bq[u] = b;
              const float4 q0 = s_rec[(r + b) * 3 + 0], q1 = s_rec[(r + b) * 3 + 1];
              const float2 q2 = *reinterpret_cast<const float2*>(&s_rec[(r + b) * 3 + 2]);
              const float X = __builtin_fmaf(pxr, q0.z, __builtin_fmaf(pyr, q0.w, -q0.x));
              const float Y = __builtin_fmaf(pxr, q1.x, __builtin_fmaf(pyr, q1.y, -q0.y));
              const float a = fwd_alpha(__builtin_amdgcn_exp2f(-__builtin_fmaf(Y, Y, __builtin_fmaf(X, X, q1.z))), rp.clamp_max_alpha);
              const float w = a > GATE ? a * T : 0.0f;
              T = __builtin_fmaf(-TSCALE, w, T);
              c0 += q1.w * w; c1 += q2.x * w; c2 += q2.y * w;
              wq[u] = w;
            }
          }
          const float total = wave_reduce4(wq, (lane & 1) != 0, (lane & 2) != 0);
          if (lane >= 60 && total != 0.0f) {
            const int k = lane & 3;
            atomicAdd(&s_vis[r + (k == 0 ? bq[0] : k == 1 ? bq[1] : k == 2 ? bq[2] : bq[3])], total);
          }
        }
        continue;
      }
      // plain hit loop: one scalar bit scan + bit clear per hit (the compiler turns a hand-written prefetch of the next
      // record into a dozen scalar instructions per hit and drops the prefetch itself).  Round 6: hits are taken TWO at
      // a time — a wave alone needs ~190 cycles per hit (LDS round trip, then fourteen dependent VALU instructions)
      // while its share of the SIMD's issue slots is ~40: with both records requested up front and the two alpha chains
      // independent until the transmittance, the scheduler interleaves them (profiles/r06_raster_fwd_phase_split.txt)
#if MS_FWD_PAIR
      while (m & (m - 1)) {                                // at least two hits left
        const int b0 = __builtin_ctzll(m);
        asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(b0));
        const int b1 = __builtin_ctzll(m);
        asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(b1));
        const float4 p0 = s_rec[(r + b0) * 3 + 0], p1 = s_rec[(r + b0) * 3 + 1];
        const float4 q0 = s_rec[(r + b1) * 3 + 0], q1 = s_rec[(r + b1) * 3 + 1];
        const float2 p2 = *reinterpret_cast<const float2*>(&s_rec[(r + b0) * 3 + 2]);
        const float2 q2 = *reinterpret_cast<const float2*>(&s_rec[(r + b1) * 3 + 2]);
        const float X0 = __builtin_fmaf(pxr, p0.z, __builtin_fmaf(pyr, p0.w, -p0.x));
        const float X1 = __builtin_fmaf(pxr, q0.z, __builtin_fmaf(pyr, q0.w, -q0.x));
        const float Y0 = __builtin_fmaf(pxr, p1.x, __builtin_fmaf(pyr, p1.y, -p0.y));
        const float Y1 = __builtin_fmaf(pxr, q1.x, __builtin_fmaf(pyr, q1.y, -q0.y));
        const float a0 = fwd_alpha(__builtin_amdgcn_exp2f(-__builtin_fmaf(Y0, Y0, __builtin_fmaf(X0, X0, p1.z))), rp.clamp_max_alpha);
        const float a1 = fwd_alpha(__builtin_amdgcn_exp2f(-__builtin_fmaf(Y1, Y1, __builtin_fmaf(X1, X1, q1.z))), rp.clamp_max_alpha);
        const float w0 = a0 > GATE ? a0 * T : 0.0f;
        T = __builtin_fmaf(-TSCALE, w0, T);
        c0 += p1.w * w0; c1 += p2.x * w0; c2 += p2.y * w0;
        const float w1 = a1 > GATE ? a1 * T : 0.0f;
        T = __builtin_fmaf(-TSCALE, w1, T);
        c0 += q1.w * w1; c1 += q2.x * w1; c2 += q2.y * w1;
      }
#endif
      while (m != 0) {
        const int b = __builtin_ctzll(m);
        asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(b));
        // the forward uses 40 of the record's 48 bytes: the third read is 64 bits
        const float4 q0 = s_rec[(r + b) * 3 + 0], q1 = s_rec[(r + b) * 3 + 1];
#if !MS_FWD_SKIP_EMPTY
        const float2 q2 = *reinterpret_cast<const float2*>(&s_rec[(r + b) * 3 + 2]);
#endif
        // (X, Y) = basis * (pixel - mean), expanded around the tile centre (write_records<true>)
        const float X = __builtin_fmaf(pxr, q0.z, __builtin_fmaf(pyr, q0.w, -q0.x));
        const float Y = __builtin_fmaf(pxr, q1.x, __builtin_fmaf(pyr, q1.y, -q0.y));
        // alpha * g in one exponential: A..D pre-scaled, q1.z = -log2(alpha) [+ log2(clamp_max)] (write_records<true>)
        const float a = fwd_alpha(__builtin_amdgcn_exp2f(-__builtin_fmaf(Y, Y, __builtin_fmaf(X, X, q1.z))), rp.clamp_max_alpha);
        // (the gate as arithmetic — clamp((E0 - e) 2^40, 0, 1) folded into an FMA, then a multiply — was measured in
        // round 4: 0.545 -> 0.560 ms on the same box; the compare + select stays)
        const bool pass = a > GATE;
#if MS_FWD_SKIP_EMPTY
        if (__ballot(pass) == 0) continue;
        const float2 q2 = *reinterpret_cast<const float2*>(&s_rec[(r + b) * 3 + 2]);
#endif
        const float w = pass ? a * T : 0.0f;
        T = __builtin_fmaf(-TSCALE, w, T);
        c0 += q1.w * w; c1 += q2.x * w; c2 += q2.y * w;
      }
      MS_FPH(4);
    }
  }
  MS_FPH(6);

  if (VIS && end > start) {                // last batch
    __syncthreads();
    if (stager && s_vis[t] != 0.0f) atomic_add_noret(visibility + s_id[t], s_vis[t]);
  }

#if MS_FWD_CLAMP_FOLD
  T = T / TSCALE;                          // back to the transmittance itself (correctly rounded division, once per pixel)
#endif
#if MS_FWD_PHASES
  if (g_fwd_phase_rows && lane == 0) {
    unsigned long long* row = g_fwd_phase_rows + ((size_t)blockIdx.x * (TS * TS / 64) + wave) * 12;
    const uint64_t now = __builtin_readcyclecounter();
    ph_acc[6] += (uint32_t)(now - ph_last);
    for (int i = 0; i < 8; ++i) row[i] = ph_acc[i];
    row[8] = now - ph_start; row[9] = ph_acc[9]; row[10] = ph_acc[10]; row[11] = 1;
  }
#endif
  if constexpr (SEGS) {
    // (C_s, P_s) of this segment for the pixel; out-of-image pixels carry T = 0 throughout
    if constexpr (!VIS) rp.split_state[(int64_t)blockIdx.x * (TS * TS) + t] = make_float4(c0, c1, c2, T);
    return;
  }
  if (in_bounds) {
    const int64_t p = (int64_t)pix_y * rp.width + pix_x;
    image[p * 3 + 0] = c0; image[p * 3 + 1] = c1; image[p * 3 + 2] = c2;
    image_alpha[p] = 1.0f - T;
  }
}

// One thread per tile of the launch: tiles whose run exceeds SPLIT_MIN_RUN are cut into <= SPLIT_MAX_SEG segments of
// >= seg_len entries (multiples of 256: every batch size divides them).  counts[0..2] are zero on entry.  The item
// order depends on the order of the atomics; nothing else does (results are addressed by item).
__global__ void __launch_bounds__(256)
split_plan_kernel(const int32_t* __restrict__ ranges, int tile_begin, int num_tiles, int min_run, int seg_len, int item_cap,
                  int long_cap, int32_t* __restrict__ counts, int4* __restrict__ long_tiles, int4* __restrict__ items) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_tiles) return;
  const int tile = tile_begin + i;
  const int start = ranges[tile * 2 + 0], end = ranges[tile * 2 + 1], run = end - start;
  if (run <= min_run) return;
  int nseg = (run + seg_len - 1) / seg_len;
  if (nseg > SPLIT_MAX_SEG) nseg = SPLIT_MAX_SEG;
  const int seg = ((run + nseg - 1) / nseg + 255) & ~255;
  nseg = (run + seg - 1) / seg;
  const int first = atomicAdd(&counts[0], nseg), li = atomicAdd(&counts[1], 1);
  // the capacities are upper bounds when k_capacity >= K; otherwise the whole plan is void (the kernels that read it
  // test counts[2] first and the per-tile launches take every tile)
  if (first + nseg > item_cap || li >= long_cap) { counts[2] = 1; return; }
  long_tiles[li] = make_int4(tile, first, nseg, 0);
  for (int k = 0; k < nseg; ++k) {
    const int b = start + k * seg, e = b + seg < end ? b + seg : end;
    items[first + k] = make_int4(tile, b, e, first);
  }
}

// One workgroup per long tile, thread = pixel in the forward kernel's order: front-to-back composition of the
// segments' (C_s, P_s); each state row is replaced by (colour in front of the segment, transmittance at its start).
template <int TS>
__global__ void __launch_bounds__(TS * TS)
split_combine_kernel(FastParams rp, const int4* __restrict__ long_tiles, float* __restrict__ image,
                     float* __restrict__ image_alpha) {
  using G = TileGeom<TS>;
  if (rp.split_counts[2] != 0 || (int)blockIdx.x >= rp.split_counts[1]) return;
  const int4 lt = long_tiles[blockIdx.x];
  const int tile_id = lt.x, first = lt.y, nseg = lt.z;
  const int tile_u = tile_id % rp.tiles_wide, tile_v = tile_id / rp.tiles_wide;
  const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
  const int pix_x = tile_u * TS + (wave % G::WAVES_WIDE) * 8 + (lane & 7);
  const int pix_y = tile_v * TS + (wave / G::WAVES_WIDE) * 8 + (lane >> 3);
  const bool in_bounds = pix_x < rp.width && pix_y < rp.height;
  float c0 = 0.f, c1 = 0.f, c2 = 0.f, T = in_bounds ? 1.0f : 0.0f;
  float4* row = rp.split_state + (int64_t)first * (TS * TS) + t;
  float4 v = row[0];
  for (int k = 0; k < nseg; ++k) {
    const float4 nv = k + 1 < nseg ? row[(int64_t)(k + 1) * (TS * TS)] : v;
    row[(int64_t)k * (TS * TS)] = make_float4(c0, c1, c2, T);
    c0 = __builtin_fmaf(T, v.x, c0); c1 = __builtin_fmaf(T, v.y, c1); c2 = __builtin_fmaf(T, v.z, c2);
    T *= v.w;
    v = nv;
  }
  if (in_bounds) {
    const int64_t p = (int64_t)pix_y * rp.width + pix_x;
    image[p * 3 + 0] = c0; image[p * 3 + 1] = c1; image[p * 3 + 2] = c2;
    image_alpha[p] = 1.0f - T;
  }
}


// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
template <int TS, bool HEUR>
__global__ void __launch_bounds__(TS * TS)
raster_bwd_f32x3_kernel(const float* __restrict__ points, const float* __restrict__ feats,
                        const int32_t* __restrict__ ranges, const int32_t* __restrict__ o2p,
                        const float* __restrict__ image, const float* __restrict__ grad_image,
                        FastParams rp, float* __restrict__ grad_points, float* __restrict__ grad_feats,
                        float* __restrict__ heuristic) {
  using G = TileGeom<TS>;
  constexpr int BATCH = G::BATCH;
  __shared__ float4 s_rec[BATCH * 3];
  __shared__ float4 s_cull[BATCH * 2];
  // element offsets of the staged splats' gradient rows, one array per output (id * 7, id * 3, id * 2): the
  // stager multiplies once per splat instead of every (patch, splat) commit doing it
  constexpr int NOUT = HEUR ? 3 : 2;
  __shared__ uint32_t s_off[NOUT][BATCH];

  unsigned part_;
  const int local_tile = xcd_tile<BWD_XCD_CHUNK>(rp.num_tiles, blockIdx.x, 1, &part_);
  if (local_tile < 0) return;
  const int tile_id = rp.tile_begin + local_tile;
  const int tile_u = tile_id % rp.tiles_wide, tile_v = tile_id / rp.tiles_wide;
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int patch_x = tile_u * TS + (wave % G::WAVES_WIDE) * 8;
  const int patch_y = tile_v * TS + (wave / G::WAVES_WIDE) * 8;
  const int pix_x = patch_x + (lane & 7), pix_y = patch_y + (lane >> 3);
  const float px = (float)pix_x + 0.5f, py = (float)pix_y + 0.5f;
  const float rcx = (float)patch_x + 4.0f, rcy = (float)patch_y + 4.0f;
  const bool in_bounds = pix_x < rp.width && pix_y < rp.height;

  // per-pixel state (backward.py:97-110): T = 1 - W, G = dL/dC and RG = <R, G> where R is the colour
  // still to come (remaining_features).  Only the inner product with G is ever used:
  //   d(alpha) = sum_c (f_c T - R_c / (1 - alpha)) G_c = T <f, G> - <R, G> / (1 - alpha),
  // and R -= f w becomes RG -= w <f, G>, so one scalar replaces the three R channels.
  float G0 = 0.f, G1 = 0.f, G2 = 0.f, RG = 0.f;
  float T = 0.0f;
  if (in_bounds) {
    const int64_t p = (int64_t)pix_y * rp.width + pix_x;
    G0 = grad_image[p * 3 + 0]; G1 = grad_image[p * 3 + 1]; G2 = grad_image[p * 3 + 2];
    RG = image[p * 3 + 0] * G0 + image[p * 3 + 1] * G1 + image[p * 3 + 2] * G2;
    T = 1.0f;
  }

  // which output word this lane commits after the butterfly: value k < 7 -> grad_points[id][k],
  // k < 10 -> grad_feats[id][k - 7], then the two heuristics
  float* tgt = nullptr;
  const uint32_t* tgt_off = s_off[0];      // this lane's row-offset array
  if ((lane & 15) >= 12) {
    const int k = butterfly_slot(lane);
    if (k < 7) { if (grad_points) tgt = grad_points + k; }
    else if (k < 10) { if (grad_feats) { tgt = grad_feats + (k - 7); tgt_off = s_off[1]; } }
    else if (HEUR && k < 12) { if (heuristic) { tgt = heuristic + (k - 10); tgt_off = s_off[NOUT - 1]; } }
  }
  const bool b0 = lane & 1, b1 = lane & 2;

  const int start = ranges[tile_id * 2 + 0], end = ranges[tile_id * 2 + 1];
  const int t = threadIdx.x;

  Raw raw;
  int next_id = 0;
  const bool stager = t < BATCH;
  if (stager && start + t < end) raw = load_raw(points, feats, o2p[start + t]);
  if (stager && start + BATCH + t < end) next_id = o2p[start + BATCH + t];

  for (int begin = start; begin < end; begin += BATCH) {
    const int count = (end - begin) < BATCH ? (end - begin) : BATCH;
    // tile-wide early out once every pixel is saturated (backward.py:116)
    if (__syncthreads_and(T <= rp.one_minus_saturate)) break;
    if (stager && begin + t < end) {
      write_records(raw, rp.alpha_threshold, &s_rec[t * 3], &s_cull[t * 2]);
      const uint32_t id = (uint32_t)raw.id;
      s_off[0][t] = id * 7u;
      s_off[1][t] = id * 3u;
      if (HEUR) s_off[NOUT - 1][t] = id * 2u;
    }
    if (stager && begin + BATCH + t < end) raw = load_raw(points, feats, next_id);
    if (stager && begin + 2 * BATCH + t < end) next_id = o2p[begin + 2 * BATCH + t];
    __syncthreads();

    // wave-wide early out (backward.py:142)
    if (__ballot(T > rp.one_minus_saturate) == 0) continue;

    for (int r = 0; r < count; r += 64) {
      const int j = r + lane;
      bool hit = false;
      if (j < count) hit = patch_hit(s_cull[j * 2], s_cull[j * 2 + 1], rcx, rcy);
      unsigned long long m = __ballot(hit);
      while (m) {
        const int b = __builtin_ctzll(m);
        asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(b));
        const int ri = (r + b) * 3;
        const float4 q0 = s_rec[ri + 0], q1 = s_rec[ri + 1], q2 = s_rec[ri + 2];

        const float A = q0.z, B = q0.w, C = q1.x, D = q1.y, alpha_pt = q1.z;
        const float f0 = q1.w, f1 = q2.x, f2 = q2.y, isx = q2.z, isy = q2.w;
        const float dx = px - q0.x, dy = py - q0.y;
        const float X = dx * A + dy * B;
        const float Y = dx * C + dy * D;
        const float g = __builtin_amdgcn_exp2f((X * X + Y * Y) * EXP2_SCALE);
        const float a_raw = alpha_pt * g;
        const bool active = (a_raw > rp.alpha_threshold) && (T > rp.one_minus_saturate);

        if (__ballot(active) != 0) {
          const float a = min_f32(a_raw, rp.clamp_max_alpha);
          const float w = active ? a * T : 0.0f;
          const float inv = __builtin_amdgcn_rcpf(1.0f - a);
          const float fG = f0 * G0 + f1 * G1 + f2 * G2;
          RG -= w * fG;
          // d(alpha) = T <f, G> - <R, G> / (1 - alpha)  (backward.py:171-175), T before the update
          float ag = T * fG - RG * inv;
          ag = active ? ag : 0.0f;
          T -= w;
          const float aag = alpha_pt * ag;            // straight-through clamp (backward.py:158-163)

          // dp/dmean = g (X/sx axis + Y/sy perp(axis)); dp/daxis = g (X/sx (-d) + Y/sy perp(d));
          // dp/dsigma = g (X^2/sx, Y^2/sy)   (generic.py:321-336), all scaled by aag
          const float qX = aag * g * X, qY = aag * g * Y;
          const float u = qX * isx, wv = qY * isy;
          float v[16];
          v[0] = qX * A + qY * C;
          v[1] = qX * B + qY * D;
          v[2] = -(u * dx + wv * dy);
          v[3] = wv * dx - u * dy;
          v[4] = u * X;
          v[5] = wv * Y;
          v[6] = g * ag;
          v[7] = w * G0; v[8] = w * G1; v[9] = w * G2;
          if (HEUR) {
            v[10] = aag * aag;                         // backward.py:190-194
            v[11] = fabsf(v[0]) + fabsf(v[1]);
          } else {
            v[10] = 0.f; v[11] = 0.f;
          }
          v[12] = 0.f; v[13] = 0.f; v[14] = 0.f; v[15] = 0.f;

#if MS_ABLATE == 1   /* profiling only: no cross-lane reduction */
          const float total = v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7] + v[8] + v[9];
#else
          const float total = wave_reduce16(v, b0, b1);
#endif
#if MS_ABLATE == 2   /* profiling only: no atomic commit */
          if (tgt && total == 123.456f) {
#else
          if (tgt) {
#endif
            atomic_add_noret(tgt + (size_t)tgt_off[r + b], total);
          }
        }

      }
    }
  }
}

}  // namespace ms

using namespace ms;

#if MS_FWD_PHASES
// rows: device buffer of (waves of the launch) x 12 uint64 the next launches fill (NULL: stop recording)
extern "C" int ms_debug_fwd_phases(unsigned long long* rows, int unused) {
  (void)unused;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_phase_rows), &rows, sizeof(rows));
  return 0;
}
#endif

static FastParams make_fast_params(int w, int h, const ms_raster_config* cfg, int row_begin, int num_tiles) {
  FastParams rp{};
  rp.width = w; rp.height = h;
  rp.tiles_wide = (w + cfg->tile_size - 1) / cfg->tile_size;
  rp.tile_begin = row_begin * rp.tiles_wide;
  rp.clamp_max_alpha = (float)cfg->clamp_max_alpha;
  rp.alpha_threshold = (float)cfg->alpha_threshold;
  rp.one_minus_saturate = (float)(1.0 - cfg->saturate_threshold);
  rp.deterministic = 0;
  rp.num_tiles = num_tiles;
  rp.grad_broadcast = 0;
  return rp;
}

// Called from raster.hip's dispatch.  Returns true if the fast path handled the launch.
bool ms_raster_fwd_fast(const void* points, const void* feats, const int32_t* ranges, const int32_t* o2p,
                        int w, int h, const ms_raster_config* cfg, void* image, void* alpha, void* visibility,
                        int row_begin, int num_tiles, hipStream_t s, const float* splat_rows, const SplitScratch* split,
                        int32_t* long_run_word) {
  FastParams rp = make_fast_params(w, h, cfg, row_begin, num_tiles);
  rp.long_run_word = long_run_word;
  // long runs are cut when a scratch block is given; with visibility every segment is walked twice (a segment does not
  // know the transmittance in front of it before the composition pass has run)
  const bool cut = split != nullptr;
  if (cut) {
    rp.split_min_run = split->min_run;
    rp.split_items = split->items; rp.split_counts = split->counts; rp.split_state = split->state;
    (void)hipMemsetAsync(split->counts, 0, 4 * sizeof(int32_t), s);
    split_plan_kernel<<<dim3((unsigned)((num_tiles + 255) / 256)), dim3(256), 0, s>>>(
        ranges, rp.tile_begin, num_tiles, split->min_run, split->seg_len, (int)split->item_cap, (int)split->long_cap,
        split->counts, split->long_tiles, split->items);
  }
  const dim3 grid(xcd_grid<FWD_XCD_CHUNK>(rp.num_tiles, 1));
#define MS_GO3(TS, VIS, ROWS)                                                                                   \
  raster_fwd_f32x3_kernel<TS, VIS, ROWS><<<grid, dim3(TS * TS), 0, s>>>(                                        \
      ROWS ? splat_rows : (const float*)points, (const float*)feats, ranges, o2p, rp, (float*)image, (float*)alpha,  \
      VIS ? (float*)visibility : nullptr)
#define MS_SEG(TS, VIS, ROWS)                                                                                   \
  raster_fwd_f32x3_kernel<TS, VIS, ROWS, true><<<dim3((unsigned)split->item_cap), dim3(TS * TS), 0, s>>>(       \
      ROWS ? splat_rows : (const float*)points, (const float*)feats, ranges, o2p, rp, (float*)image, (float*)alpha, \
      VIS ? (float*)visibility : nullptr)
#define MS_GO(TS)                                                                                               \
  do {                                                                                                          \
    if (splat_rows) { if (visibility) MS_GO3(TS, true, true); else MS_GO3(TS, false, true); }                   \
    else { if (visibility) MS_GO3(TS, true, false); else MS_GO3(TS, false, false); }                            \
    if (cut) {                                                                                                  \
      if (splat_rows) MS_SEG(TS, false, true); else MS_SEG(TS, false, false);                                   \
      split_combine_kernel<TS><<<dim3((unsigned)split->long_cap), dim3(TS * TS), 0, s>>>(                       \
          rp, split->long_tiles, (float*)image, (float*)alpha);                                                 \
      if (visibility) { if (splat_rows) MS_SEG(TS, true, true); else MS_SEG(TS, true, false); }                 \
    }                                                                                                           \
  } while (0)
  switch (cfg->tile_size) {
    case 8: MS_GO(8); return true;
    case 16: MS_GO(16); return true;
    case 32: MS_GO(32); return true;
  }
#undef MS_GO
#undef MS_SEG
#undef MS_GO3
  return false;
}

bool ms_raster_bwd_fast(const void* points, const void* feats, const int32_t* ranges, const int32_t* o2p,
                        const void* image, const void* grad_image, int w, int h, const ms_raster_config* cfg,
                        void* gp, void* gf, void* heur, int row_begin, int num_tiles, hipStream_t s) {
  const FastParams rp = make_fast_params(w, h, cfg, row_begin, num_tiles);
  const dim3 grid(xcd_grid<BWD_XCD_CHUNK>(rp.num_tiles, 1));
  const bool hf = cfg->compute_point_heuristic && heur;
#define MS_GO(TS, HEUR) raster_bwd_f32x3_kernel<TS, HEUR><<<grid, dim3(TS * TS), 0, s>>>(                   \
      (const float*)points, (const float*)feats, ranges, o2p, (const float*)image, (const float*)grad_image, \
      rp, (float*)gp, (float*)gf, (float*)heur)
  switch (cfg->tile_size) {
    case 8: if (hf) MS_GO(8, true); else MS_GO(8, false); return true;
    case 16: if (hf) MS_GO(16, true); else MS_GO(16, false); return true;
    case 32: if (hf) MS_GO(32, true); else MS_GO(32, false); return true;
  }
#undef MS_GO
  return false;
}
