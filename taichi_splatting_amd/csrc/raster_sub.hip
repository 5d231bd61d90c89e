// raster_sub.hip — product-path raster kernels, sub-patch edition (float32, RGB, plain pdf, blending).
//
// Same tile / wave / pixel ownership as raster_fast.hip (one workgroup per tile, one wave64 per 8x8
// pixel patch, one lane per pixel) but the wave no longer walks ONE splat at a time.  With the ~3 px
// radius splats of the headline workload a splat touches ~10 of the 64 pixels of a patch it hits, so
// 5/6 of the lanes of a hit were idle.  Here the four 16-lane DPP rows of the wave own the four 4x4
// sub-patches of the patch and each row walks ITS OWN depth-ordered hit list:
//
//   phase 1 (per LDS batch, lane = staged splat): test the splat against the 4 sub-patch rectangles
//           (extent + oriented-box test, conservative), ballot, and compact the hits of every
//           sub-patch into a per-row index list in LDS (v_mbcnt ranks keep the depth order);
//   phase 2 (lane = pixel): iteration k blends, in every row, the k-th splat of that row's list —
//           four different splats per wave instruction (record reads are 4-address LDS broadcasts).
//           Rows whose list is exhausted read a dummy record with alpha = 0.
//
// Per-pixel order is untouched (each row consumes its list front to back), so results equal the
// one-splat-at-a-time loop; the trip count per batch is max(row list length) ~ 0.55x the patch's hit
// count on config D.  Backward: the gradient sum of a (sub-patch, splat) pair is a 16-lane reduction
// — two halving quad stages + row_shr:4/8, no cross-row stage — and the 12 totals of each row sit
// in its last four lanes, committed by three global_atomic_add_f32 instructions for all four rows.
#include "raster_common.h"

namespace ms {

template <int TS> struct SubGeom {
  static constexpr int THREADS = TS * TS;
  static constexpr int WAVES = THREADS / 64;
  static constexpr int BATCH = THREADS < 256 ? THREADS : 256;
  static constexpr int WAVES_WIDE = TS / 8;
};

// number of set bits of the wave-wide mask below this lane
__device__ __forceinline__ int rank_below(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Conservative test of one splat against the four 4x4 sub-patches of a patch at once.  (rcx, rcy) is
// the patch centre; sub-patch q has its pixel-centre rectangle centred at (rcx -+ 2, rcy -+ 2), half
// size 1.5.  Same two criteria as rect_hit (axis-aligned extents, oriented-box axes of the ellipse) with
// the shared terms computed once: ~45 VALU for the four answers instead of four independent tests.
__device__ __forceinline__ void subpatch_hits(const float4 c0, const float4 c1, float rcx, float rcy, bool (&h)[4]) {
  const float dx = rcx - c0.x, dy = rcy - c0.y;
  const float lx = c0.z + 1.5f, ly = c0.w + 1.5f;
  const bool x0 = fabsf(dx - 2.0f) <= lx, x1 = fabsf(dx + 2.0f) <= lx;
  const bool y0 = fabsf(dy - 2.0f) <= ly, y1 = fabsf(dy + 2.0f) <= ly;
  // projections of the four rectangle centres on the two ellipse axes
  const float p1 = c1.x * dx + c1.y * dy, a1 = 2.0f * c1.x, b1 = 2.0f * c1.y;
  const float p2 = c1.z * dx + c1.w * dy, a2 = 2.0f * c1.z, b2 = 2.0f * c1.w;
  const float t1 = 1.002f + (fabsf(c1.x) + fabsf(c1.y)) * 1.5f;
  const float t2 = 1.002f + (fabsf(c1.z) + fabsf(c1.w)) * 1.5f;
  const float p1m = p1 - b1, p1p = p1 + b1, p2m = p2 - b2, p2p = p2 + b2;
  h[0] = x0 && y0 && fabsf(p1m - a1) <= t1 && fabsf(p2m - a2) <= t2;   // (-2, -2)
  h[1] = x1 && y0 && fabsf(p1m + a1) <= t1 && fabsf(p2m + a2) <= t2;   // (+2, -2)
  h[2] = x0 && y1 && fabsf(p1p - a1) <= t1 && fabsf(p2p - a2) <= t2;   // (-2, +2)
  h[3] = x1 && y1 && fabsf(p1p + a1) <= t1 && fabsf(p2p + a2) <= t2;   // (+2, +2)
}

// Phase 1: build the four per-row hit lists of this wave for the staged batch (indices < 256 fit a
// byte).  Returns the list lengths (wave-uniform).
template <int BATCH>
__device__ __forceinline__ void build_lists(const float4* s_cull, int count, int lane, float rcx, float rcy,
                                            unsigned char* list, int (&cnt)[4]) {
  cnt[0] = cnt[1] = cnt[2] = cnt[3] = 0;
  for (int r0 = 0; r0 < count; r0 += 64) {
    const int j = r0 + lane;
    bool h[4] = {false, false, false, false};
    if (j < count) subpatch_hits(s_cull[j * 2], s_cull[j * 2 + 1], rcx, rcy, h);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const unsigned long long m = __ballot(h[q]);
      if (h[q]) list[q * BATCH + cnt[q] + rank_below(m)] = (unsigned char)j;
      cnt[q] += __popcll(m);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int TS>
__global__ void __launch_bounds__(TS * TS)
raster_fwd_sub_kernel(const float* __restrict__ points, const float* __restrict__ feats,
                      const int32_t* __restrict__ ranges, const int32_t* __restrict__ o2p,
                      FastParams rp, float* __restrict__ image, float* __restrict__ image_alpha) {
  using G = SubGeom<TS>;
  constexpr int BATCH = G::BATCH;
  __shared__ float4 s_rec[(BATCH + 1) * 3];            // + the dummy record (alpha = 0)
  __shared__ float4 s_cull[BATCH * 2];
  __shared__ unsigned char s_list[G::WAVES * 4 * BATCH];

  const int tile_id = rp.tile_begin + blockIdx.x;
  const int tile_u = tile_id % rp.tiles_wide, tile_v = tile_id / rp.tiles_wide;
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int row = lane >> 4, li = lane & 15;
  const int patch_x = tile_u * TS + (wave % G::WAVES_WIDE) * 8;
  const int patch_y = tile_v * TS + (wave / G::WAVES_WIDE) * 8;
  const int pix_x = patch_x + (row & 1) * 4 + (li & 3), pix_y = patch_y + (row >> 1) * 4 + (li >> 2);
  const float px = (float)pix_x + 0.5f, py = (float)pix_y + 0.5f;
  const float rcx = (float)patch_x + 4.0f, rcy = (float)patch_y + 4.0f;
  const bool in_bounds = pix_x < rp.width && pix_y < rp.height;

  float c0 = 0.f, c1 = 0.f, c2 = 0.f;
  float T = in_bounds ? 1.0f : 0.0f;     // transmittance = 1 - accumulated weight

  const int start = ranges[tile_id * 2 + 0], end = ranges[tile_id * 2 + 1];
  const int t = threadIdx.x;
  if (t == 0) {
    s_rec[BATCH * 3 + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
    s_rec[BATCH * 3 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    s_rec[BATCH * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  Raw raw;
  int next_id = 0;
  const bool stager = t < BATCH;
  if (stager && start + t < end) raw = load_raw(points, feats, o2p[start + t]);
  if (stager && start + BATCH + t < end) next_id = o2p[start + BATCH + t];

  unsigned char* list = s_list + wave * 4 * BATCH;
  const unsigned char* my_list = list + row * BATCH;

  for (int begin = start; begin < end; begin += BATCH) {
    const int count = (end - begin) < BATCH ? (end - begin) : BATCH;
    __syncthreads();                       // previous batch fully consumed
    if (stager && begin + t < end) write_records(raw, rp.alpha_threshold, &s_rec[t * 3], &s_cull[t * 2]);
    if (stager && begin + BATCH + t < end) raw = load_raw(points, feats, next_id);
    if (stager && begin + 2 * BATCH + t < end) next_id = o2p[begin + 2 * BATCH + t];
    __syncthreads();

    int cnt[4];
    build_lists<BATCH>(s_cull, count, lane, rcx, rcy, list, cnt);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int my_cnt = row == 0 ? cnt[0] : (row == 1 ? cnt[1] : (row == 2 ? cnt[2] : cnt[3]));
    const int n_iter = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
    if (n_iter == 0) continue;

    // two-deep software pipeline over the row's list: record of iteration k, index of k + 1 in flight
    int idx = 0 < my_cnt ? (int)my_list[0] : BATCH;
    float4 q0 = s_rec[idx * 3 + 0], q1 = s_rec[idx * 3 + 1], q2 = s_rec[idx * 3 + 2];
    int idn = 1 < my_cnt ? (int)my_list[1] : BATCH;
    for (int k = 0; k < n_iter; ++k) {
      const float4 n0 = s_rec[idn * 3 + 0], n1 = s_rec[idn * 3 + 1], n2 = s_rec[idn * 3 + 2];
      const int k2 = k + 2 < BATCH ? k + 2 : BATCH - 1;
      const int idn2 = k + 2 < my_cnt ? (int)my_list[k2] : BATCH;

      const float dx = px - q0.x, dy = py - q0.y;
      const float X = dx * q0.z + dy * q0.w;
      const float Y = dx * q1.x + dy * q1.y;
      const float g = __builtin_amdgcn_exp2f((X * X + Y * Y) * EXP2_SCALE);
      const float a = min_f32(q1.z * g, rp.clamp_max_alpha);
      const float w = a > rp.alpha_threshold ? a * T : 0.0f;
      T -= w;
      c0 += q1.w * w; c1 += q2.x * w; c2 += q2.y * w;

      q0 = n0; q1 = n1; q2 = n2;
      idn = idn2;
    }
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

// 16-lane (DPP row) halving butterfly: v[0..11] -> r[j] holds, in the lanes with (lane & 15) >= 12, the
// row total of value 4 j + (lane & 3).
__device__ __forceinline__ void row_reduce12(const float (&v)[12], bool b0, bool b1, float (&r)[3]) {
  float r1[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const float keep = b0 ? v[2 * i + 1] : v[2 * i];
    const float send = b0 ? v[2 * i] : v[2 * i + 1];
    r1[i] = add_dpp<0xB1>(keep, send);                      // quad_perm:[1,0,3,2]
  }
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const float keep = b1 ? r1[2 * j + 1] : r1[2 * j];
    const float send = b1 ? r1[2 * j] : r1[2 * j + 1];
    float x = add_dpp<0x4E>(keep, send);                    // quad_perm:[2,3,0,1]
    x = add_dpp<0x114>(x, x);                               // row_shr:4
    x = add_dpp<0x118>(x, x);                               // row_shr:8
    r[j] = x;
  }
}

// Commit the tile-level sums of the 64 staged splats a wave owns (entries base .. base + 63).  Lanes
// map to CONSECUTIVE gradient words (lane -> (splat, value)), so the lanes of one atomic instruction
// fall into a handful of cache lines: the L2 atomic units see ~2 line operations per tile overlap
// instead of one per value (measured: one-lane-per-splat flushing made the L2 atomics the bottleneck,
// 3.1 ms of a 5.8 ms launch).
template <bool HEUR>
__device__ __forceinline__ void flush_wave(const float* s_acc, const int32_t* s_id, int base, int lane,
                                           float* grad_points, float* grad_feats, float* heuristic) {
  constexpr int NV = HEUR ? 12 : 10;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = i * 64 + lane;
    const int sl = e / NV, k = e - sl * NV;
    const float v = s_acc[(base + sl) * 12 + k];
    if (v != 0.f) {
      const size_t id = (size_t)(unsigned)s_id[base + sl];
      float* dst = k < 7 ? (grad_points ? grad_points + id * 7 + k : nullptr)
                         : (k < 10 ? (grad_feats ? grad_feats + id * 3 + (k - 7) : nullptr)
                                   : (heuristic ? heuristic + id * 2 + (k - 10) : nullptr));
      if (dst) atomic_add_noret(dst, v);
    }
  }
}

template <int TS, bool HEUR>
__global__ void __launch_bounds__(TS * TS)
raster_bwd_sub_kernel(const float* __restrict__ points, const float* __restrict__ feats,
                      const int32_t* __restrict__ ranges, const int32_t* __restrict__ o2p,
                      const float* __restrict__ image, const float* __restrict__ grad_image,
                      FastParams rp, float* __restrict__ grad_points, float* __restrict__ grad_feats,
                      float* __restrict__ heuristic) {
  using G = SubGeom<TS>;
  constexpr int BATCH = G::BATCH;
  __shared__ float4 s_rec[(BATCH + 1) * 3];
  __shared__ float4 s_cull[BATCH * 2];
  __shared__ int32_t s_id[BATCH + 1];
  __shared__ unsigned char s_list[G::WAVES * 4 * BATCH];
  __shared__ float s_acc[BATCH * 12];    // per staged splat: the 12 gradient sums over the whole TILE

  const int tile_id = rp.tile_begin + blockIdx.x;
  const int tile_u = tile_id % rp.tiles_wide, tile_v = tile_id / rp.tiles_wide;
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int row = lane >> 4, li = lane & 15;
  const int patch_x = tile_u * TS + (wave % G::WAVES_WIDE) * 8;
  const int patch_y = tile_v * TS + (wave / G::WAVES_WIDE) * 8;
  const int pix_x = patch_x + (row & 1) * 4 + (li & 3), pix_y = patch_y + (row >> 1) * 4 + (li >> 2);
  const float px = (float)pix_x + 0.5f, py = (float)pix_y + 0.5f;
  const float rcx = (float)patch_x + 4.0f, rcy = (float)patch_y + 4.0f;
  const bool in_bounds = pix_x < rp.width && pix_y < rp.height;

  // per-pixel state (backward.py:97-110): T = 1 - W, G = dL/dC and RG = <R, G> (see raster_fast.hip)
  float G0 = 0.f, G1 = 0.f, G2 = 0.f, RG = 0.f;
  float T = 0.0f;
  if (in_bounds) {
    const int64_t p = (int64_t)pix_y * rp.width + pix_x;
    G0 = grad_image[p * 3 + 0]; G1 = grad_image[p * 3 + 1]; G2 = grad_image[p * 3 + 2];
    RG = image[p * 3 + 0] * G0 + image[p * 3 + 1] * G1 + image[p * 3 + 2] * G2;
    T = 1.0f;
  }

  // after row_reduce12 the lanes with (lane & 15) >= 12 hold the row totals of values 4 j + (lane & 3),
  // j = 0..2; they are accumulated per staged splat in LDS (ds_add_f32) and flushed to HBM once per
  // batch by the staging thread: 10 global atomics per TILE overlap instead of 10 per (sub-patch, splat)
  const bool b0 = lane & 1, b1 = lane & 2;

  const int start = ranges[tile_id * 2 + 0], end = ranges[tile_id * 2 + 1];
  const int t = threadIdx.x;
  if (t == 0) {
    s_rec[BATCH * 3 + 0] = make_float4(0.f, 0.f, 0.f, 0.f);
    s_rec[BATCH * 3 + 1] = make_float4(0.f, 0.f, 0.f, 0.f);
    s_rec[BATCH * 3 + 2] = make_float4(0.f, 0.f, 0.f, 0.f);
    s_id[BATCH] = 0;
  }

  Raw raw;
  int next_id = 0;
  const bool stager = t < BATCH;
  if (stager && start + t < end) raw = load_raw(points, feats, o2p[start + t]);
  if (stager && start + BATCH + t < end) next_id = o2p[start + BATCH + t];

  unsigned char* list = s_list + wave * 4 * BATCH;
  const unsigned char* my_list = list + row * BATCH;

  for (int begin = start; begin < end; begin += BATCH) {
    const int count = (end - begin) < BATCH ? (end - begin) : BATCH;
    // tile-wide early out once every pixel is saturated (backward.py:116)
    if (__syncthreads_and(T <= rp.one_minus_saturate)) break;
    if (stager) {
      if (begin > start) {
        flush_wave<HEUR>(s_acc, s_id, wave * 64, lane, grad_points, grad_feats, heuristic);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();     // the wave's entries are re-zeroed / re-staged below
      }
      if (begin + t < end) {
        write_records(raw, rp.alpha_threshold, &s_rec[t * 3], &s_cull[t * 2]);
        s_id[t] = raw.id;
      }
      float4* z = reinterpret_cast<float4*>(&s_acc[t * 12]);
      z[0] = z[1] = z[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (stager && begin + BATCH + t < end) raw = load_raw(points, feats, next_id);
    if (stager && begin + 2 * BATCH + t < end) next_id = o2p[begin + 2 * BATCH + t];
    __syncthreads();

    // wave-wide early out (backward.py:142)
    if (__ballot(T > rp.one_minus_saturate) == 0) continue;

    int cnt[4];
    build_lists<BATCH>(s_cull, count, lane, rcx, rcy, list, cnt);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    const int my_cnt = row == 0 ? cnt[0] : (row == 1 ? cnt[1] : (row == 2 ? cnt[2] : cnt[3]));
    const int n_iter = max(max(cnt[0], cnt[1]), max(cnt[2], cnt[3]));
    if (n_iter == 0) continue;

    int idx = 0 < my_cnt ? (int)my_list[0] : BATCH;
    float4 q0 = s_rec[idx * 3 + 0], q1 = s_rec[idx * 3 + 1], q2 = s_rec[idx * 3 + 2];
    int idn = 1 < my_cnt ? (int)my_list[1] : BATCH;
    for (int k = 0; k < n_iter; ++k) {
      const float4 n0 = s_rec[idn * 3 + 0], n1 = s_rec[idn * 3 + 1], n2 = s_rec[idn * 3 + 2];
      const int k2 = k + 2 < BATCH ? k + 2 : BATCH - 1;
      const int idn2 = k + 2 < my_cnt ? (int)my_list[k2] : BATCH;

      const float A = q0.z, B = q0.w, C = q1.x, D = q1.y, alpha_pt = q1.z;
      const float f0 = q1.w, f1 = q2.x, f2 = q2.y, isx = q2.z, isy = q2.w;
      const float dx = px - q0.x, dy = py - q0.y;
      const float X = dx * A + dy * B;
      const float Y = dx * C + dy * D;
      const float g = __builtin_amdgcn_exp2f((X * X + Y * Y) * EXP2_SCALE);
      const float a_raw = alpha_pt * g;
      const bool active = (a_raw > rp.alpha_threshold) && (T > rp.one_minus_saturate);
      const unsigned long long am = __ballot(active);

      if (am != 0) {
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

        const float qX = aag * g * X, qY = aag * g * Y;
        const float u = qX * isx, wv = qY * isy;
        float v[12];
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

        float r[3];
        row_reduce12(v, b0, b1, r);
        // rows without an active pixel (or walking the dummy record) commit nothing
        const bool row_active = ((unsigned)(am >> (row * 16)) & 0xffffu) != 0;
        if (row_active && li >= 12) {
          float* acc = &s_acc[idx * 12 + (lane & 3)];
#pragma unroll
          for (int j = 0; j < (HEUR ? 3 : 3); ++j)
            __hip_atomic_fetch_add(acc + 4 * j, r[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }

      q0 = n0; q1 = n1; q2 = n2;
      idx = idn; idn = idn2;
    }
  }
  // flush the last staged batch
  __syncthreads();
  if (stager && end > start) flush_wave<HEUR>(s_acc, s_id, wave * 64, lane, grad_points, grad_feats, heuristic);
}

}  // namespace ms

using namespace ms;

static FastParams make_sub_params(int w, int h, const ms_raster_config* cfg, int row_begin) {
  FastParams rp;
  rp.width = w; rp.height = h;
  rp.tiles_wide = (w + cfg->tile_size - 1) / cfg->tile_size;
  rp.tile_begin = row_begin * rp.tiles_wide;
  rp.clamp_max_alpha = (float)cfg->clamp_max_alpha;
  rp.alpha_threshold = (float)cfg->alpha_threshold;
  rp.one_minus_saturate = (float)(1.0 - cfg->saturate_threshold);
  return rp;
}

bool ms_raster_fwd_sub(const void* points, const void* feats, const int32_t* ranges, const int32_t* o2p,
                       int w, int h, const ms_raster_config* cfg, void* image, void* alpha, int row_begin,
                       int num_tiles, hipStream_t s) {
  const FastParams rp = make_sub_params(w, h, cfg, row_begin);
  const dim3 grid((unsigned)num_tiles);
#define MS_GO(TS) raster_fwd_sub_kernel<TS><<<grid, dim3(TS * TS), 0, s>>>(                                 \
      (const float*)points, (const float*)feats, ranges, o2p, rp, (float*)image, (float*)alpha)
  switch (cfg->tile_size) {
    case 8: MS_GO(8); return true;
    case 16: MS_GO(16); return true;
    case 32: MS_GO(32); return true;
  }
#undef MS_GO
  return false;
}

bool ms_raster_bwd_sub(const void* points, const void* feats, const int32_t* ranges, const int32_t* o2p,
                       const void* image, const void* grad_image, int w, int h, const ms_raster_config* cfg,
                       void* gp, void* gf, void* heur, int row_begin, int num_tiles, hipStream_t s) {
  const FastParams rp = make_sub_params(w, h, cfg, row_begin);
  const dim3 grid((unsigned)num_tiles);
  const bool hf = cfg->compute_point_heuristic && heur;
#define MS_GO(TS, HEUR) raster_bwd_sub_kernel<TS, HEUR><<<grid, dim3(TS * TS), 0, s>>>(                     \
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
