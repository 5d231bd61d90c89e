// raster_pairs.hip — backward alpha-composite kernel with ACTIVE-PAIR COMPACTION (float32, RGB, plain
// pdf, blending, no point heuristics).
//
// Motivation (config D, raster_fast.hip): a (patch, splat) hit has on average ~10 contributing pixels
// out of the 64 lanes that evaluate it, and ~2/3 of the backward's VALU work (gradient terms + the
// 64-lane reduction) is spent on lanes that contribute nothing.  Only the transmittance recurrence is
// inherently "one lane per pixel, splats in depth order"; the gradient terms are independent per
// (pixel, splat) pair.  So the work is split:
//
//   phase A  (lane = pixel, one splat at a time, as before): evaluate g, alpha, the contribution gate,
//            update T and <R,G>, and form the per-pair scalars  q = alpha_pt * dalpha * g,
//            dalpha_pt = g * dalpha,  w * G_c.  The ACTIVE lanes append a 24-byte pair record to a
//            per-wave LDS queue, compacted with the ballot rank (v_mbcnt), in rows of 16 entries: a hit
//            with n active pixels occupies ceil(n / 16) rows, each row belongs to one splat.
//   phase B  (runs whenever 4 rows are queued; lane = queue entry, the four DPP rows of the wave work on
//            four different (splat, row) groups): rebuild dx, dy, X, Y from the pixel id and the splat
//            record, expand q into the six geometric gradient terms, reduce the 10 values over the 16
//            lanes of the row (two halving quad stages + row_shr:4/8) and commit them with global
//            atomics from the row's last four lanes.
//
// The gradient terms and the reduction are thereby evaluated on rows that are ~50 % full instead of
// waves that are ~16 % full, while the number of atomic commits stays ~1.3 per hit (one per queue row).
// Semantics are those of raster_fast.hip / rasterizer/backward.py:97-224.
#include "raster_common.h"

#ifndef MS_ABLATE
#define MS_ABLATE 0      // profiling only: 1 = no atomic commit, 2 = no phase B, 3 = no queue writes either
#endif

namespace ms {

constexpr int PQ_ROWS = 8;                 // queue capacity in rows (ring): <= 3 pending + <= 4 from one hit
constexpr int PQ_ENTRIES = PQ_ROWS * 16;

struct PairQueue {
  float4* rec;        // [PQ_ENTRIES] {pixel id bits, q, dalpha_pt, w*G0}
  float2* rec2;       // [PQ_ENTRIES] {w*G1, w*G2}
  int* row_splat;     // [PQ_ROWS] staged-splat index of each row
};

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

// Phase B on the 4 queue rows starting at `head` (ring); rows >= valid_rows are padding.
__device__ __forceinline__ void process_rows(const PairQueue& q, int head, int valid_rows, int lane,
                                             const float4* s_rec, const int32_t* s_id, float patch_x,
                                             float patch_y, float* const (&tgt)[3], const unsigned (&tgt_stride)[3]) {
  const int row = lane >> 4, li = lane & 15;
  const int qrow = (head + row) & (PQ_ROWS - 1);
  const int e = qrow * 16 + li;
  const float4 a = q.rec[e];
  const float2 b = q.rec2[e];
  const int sidx = q.row_splat[qrow];
  const float4 q0 = s_rec[sidx * 3 + 0], q1 = s_rec[sidx * 3 + 1], q2 = s_rec[sidx * 3 + 2];
  // leave the entry clean for its next use (padding lanes of a future row must read zeros)
  q.rec[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  q.rec2[e] = make_float2(0.f, 0.f);

  const int pix = __float_as_int(a.x) & 63;
  const float px = patch_x + (float)(pix & 7) + 0.5f, py = patch_y + (float)(pix >> 3) + 0.5f;
  const float A = q0.z, B = q0.w, C = q1.x, D = q1.y, isx = q2.z, isy = q2.w;
  const float dx = px - q0.x, dy = py - q0.y;
  const float X = dx * A + dy * B;
  const float Y = dx * C + dy * D;
  const float qq = a.y;
  const float qX = qq * X, qY = qq * Y;
  const float u = qX * isx, wv = qY * isy;
  float v[12];
  v[0] = qX * A + qY * C;              // d mean      (generic.py:321-336, scaled by alpha_pt * dalpha)
  v[1] = qX * B + qY * D;
  v[2] = -(u * dx + wv * dy);          // d axis
  v[3] = wv * dx - u * dy;
  v[4] = u * X;                        // d sigma
  v[5] = wv * Y;
  v[6] = a.z;                          // d alpha_pt = g * dalpha
  v[7] = a.w; v[8] = b.x; v[9] = b.y;  // d colour = w * G
  v[10] = 0.f; v[11] = 0.f;

  float r[3];
  row_reduce12(v, lane & 1, lane & 2, r);
  if (MS_ABLATE == 1) { if (r[0] + r[1] + r[2] == 123.456f) q.rec[e].x = 1.f; return; }
  // gather the 10 row totals into lanes 4, 5, 8..15 of the row (see the tgt set-up): one atomic
  // instruction per four rows, <= 2 cache lines per row
  const float c1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r[1]), 0x104, 0xf, 0xf, true));  // row_shl:4
  const float c2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, r[2]), 0x108, 0xf, 0xf, true));  // row_shl:8
  const float c = li >= 12 ? r[0] : (li >= 8 ? c1 : c2);
  bool commit_ok = true;
  if (MS_ABLATE == 4) commit_ok = row == 0 || q.row_splat[(head + row - 1) & (PQ_ROWS - 1)] != sidx;
  if (MS_ABLATE == 5) commit_ok = row == 0;
  if (tgt[0] && row < valid_rows && commit_ok) {
    typedef __attribute__((address_space(1))) float gfloat;
    const unsigned id = (unsigned)s_id[sidx];
    __hip_atomic_fetch_add((gfloat*)(tgt[0] + (size_t)(id * tgt_stride[0])), c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int TS>
__global__ void __launch_bounds__(TS * TS)
raster_bwd_pairs_kernel(const float* __restrict__ points, const float* __restrict__ feats,
                        const int32_t* __restrict__ ranges, const int32_t* __restrict__ o2p,
                        const float* __restrict__ image, const float* __restrict__ grad_image,
                        FastParams rp, float* __restrict__ grad_points, float* __restrict__ grad_feats) {
  using G = TileGeom<TS>;
  constexpr int BATCH = G::BATCH;
  constexpr int WAVES = G::THREADS / 64;
  __shared__ float4 s_rec[BATCH * 3];
  __shared__ float4 s_cull[BATCH * 2];
  __shared__ int32_t s_id[BATCH];
  __shared__ float4 s_qrec[WAVES * PQ_ENTRIES];
  __shared__ float2 s_qrec2[WAVES * PQ_ENTRIES];
  __shared__ int s_qrow[WAVES * PQ_ROWS];

  const int tile_id = rp.tile_begin + blockIdx.x;
  const int tile_u = tile_id % rp.tiles_wide, tile_v = tile_id / rp.tiles_wide;
  const int wave = threadIdx.x >> 6, lane = lane_id();
  const int patch_x = tile_u * TS + (wave % G::WAVES_WIDE) * 8;
  const int patch_y = tile_v * TS + (wave / G::WAVES_WIDE) * 8;
  const int pix_x = patch_x + (lane & 7), pix_y = patch_y + (lane >> 3);
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

  // phase B commit targets: row total k -> grad_points[id][k] (k < 7) / grad_feats[id][k - 7] (k < 10)
  float* tgt[3] = {nullptr, nullptr, nullptr};
  unsigned tgt_stride[3] = {0, 0, 0};
  {
    const int li = lane & 15;
    const int k = li >= 12 ? li - 12 : (li >= 8 ? li - 4 : li + 4);     // lanes 12..15: 0..3, 8..11: 4..7, 4..5: 8..9
    if (li >= 4 && k < 10) {
      if (k < 7) { if (grad_points) { tgt[0] = grad_points + k; tgt_stride[0] = 7; } }
      else if (grad_feats) { tgt[0] = grad_feats + (k - 7); tgt_stride[0] = 3; }
    }
  }

  PairQueue q;
  q.rec = s_qrec + wave * PQ_ENTRIES;
  q.rec2 = s_qrec2 + wave * PQ_ENTRIES;
  q.row_splat = s_qrow + wave * PQ_ROWS;
  for (int e = lane; e < PQ_ENTRIES; e += 64) {
    q.rec[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    q.rec2[e] = make_float2(0.f, 0.f);
  }
  if (lane < PQ_ROWS) q.row_splat[lane] = 0;
  int q_head = 0, q_rows = 0;     // wave-uniform ring state

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
      s_id[t] = raw.id;
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
      const int rec_index = j * 3;
      while (m) {
        const int b = __builtin_ctzll(m);
        m &= ~(1ull << b);
        const int ri = __builtin_amdgcn_readlane(rec_index, b);
        const float4 q0 = s_rec[ri + 0], q1 = s_rec[ri + 1], q2 = s_rec[ri + 2];

        // ---- phase A: the sequential part ------------------------------------------------------------
        const float dx = px - q0.x, dy = py - q0.y;
        const float X = dx * q0.z + dy * q0.w;
        const float Y = dx * q1.x + dy * q1.y;
        const float g = __builtin_amdgcn_exp2f((X * X + Y * Y) * EXP2_SCALE);
        const float alpha_pt = q1.z;
        const float a_raw = alpha_pt * g;
        const bool active = (a_raw > rp.alpha_threshold) && (T > rp.one_minus_saturate);
        const unsigned long long am = __ballot(active);
        if (am == 0) continue;

        const float a = min_f32(a_raw, rp.clamp_max_alpha);
        const float w = a * T;                                      // only used by active lanes
        const float inv = __builtin_amdgcn_rcpf(1.0f - a);
        const float fG = q1.w * G0 + q2.x * G1 + q2.y * G2;
        // d(alpha) = T <f, G> - <R, G> / (1 - alpha)  (backward.py:171-175), T before the update
        const float RGn = RG - w * fG;
        const float ag = T * fG - RGn * inv;
        if (active) { RG = RGn; T -= w; }

        // append the active pairs, compacted, in rows of 16 entries
        const int n_act = __popcll(am);
        const int rows = (n_act + 15) >> 4;
        const int tail = (q_head + q_rows) & (PQ_ROWS - 1);
        if (MS_ABLATE == 3) { if (active && w * G1 + w * G2 + g * ag == 123.456f) q.rec[0].x = 1.f; continue; }
        if (active) {
          const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0u));
          const int e = (tail * 16 + rank) & (PQ_ENTRIES - 1);
          const float gag = g * ag;
          q.rec[e] = make_float4(__int_as_float(lane), alpha_pt * gag, gag, w * G0);
          q.rec2[e] = make_float2(w * G1, w * G2);
        }
        if (lane < rows) q.row_splat[(tail + lane) & (PQ_ROWS - 1)] = ri / 3;
        q_rows += rows;

        // ---- phase B whenever four rows are queued ----------------------------------------------------
        while (q_rows >= 4) {
          if (MS_ABLATE == 2) { q_head = (q_head + 4) & (PQ_ROWS - 1); q_rows -= 4; continue; }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          process_rows(q, q_head, 4, lane, s_rec, s_id, (float)patch_x, (float)patch_y, tgt, tgt_stride);
          q_head = (q_head + 4) & (PQ_ROWS - 1);
          q_rows -= 4;
        }
      }
    }

    // the queue refers to this batch's staged records: drain it before they are replaced
    if (q_rows > 0 && MS_ABLATE < 2) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      process_rows(q, q_head, q_rows, lane, s_rec, s_id, (float)patch_x, (float)patch_y, tgt, tgt_stride);
      q_head = (q_head + q_rows) & (PQ_ROWS - 1);
      q_rows = 0;
    }
  }
}

}  // namespace ms

using namespace ms;

bool ms_raster_bwd_pairs(const void* points, const void* feats, const int32_t* ranges, const int32_t* o2p,
                         const void* image, const void* grad_image, int w, int h, const ms_raster_config* cfg,
                         void* gp, void* gf, int row_begin, int num_tiles, hipStream_t s) {
  FastParams rp;
  rp.width = w; rp.height = h;
  rp.tiles_wide = (w + cfg->tile_size - 1) / cfg->tile_size;
  rp.tile_begin = row_begin * rp.tiles_wide;
  rp.clamp_max_alpha = (float)cfg->clamp_max_alpha;
  rp.alpha_threshold = (float)cfg->alpha_threshold;
  rp.one_minus_saturate = (float)(1.0 - cfg->saturate_threshold);
  const dim3 grid((unsigned)num_tiles);
#define MS_GO(TS) raster_bwd_pairs_kernel<TS><<<grid, dim3(TS * TS), 0, s>>>(                              \
      (const float*)points, (const float*)feats, ranges, o2p, (const float*)image, (const float*)grad_image, \
      rp, (float*)gp, (float*)gf)
  switch (cfg->tile_size) {
    case 8: MS_GO(8); return true;
    case 16: MS_GO(16); return true;
    case 32: MS_GO(32); return true;
  }
#undef MS_GO
  return false;
}
