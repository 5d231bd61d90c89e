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
// -DMS_SCAN_PHASES=1: every wave adds up the shader cycles (s_memtime) it spends in each phase of the kernel
// (g_scan_phase, read with ms_debug_scan_phases; tools/phase_split.py) — where a wave's wall time goes, parked or not
#ifndef MS_SCAN_PHASES
#define MS_SCAN_PHASES 0
#endif
// Round 5: the phases of a wave that are NOT the blend are latency bound (tools/rbench.py with the -DMS_SCAN_PHASES
// build: 52 % of a wave's cycles on config D), and every one of them parks the wave while its SIMD could blend.
//   staging        the gathered splat data is turned into LDS records in REGISTERS before the pass's commit and written
//                  after the barrier: consuming a loaded register after the commit's global atomics costs
//                  s_waitcnt vmcnt(0), i.e. the round trip of every atomic (loads and atomics share the counter);
//                  one barrier instead of the three of __syncthreads_and at the top of a batch
//   commit         point ids of the patch list resolved once per pass (into the dead sub-patch lists), so that a commit
//                  step is one LDS round trip instead of three dependent ones
//   cull           both cull levels issue all their LDS reads before the first test
#ifndef MS_COMMIT_ABLATE
#define MS_COMMIT_ABLATE 0
#endif
// wave priority: 1 = the blend phase runs at raised priority (s_setprio 2), 2 = everything BUT the blend does, 0 = off.
// Same box, config D: off 1.340-1.345 ms, blend raised 1.334, the rest raised 1.376 — the issue-bound phase should not
// lose slots to waves that are about to park on a load anyway.
#ifndef MS_PRIO_MODE
#define MS_PRIO_MODE 1
#endif
#if MS_PRIO_MODE == 1
#define MS_PRIO_BLEND(on) __builtin_amdgcn_s_setprio((on) ? 2 : 0)
#elif MS_PRIO_MODE == 2
#define MS_PRIO_BLEND(on) __builtin_amdgcn_s_setprio((on) ? 0 : 2)
#else
#define MS_PRIO_BLEND(on) do {} while (0)
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
#if MS_SCAN_PHASES
// 0 barrier at the top of a batch   1 staging (record transform + next gathers)   2 barrier after staging
// 3 cull (both levels)   4 chunk prologue (list -> patch list -> record reads landed, set-up)   5 blend (16 steps)
// 6 chunk epilogue (read-add-write of the sums)   7 commit (global atomics)   8 kernel start .. first batch
// 9 whole wave   10 chunks   11 waves
// one row of 12 counters per wave (no atomics: 65 536 waves adding to the same 12 words serialise for milliseconds and
// the queue of their atomics delays every other wave's commit)
__device__ unsigned long long* g_scan_phase_rows = nullptr;
#define MS_PH(i) do { const uint64_t ph_now = __builtin_readcyclecounter(); ph_acc[i] += (uint32_t)(ph_now - ph_last); ph_last = ph_now; } while (0)
#define MS_PH_LGKM0() __builtin_amdgcn_s_waitcnt(0xc07f)      // lgkmcnt(0): LDS reads of the phase have landed
#else
#define MS_PH(i) do {} while (0)
#define MS_PH_LGKM0() do {} while (0)
#endif

// Deterministic mode: the per-(patch, splat) sums are committed as 64-bit fixed-point integers with INTEGER atomics.
// Integer addition is associative, so the accumulated row does not depend on the order in which the patches of
// different tiles reach memory and the gradients are bitwise reproducible; everything before the commit (scans,
// per-lane sums, per-wave LDS rows) already runs in program order.  The unit is 2^-e with e chosen PER LAUNCH from
// max |dL/dimage| (ms_fixed_point_exponents; a fixed 2^-32 kept a few bits only of the gradients of a mean-reduced
// loss, dL/dC ~ 1e-7, and rounded the squared heuristic term to 0): e_main for the nine sums that are linear in
// dL/dimage and for split_score, e_h0 for prune_cost's sum of (dL/dalpha)^2.
constexpr int FIXED_POINT_BITS = 36;      // a commit of magnitude max|dL/dimage| is worth 2^36 units
constexpr float VISIBILITY_FIXED_UNIT = 4294967296.0f;      // column 11 (sum of blend weights, <= 1 per pixel): units of 2^-32
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
// Four waves per SIMD (<= 128 VGPRs) is what the 40 KB of LDS per tile-16 workgroup allow, and the kernel is written
// to that budget; without the bound the register allocator of ROCm 7.2 lets the tile-16 instantiation drift to 137
// registers, i.e. three waves (tile 8 is LDS-bound at three waves per SIMD whatever it uses).
// ROWS: `points` = splat-row table (common.h).  SEGS: one workgroup per SEGMENT of a long tile run; the pixel state
// starts from what the forward's composition pass left for the segment (raster_common.h, "Long tile runs")
template <int TS, bool HEUR, int SPLIT = 1, bool ROWS = false, bool SEGS = false>
__global__ void __launch_bounds__(TS * TS, TS == 8 ? 1 : 4)
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
  // Round 6: a heuristics row carries a TWELFTH sum, the blend weights w of the pairs this kernel visits — the splat's
  // visibility (forward.py:127-128) up to the pairs behind a pixel's saturation point, which the backward drops
  // (backward.py:154) and the forward keeps adding: at most 1 - saturate_threshold per pixel.  What the forward sums with
  // a transposing wave reduction per four hits is one addition per pixel step here.  Same LDS: CAP 110 -> 102
  constexpr int CAP = TS == 16 ? (HEUR ? 102 : 128) : (TS == 32) ? (HEUR ? MS_T32_CAP - 32 : MS_T32_CAP) : (HEUR ? MS_T8_CAP - 12 : MS_T8_CAP);
  constexpr int NACC = HEUR ? 12 : 9;
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
  // per sub-patch: patch-list positions of its hits, depth ordered.  (MS_OPT_COMMIT: once a pass has blended, the 4 * CAP
  // bytes of a wave hold the CAP point ids of its patch list for the commit.)
  __shared__ __attribute__((aligned(16))) uint8_t s_list[WAVES][4][CAP];
  static_assert((4 * CAP) % 4 == 0, "a wave's lists double as CAP ints");
  // per-pixel data, read by ALL lanes of the wave at the pixel's step (same address: LDS broadcast; v_readlane
  // from state registers costs ~12-16 cycles per value on gfx950, tools/ubench_scan.hip):
  // [dL/dC.rgb, T] and <R, G>; entry p = 16 * sub-patch + 4 * y + x
  __shared__ float4 s_pix[WAVES][64];
  __shared__ float s_rg[WAVES][64];
  __shared__ int s_done[WAVES];                  // per wave: every pixel of its patch is saturated (tile-wide early out)

#if MS_SCAN_PHASES
  const uint64_t ph_start = __builtin_readcyclecounter();
  uint64_t ph_last = ph_start;
  uint32_t ph_acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
#if MS_PRIO_MODE == 2
  __builtin_amdgcn_s_setprio(2);
#endif
  int tile_id, quarter = 0, seg_start = 0, seg_end = 0;
  if constexpr (SEGS) {
    static_assert(SPLIT == 1, "segments of whole tiles");
    if (rp.split_counts[2] != 0 || (int)blockIdx.x >= rp.split_counts[0]) return;   // (a void plan: raster_fast.hip)
    const int4 item = rp.split_items[blockIdx.x];
    tile_id = item.x; seg_start = item.y; seg_end = item.z;
  } else {
    unsigned quarter_u;
    // the quarter workgroups of a tile run on one XCD (they stage the same list); tiles themselves in plain order
    const int local_tile = xcd_tile<(SPLIT > 1 ? 1 : 0)>(rp.num_tiles, blockIdx.x, SPLIT * SPLIT, &quarter_u);
    if (local_tile < 0) return;
    tile_id = rp.tile_begin + local_tile;
    quarter = (int)quarter_u;
  }
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
      if constexpr (SEGS) {
        // (the forward kernel's thread order inside the wave's 8 x 8 patch: row-major)
        const int fx = (sub & 1) * 4 + (lane & 3), fy = (sub >> 1) * 4 + ((lane >> 2) & 3);
        const float4 st = rp.split_state[(int64_t)blockIdx.x * (TS * TS) + wave * 64 + fy * 8 + fx];
        T = st.w;                                                  // transmittance at the segment's start
        RG -= st.x * G0 + st.y * G1 + st.z * G2;                   // <colour behind the segment's start, G>
      }
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

  const int start = SEGS ? seg_start : ranges[tile_id * 2 + 0], end = SEGS ? seg_end : ranges[tile_id * 2 + 1];
  if (end <= start) return;        // (uniform over the workgroup; the staging below reads the tile's list unguarded)
  // the segment launch has this tile (unless the forward's plan overflowed: every tile then is its per-tile workgroup's)
  if (!SEGS && rp.split_min_run > 0 && end - start > rp.split_min_run && rp.split_counts[2] == 0) return;

  // equal batches: as many as it takes to stay near BATCH_TARGET (rounded to nearest), never above BATCH
  const int total = end - start;
  int num_batches = (total + BATCH_TARGET / 2) / BATCH_TARGET;
  if (num_batches < 1) num_batches = 1;
  int bsz = (total + num_batches - 1) / num_batches;
  if (bsz > BATCH) { num_batches = (total + BATCH - 1) / BATCH; bsz = (total + num_batches - 1) / num_batches; }

  // Staging pipeline (tile 16 / 32: thread t stages slot t of every batch), three batches deep:
  //   rec / rec_id   finished LDS record of the NEXT batch, made at the stage point of the batch before it
  //   raw            gathered data in flight for the batch after that
  //   next_id        point index one batch further on
  // The stage point of a batch sits after its last blend and BEFORE the commit that follows: a register filled by a
  // gather may only be read when no global atomic has been issued behind the gather — loads and atomics share vmcnt,
  // the count of atomics is not known at compile time, and the wait the compiler then places is vmcnt(0): the round
  // trip of every atomic of the commit (round 4 read the gathered data after the barrier that follows the commit: 7 %
  // of a wave's cycles, tools/rbench.py with the -DMS_SCAN_PHASES build).  The records wait in registers across the
  // commit and the barrier and are dead during the blend.
  // Every load of the pipeline is UNCONDITIONAL (list positions beyond the tile's run are clamped to its last entry, and
  // what they fetch is never written to LDS) and stage_next() has ONE call site: the compiler counts outstanding loads
  // per straight-line path, so a divergent branch around a load, or two sites whose registers must be unified at the
  // loop header, each cost a vmcnt(0) — the full latency of gathers issued a few instructions earlier.
  Raw raw = {};
  int next_id = 0;
  ScanRecord rec = {};
  int rec_id = 0;
  auto list_pos = [&](int j, int sl) { const int i = start + j * bsz + sl; return i < end ? i : end - 1; };
  // The SLOTS_B slots beyond the workgroup's thread count (tile 16: 268 - 256 = 12) are gathered ONE COMPONENT PER LANE
  // — slot PRIMARY + (t >> 4), component t & 15: 7 geometry floats, 3 colour floats, the point id (re-read from the
  // list as the lane's "component") — so that they cost every thread two registers instead of a second raw + record
  // set (24 + 13): b_val in flight, b_nid the next id; the slot's leader lane assembles the record in the slot's own LDS
  // space once the barrier has released it.  Lanes without a component load something harmless.
  constexpr int B_THREADS = 16 * SLOTS_B;
  static_assert(B_THREADS <= THREADS, "one lane per component of the extra slots");
  const int b_slot = PRIMARY + ((t >> 4) < (SLOTS_B > 0 ? SLOTS_B : 1) ? (t >> 4) : 0), b_comp = t & 15;
  const bool b_lane = SLOTS_B > 0 && t < B_THREADS && b_comp <= 10;
  float b_val = 0.0f, b_hold = 0.0f;
  int b_nid = 0;
  // per-lane address as integer arithmetic (a select between the three POINTERS becomes a table in scratch memory)
  const bool b_geom = b_comp < 7, b_col = b_comp >= 7 && b_comp < 10;
  auto b_load = [&](int id, int j) {
    const uint64_t pa = reinterpret_cast<uint64_t>(points), fa = reinterpret_cast<uint64_t>(feats),
                   oa = reinterpret_cast<uint64_t>(o2p);
    const int g_row = ROWS ? SPLAT_ROW : 7, c_row = ROWS ? SPLAT_ROW : 3, c_off = ROWS ? SPLAT_ROW_COLOUR : 0;
    const uint64_t a_geom = pa + 4ull * (uint64_t)((int64_t)id * g_row + (b_geom ? b_comp : 0));
    const uint64_t a_col = (ROWS ? pa : fa) + 4ull * (uint64_t)((int64_t)id * c_row + c_off + (b_col ? b_comp - 7 : 0));
    const uint64_t a_id = oa + 4ull * (uint64_t)list_pos(j, b_slot);
    uint64_t addr = b_geom ? a_geom : (b_col ? a_col : a_id);
    asm volatile("" : "+v"(addr));          // (one load, whatever the component; nothing here is loop invariant)
    return *reinterpret_cast<const __attribute__((address_space(1))) float*>(addr);
  };
  // The pipeline fills through the SAME stage_next() call as it runs (the batch loop below starts two stage points
  // early): a second site that loads `raw` would make the loop header unify two register assignments of the in-flight
  // data with copies — and a copy of a register that is being loaded is a vmcnt(0).
  if (PIPELINED) {
    next_id = o2p[list_pos(0, t)];
    if (SLOTS_B > 0) b_nid = o2p[list_pos(0, b_slot)];
  }
  // the stage point of batch `batch`: records of batch + 1 from the data that has landed, gathers of batch + 2, ids of
  // batch + 3
  auto stage_next = [&](int batch) {
    if (!PIPELINED) return;
    // (a tile of three batches passes five stage points; the record of a batch that does not exist is not computed —
    // a wave-uniform branch whose both sides define every register of the record, so that nothing lives across it)
    if (batch + 1 >= 0 && batch + 1 < num_batches) { rec = make_scan_record(raw, rp.alpha_threshold); rec_id = raw.id; }
    else { rec = ScanRecord{}; rec_id = 0; }
    // the record is finished before the gathers are issued: they then load straight into the registers of `raw`.  Left
    // to itself the scheduler hoists the loads above the record arithmetic, lands them in fresh registers and copies
    // them home at the loop latch — behind the commit, i.e. with a wait for every atomic of the commit again.
    asm volatile("" : "+v"(rec.r0.x), "+v"(rec.r0.y), "+v"(rec.r0.z), "+v"(rec.r0.w), "+v"(rec.r1.x), "+v"(rec.r1.y),
                      "+v"(rec.r1.z), "+v"(rec.r1.w), "+v"(rec.r2.x), "+v"(rec.r2.y), "+v"(rec.r2.z), "+v"(rec.r2.w), "+v"(rec_id));
    __builtin_amdgcn_sched_barrier(0);
    // (the id travels on in a register of its own, made by an instruction the compiler cannot fold: the load of the
    // next id then lands in next_id's register instead of a fresh one that is copied over behind the commit)
    int gather_id;
    asm volatile("v_mov_b32 %0, %1" : "=v"(gather_id) : "v"(next_id));
    raw = load_raw<ROWS>(points, feats, gather_id);
    next_id = o2p[list_pos(batch + 3, t)];
    if (SLOTS_B > 0) {
      asm volatile("v_mov_b32 %0, %1" : "=v"(b_hold) : "v"(b_val));       // the gathered component is consumed HERE
      b_val = b_load(b_nid, batch + 2);
      b_nid = o2p[list_pos(batch + 3, b_slot)];
    }
  };

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
  // ---- commit of a pass: ONE 64-byte, line-aligned row of global float atomics per (patch, splat) ----------------
  // ROWS_PER rows of NACC sums per instruction (7 x 9 = 63 lanes; 5 x 12 with heuristics): the LDS reads sweep
  // the wave's accumulator block linearly and the loop runs pcount / 7 times (16 lanes per row, 9 of them with
  // data, ran pcount / 4 times: 1.45 -> 1.37 ms on config D; the atomics themselves are 0.10 ms of instruction
  // rate + 0.04 ms of misses: 1.33 ms when every row lands in a 16 MB window, 1.21 ms without the commit).
  // Round 5: the pass has blended, so its sub-patch lists are dead and take the point ids of the patch list (two
  // dependent reads, once per pass, 64 entries at a time): a commit step is then ONE LDS round trip — the sums and the id
  // of its rows — requested a step ahead, where round 4 walked sum -> staged index -> point id for every step
  // (10.7 % of a wave's cycles on config D).
  auto commit_pass = [&](int pcount) {
    constexpr int ROWS_PER = 64 / NACC;
    const int sub_row = lane / NACC, k = lane - sub_row * NACC;
    const bool lane_used = sub_row < ROWS_PER;
    int* s_pid = reinterpret_cast<int*>(&s_list[wave][0][0]);
    for (int pp = lane; pp < pcount; pp += 64) s_pid[pp] = s_id[s_plist[wave][pp]];
    wave_lds_fence();
    float v = 0.0f;
    int point = 0;
    if (lane_used && sub_row < pcount) { v = s_acc[wave][sub_row][k]; point = s_pid[sub_row]; }
    for (int e0 = 0; e0 < pcount; e0 += ROWS_PER) {
      const int e = e0 + sub_row, en = e + ROWS_PER;
      float vn = 0.0f;
      int pointn = 0;
      if (lane_used && en < pcount) { vn = s_acc[wave][en][k]; pointn = s_pid[en]; }
      if (v != 0.0f) {
        const size_t word = (size_t)(uint32_t)point * MOMENT_ROW + k;
        if (rp.deterministic)
          __hip_atomic_fetch_add(reinterpret_cast<long long*>(moments) + word,
                                 (long long)llrintf(v * (k == 9 ? fixed_h0 : k == 11 ? VISIBILITY_FIXED_UNIT : fixed_main)),
                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else {
#if MS_COMMIT_ABLATE == 0
          atomic_add_noret(moments + word, v);
#elif MS_COMMIT_ABLATE == 1          // timing experiments (wrong gradients): one lane per row
          if (k == 0) atomic_add_noret(moments + word, v);
#elif MS_COMMIT_ABLATE == 2          // plain stores to the same addresses
          moments[word] = v;
#elif MS_COMMIT_ABLATE == 3          // no global traffic at all
#endif
        }
        s_acc[wave][e][k] = 0.0f;
      }
      v = vn; point = pointn;
    }
    wave_lds_fence();
  };

  for (int batch = PIPELINED ? -2 : 0;; ++batch) {
    bool wave_alive = false;
    int pcount = 0;
    if (batch >= 0) {
    const int begin = start + batch * bsz;
    if (begin >= end) break;
    const int count = (end - begin) < bsz ? (end - begin) : bsz;
    // all waves are done with the previous batch; tile-wide early out once every pixel is saturated
    // (backward.py:116)
    wave_lds_fence();
#if MS_SCAN_PHASES
    if (batch == 0) MS_PH(8);
#endif
    // one barrier (__syncthreads_and is a DPP reduction, an LDS word and THREE s_barrier on gfx950)
    {
      const bool wave_done = __ballot(__float_as_uint(s_pix[wave][lane].w) > oms_bits) == 0;
      if (lane == 0) s_done[wave] = wave_done ? 1 : 0;
    }
    __syncthreads();
    const bool tile_done = __ballot(s_done[lane % WAVES] != 0) == ~0ull;
    MS_PH(0);
#if MS_SCAN_STATS
    MS_FLUSH_BALANCE()
#endif
    if (tile_done) break;

    if (PIPELINED) {
      if (t < count) {
        s_rec[t * 3 + 0] = rec.r0; s_rec[t * 3 + 1] = rec.r1; s_rec[t * 3 + 2] = rec.r2;
        s_id[t] = rec_id;
      }
      if (SLOTS_B > 0 && PRIMARY < count) {
        // the extra slots: every component lane drops its value into the slot's own (released) record space, the
        // slot's leader lane reads the ten floats back, makes the record and overwrites them
        int bs = b_slot;
        asm volatile("" : "+v"(bs));       // (the slot's LDS addresses are made here, not kept in registers for the whole kernel)
        if (b_lane && bs < count) {
          if (b_comp < 10) reinterpret_cast<float*>(&s_rec[bs * 3])[b_comp] = b_hold;
          else s_id[bs] = __float_as_int(b_hold);
        }
        wave_lds_fence();
        if (t < B_THREADS && b_comp == 0 && bs < count) {
          const float4 x0 = s_rec[bs * 3 + 0], x1 = s_rec[bs * 3 + 1], x2 = s_rec[bs * 3 + 2];
          Raw rb;
          rb.g[0] = x0.x; rb.g[1] = x0.y; rb.g[2] = x0.z; rb.g[3] = x0.w; rb.g[4] = x1.x; rb.g[5] = x1.y; rb.g[6] = x1.z;
          rb.f[0] = x1.w; rb.f[1] = x2.x; rb.f[2] = x2.y;
          rb.id = 0;
          write_scan_record(rb, rp.alpha_threshold, &s_rec[bs * 3]);
        }
      }
    } else {
      for (int s = t; s < count; s += THREADS) {
        const Raw r = load_raw<ROWS>(points, feats, o2p[begin + s]);
        write_scan_record(r, rp.alpha_threshold, &s_rec[s * 3]);
        s_id[s] = r.id;
      }
    }
    MS_PH(1);
    __syncthreads();
    MS_PH(2);

    // wave-wide early out (backward.py:142): a wave whose 64 pixels are saturated skips the passes — but not the batch's
    // stage point below (one site for every path)
    wave_alive = __ballot(__float_as_uint(s_pix[wave][lane].w) > oms_bits) != 0;
#if MS_SCAN_ABLATE == 2
    wave_alive = false;
#endif

#if MS_SCAN_STATS
    // balance of the blend work between the waves of a workgroup: per batch, sum and WAVES x max of the chunks
    // the waves ran (a batch ends at a barrier, so the slowest wave sets its length)
    int batch_chunks = 0;
#endif
    // From here to the next barrier the wave works alone: it walks the staged batch in passes of at most CAP
    // patch hits (one pass per batch unless most staged splats touch this 8x8 patch).
    // The LAST pass of a batch is committed below the loop, behind the batch's stage point.
    int r = 0;
    while (wave_alive) {
      // ---- cull, level 1: the staged splats that can touch this wave's 8x8 patch -> patch list ------------------
      pcount = 0;
      const float pcx = (float)patch_x + 4.0f, pcy = (float)patch_y + 4.0f;
      // the records of the NEXT 64 staged splats are requested before the current 64 are tested: one exposed LDS round
      // trip per pass instead of one per group (a lane beyond the batch reads record 0 and ignores it)
      auto group_records = [&](int first, float4& a, float4& b) {
        const int j = first + lane < count ? first + lane : 0;
        a = s_rec[j * 3 + 0]; b = s_rec[j * 3 + 1];
      };
      float4 g0, g1;
      group_records(r, g0, g1);
      while (r < count) {
        float4 n0 = g0, n1 = g1;
        if (r + 64 < count) group_records(r + 64, n0, n1);
        const int j = r + lane;
        const bool hit = j < count && scan_rect_hit(g0, g1, pcx, pcy, 3.5f);
        const unsigned long long m = __ballot(hit);
        const int nhit = __builtin_popcountll(m);
        if (pcount + nhit > CAP) break;          // next pass (nhit <= 64 <= CAP: an empty list always takes the group)
        const int ppos = pcount + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (hit) s_plist[wave][ppos] = (uint16_t)j;
        pcount += nhit;
        r += 64;
        g0 = n0; g1 = n1;
      }
      wave_lds_fence();
      // ---- cull, level 2: only the patch hits (about 40 % of a batch) are tested against the four 4x4 sub-patches;
      // both lists are appended in order, so they stay depth sorted
      int cnt[4] = {0, 0, 0, 0};
      {
        // CAP <= 128: at most two groups of patch hits; their staged indices, then their records, are requested
        // together (two exposed round trips per pass instead of two per group)
        static_assert(CAP <= 128, "level-2 cull: two groups");
        const bool two = pcount > 64;
        const bool in_a = lane < pcount, in_b = 64 + lane < pcount;
        const int ja = in_a ? (int)s_plist[wave][lane] : 0;
        const int jb = (two && in_b) ? (int)s_plist[wave][64 + lane] : 0;
        const float4 a0 = s_rec[ja * 3 + 0], a1 = s_rec[ja * 3 + 1];
        float4 b0 = a0, b1 = a1;
        if (two) { b0 = s_rec[jb * 3 + 0]; b1 = s_rec[jb * 3 + 1]; }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          if (u == 1 && !two) break;
          const bool in = u == 0 ? in_a : in_b;
          const float4 q0 = u == 0 ? a0 : b0, q1 = u == 0 ? a1 : b1;
          const int ppos = u * 64 + lane;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float rcx = (float)(patch_x + (q & 1) * 4) + 2.0f, rcy = (float)(patch_y + (q >> 1) * 4) + 2.0f;
            const bool hit = in && scan_rect_hit(q0, q1, rcx, rcy, 1.5f);
            const unsigned long long m = __ballot(hit);
            const int pos = cnt[q] + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (hit) s_list[wave][q][pos] = (uint8_t)ppos;
            cnt[q] += __builtin_popcountll(m);
          }
        }
      }
      wave_lds_fence();
      MS_PH(3);
#if MS_SCAN_STATS
      if (lane == 0) {
        atomicAdd(&g_scan_stats[0], 1ull);                                                      // passes
        atomicAdd(&g_scan_stats[1], (unsigned long long)(cnt[0] + cnt[1] + cnt[2] + cnt[3]));   // (sub-patch, splat) hits
        atomicAdd(&g_scan_stats[6], (unsigned long long)pcount);                                // (patch, splat) hits
      }
#endif

      // ---- blend: lane = splat, 16 pixel steps per chunk ------------------------------------------------------
#if MS_SCAN_ABLATE != 1
      MS_PRIO_BLEND(true);
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
          const float nl2a = valid ? q2.x : __builtin_inff();       // idle lanes: alpha g = exp2(-inf) = 0
          const float f0 = q2.y, f1 = q2.z, f2 = q2.w;
          const float dx0 = fx - q0.x, dy0 = fy - q0.y;             // first pixel centre of the sub-patch - mean
          const float X00 = A * dx0 + B * dy0, Y00 = C * dx0 + D * dy0;
          // (X', Y') at the first pixel of each of the four pixel rows; a step adds x * (A, C)
          const float Xr[4] = {X00, X00 + B, __builtin_fmaf(B, 2.0f, X00), __builtin_fmaf(B, 3.0f, X00)};
          const float Yr[4] = {Y00, Y00 + D, __builtin_fmaf(D, 2.0f, Y00), __builtin_fmaf(D, 3.0f, Y00)};

          float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f, m5 = 0.f, a0 = 0.f, a1 = 0.f, a2 = 0.f;
          float h0 = 0.f, h1 = 0.f, vs = 0.f;
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
#if MS_SCAN_PHASES
          MS_PH_LGKM0();
          MS_PH(4);
          ++ph_acc[10];
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
                  vs += w[u];                                         // column 11 (see NACC)
                  // (d alpha)^2 of the pairs that blend: the mask as a clamped multiply (a_st is 0 or above the blend
                  // gate), and |q tx| + |q ty| as |q| (|tx| + |ty|) — the absolute values are source modifiers: six VALU
                  // instructions per pixel where compare / select and two products took ten
                  const float agm = ag * __builtin_amdgcn_fmed3f(a_st[u] * 0x1p60f, 0.0f, 1.0f);
                  h0 = __builtin_fmaf(agm, agm, h0);
                  if (GRID) {
                    const int x = (i + u) & 3, y = i >> 2;
                    const float bx = y == 0 ? gx0 : __builtin_fmaf(gxy, (float)y, gx0), by = y == 0 ? gy0 : __builtin_fmaf(gyy, (float)y, gy0);
                    const float tx = x == 0 ? bx : __builtin_fmaf(gxx, (float)x, bx), ty = x == 0 ? by : __builtin_fmaf(gxy, (float)x, by);
                    h1 = __builtin_fmaf(fabsf(q_), fabsf(tx) + fabsf(ty), h1);
                  } else {
                    h1 = __builtin_fmaf(fabsf(q_), fabsf(__builtin_fmaf(X[u], A, Y[u] * C)) + fabsf(__builtin_fmaf(X[u], B, Y[u] * D)), h1);
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

          MS_PH(5);
          // this wave's row of the splat: the lanes of a chunk hold distinct splats, chunks run one after the other
          if (valid) {
            float* row = &s_acc[wave][pos][0];
            const float v[12] = {m0, m1, m2, m3, m4, m5, a0, a1, a2, h0, h1, vs};
#pragma unroll
            for (int k = 0; k < NACC; ++k) row[k] += v[k];
          }
          MS_PH(6);
        }
      }

      MS_PRIO_BLEND(false);
#endif      // MS_SCAN_ABLATE != 1
      wave_lds_fence();
      MS_PH(4);        // (the sub-patch loop's own bookkeeping between the last chunk and the commit)
      if (r >= count) break;       // last pass of the batch: committed below, behind the stage point
      commit_pass(pcount);
      MS_PH(7);
    }
#if MS_SCAN_STATS
    if (lane == 0) { atomicAdd(&s_bsum, batch_chunks); atomicMax(&s_bmax, batch_chunks); }
#endif
    }       // batch >= 0
    // stage point: the gathers issued a batch ago have landed and no atomic is in flight behind them
    stage_next(batch);
    MS_PH(1);
    if (wave_alive) commit_pass(pcount);
    MS_PH(7);
  }
#if MS_SCAN_PHASES
  if (lane == 0 && g_scan_phase_rows) {
    const uint64_t ph_end = __builtin_readcyclecounter();
    unsigned long long* row = g_scan_phase_rows + ((size_t)blockIdx.x * WAVES + wave) * 12;
#pragma unroll
    for (int i = 0; i < 9; ++i) row[i] = ph_acc[i];
    row[9] = ph_end - ph_start; row[10] = ph_acc[10]; row[11] = 1;
  }
#endif
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

// Measurement hook (include/mi355_splat.h: ms_probe_raster_bwd): the next launch of the product backward records these two
// events around its per-tile kernel, on the stream it is launched on — the kernel's duration INSIDE a frame (bench.py).
static hipEvent_t g_probe_start = nullptr, g_probe_stop = nullptr;
extern "C" int ms_probe_raster_bwd(void* start_event, void* stop_event) {
  g_probe_start = (hipEvent_t)start_event; g_probe_stop = (hipEvent_t)stop_event;
  return 0;
}

static int launch_scan_backward(const float* points7, const float* features,
                                const int32_t* tile_ranges, const int32_t* overlap_to_point, const float* image,
                                const float* grad_image, int image_w, int image_h, const ms_raster_config* cfg,
                                float* moments, int deterministic, const int32_t* fixed_exp, int tile_row_begin,
                                int tile_row_end, hipStream_t s, const char* who, int grad_broadcast = 0,
                                const float* splat_rows = nullptr, const SplitScratch* split = nullptr) {
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

  FastParams rp{};
  rp.width = image_w; rp.height = image_h; rp.tiles_wide = tiles_wide; rp.tile_begin = tile_row_begin * tiles_wide;
  rp.clamp_max_alpha = (float)cfg->clamp_max_alpha;
  rp.alpha_threshold = (float)cfg->alpha_threshold;
  rp.one_minus_saturate = (float)(1.0 - cfg->saturate_threshold);
  rp.deterministic = deterministic != 0;
  rp.grad_broadcast = grad_broadcast != 0;
  rp.num_tiles = (tile_row_end - tile_row_begin) * tiles_wide;
  if (split && ts == 32 && tile32_quarters()) split = nullptr;     // (the quarter variant walks whole tile lists)
  if (split) {
    rp.split_min_run = split->min_run;
    rp.split_items = split->items; rp.split_counts = split->counts; rp.split_state = split->state;
  }
#define MS_GO(TS, HEUR, SPLIT, ROWS) raster_bwd_scan_kernel<TS, HEUR, SPLIT, ROWS>                               \
      <<<dim3(xcd_grid<(SPLIT > 1 ? 1 : 0)>(rp.num_tiles, SPLIT * SPLIT)), dim3(TS * TS), 0, s>>>(              \
          ROWS ? splat_rows : points7, features, tile_ranges, overlap_to_point, image, grad_image, rp, moments, fixed_exp)
#define MS_GO_TILE(TS, SPLIT, ROWS)                                                                             \
  do { if (hf) MS_GO(TS, true, SPLIT, ROWS); else MS_GO(TS, false, SPLIT, ROWS); } while (0)
  // the segments of long tile runs (the plan and the start states are the forward's: raster_common.h); dense arrays only
#define MS_GO_SEGS(TS)                                                                                          \
  do {                                                                                                          \
    if (hf) raster_bwd_scan_kernel<TS, true, 1, false, true><<<dim3((unsigned)split->item_cap), dim3(TS * TS), 0, s>>>(  \
        points7, features, tile_ranges, overlap_to_point, image, grad_image, rp, moments, fixed_exp);          \
    else raster_bwd_scan_kernel<TS, false, 1, false, true><<<dim3((unsigned)split->item_cap), dim3(TS * TS), 0, s>>>(    \
        points7, features, tile_ranges, overlap_to_point, image, grad_image, rp, moments, fixed_exp);          \
  } while (0)
  // the splat-row table serves the one-workgroup-per-tile kernels of tile 16 and 32 (tile 8 gathers inside its pass
  // loop and measured 10 % SLOWER with the 16-byte loads; the quarter-tile variant is a fallback)
  const bool hf = cfg->compute_point_heuristic;
#ifdef MS_WITH_ROWS_KERNEL      // tools/experiments/raster_bwd_rows.hip (measured and dropped in round 4)
  if (launch_rows_backward(points7, features, tile_ranges, overlap_to_point, image, grad_image, rp, ts, hf, moments,
                           fixed_exp, s)) {
    MS_CHECK_LAUNCH();
    return 0;
  }
#endif
  const hipEvent_t probe_start = g_probe_start, probe_stop = g_probe_stop;
  g_probe_start = g_probe_stop = nullptr;                      // one launch per arming
  if (probe_start) (void)hipEventRecord(probe_start, s);
#define MS_PROBE_STOP() do { if (probe_stop) (void)hipEventRecord(probe_stop, s); } while (0)
  switch (ts) {
    case 8: MS_GO_TILE(8, 1, false); MS_PROBE_STOP(); if (split) MS_GO_SEGS(8); break;
    case 16: if (splat_rows) MS_GO_TILE(16, 1, true); else MS_GO_TILE(16, 1, false); MS_PROBE_STOP(); if (split) MS_GO_SEGS(16); break;
    default:
      // tile 32: ONE 1024-thread workgroup per tile with 896-splat batches (152 KB LDS), or — MS_TILE32_BWD=quarters —
      // four 16 x 16 quarter workgroups per tile that each stage the whole tile list
      if (tile32_quarters()) MS_GO_TILE(16, 2, false);
      else if (splat_rows) MS_GO_TILE(32, 1, true);
      else MS_GO_TILE(32, 1, false);
      MS_PROBE_STOP();
      if (split) MS_GO_SEGS(32);
      break;
  }
#undef MS_PROBE_STOP
#undef MS_GO_SEGS
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

extern "C" int ms_raster_bwd_moments_rows(const float* rows, const int32_t* tile_ranges, const int32_t* overlap_to_point,
                                          const float* image, const float* grad_image, int image_w, int image_h,
                                          const ms_raster_config* cfg, float* moments, int deterministic,
                                          const int32_t* fixed_exp, int tile_row_begin, int tile_row_end, void* stream) {
  MS_CHECK_ARG(cfg && rows && tile_ranges && image && grad_image && moments, "null pointer");
  // (tile 8 and the quarter-tile variant of tile 32 gather row by row from the same table: points7 at stride
  // MS_SPLAT_ROW is not what their dense loads expect, so they are not offered here)
  if (cfg->tile_size == 8 || (cfg->tile_size == 32 && tile32_quarters())) {
    set_error("ms_raster_bwd_moments_rows: tile 16, or tile 32 with one workgroup per tile");
    return MS_ERR_UNSUPPORTED;
  }
  return launch_scan_backward(rows, rows, tile_ranges, overlap_to_point, image, grad_image, image_w, image_h, cfg, moments,
                              deterministic, fixed_exp, tile_row_begin, tile_row_end, (hipStream_t)stream,
                              "ms_raster_bwd_moments_rows", 0, rows);
}

extern "C" int ms_raster_bwd_moments_split(const float* points7, const float* features, const int32_t* tile_ranges,
                                           const int32_t* overlap_to_point, int64_t k_capacity, const float* image,
                                           const float* grad_image, int image_w, int image_h,
                                           const ms_raster_config* cfg, float* moments, int deterministic,
                                           const int32_t* fixed_exp, const void* split_scratch, int split_min_run,
                                           int split_seg_len, int tile_row_begin, int tile_row_end, void* stream) {
  MS_CHECK_ARG(cfg && points7 && features && tile_ranges && image && grad_image && moments && split_scratch, "null pointer");
  MS_CHECK_ARG(k_capacity >= 0 && k_capacity < (1ll << 31), "k_capacity must be in [0, 2^31)");
  MS_CHECK_ARG(split_min_run >= 0 && split_seg_len >= 0, "negative split parameter");
  MS_CHECK_ARG((reinterpret_cast<uintptr_t>(split_scratch) & 255) == 0, "split_scratch must be 256-byte aligned");
  if (cfg->tile_size != 8 && cfg->tile_size != 16 && cfg->tile_size != 32) {
    set_error("ms_raster_bwd_moments_split: tile_size must be 8, 16 or 32 (got %d)", cfg->tile_size);
    return MS_ERR_UNSUPPORTED;
  }
  const SplitScratch sc = split_scratch_carve(const_cast<void*>(split_scratch), k_capacity, cfg->tile_size,
                                              split_params(cfg->tile_size, split_min_run, split_seg_len));
  return launch_scan_backward(points7, features, tile_ranges, overlap_to_point, image, grad_image, image_w, image_h, cfg,
                              moments, deterministic, fixed_exp, tile_row_begin, tile_row_end, (hipStream_t)stream,
                              "ms_raster_bwd_moments_split", 0, nullptr, &sc);
}

extern "C" int ms_fixed_point_exponents(const float* amax_dev, int32_t* out_exp2, void* stream) {
  MS_CHECK_ARG(amax_dev && out_exp2, "null pointer");
  fixed_point_exponents_kernel<<<1, 1, 0, (hipStream_t)stream>>>(amax_dev, out_exp2);
  MS_CHECK_LAUNCH();
  return 0;
}

#if MS_SCAN_PHASES
// rows: device buffer of (waves of the launch) x 12 uint64 the next launches fill (NULL: stop recording)
extern "C" int ms_debug_scan_phases(unsigned long long* rows, int unused) {
  (void)unused;
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_scan_phase_rows), &rows, sizeof(rows));
  return 0;
}
#endif

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
                              hipStream_t s, const float* splat_rows, const SplitScratch* split) {
  return launch_scan_backward((const float*)points7, (const float*)features, tile_ranges, overlap_to_point,
                              (const float*)image, (const float*)grad_image, image_w, image_h, cfg, moments,
                              deterministic, fixed_exp, tile_row_begin, tile_row_end, s, "ms_frame_backward", grad_broadcast,
                              splat_rows, split);
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
