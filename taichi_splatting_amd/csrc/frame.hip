// frame.hip — the frame executor: one frame of the render path (project -> SH colour -> overlap count / scan / emit
// -> stable sort on the tile bits -> ranges -> per-tile depth sort [or: depth pre-sort -> count / scan / emit -> tile
// sort -> ranges; ms_frame_desc.mapper] -> raster forward; raster backward -> one per-gaussian backward pass) as
// a FIXED sequence of launches on one stream, with no host round trip in between (include/mi355_splat.h,
// "frame executor").  The reference orchestrates the same stages from Python with two device synchronisations per
// frame (the visible count in torch.nonzero, perspective/projection.py:147-150; the overlap total,
// cuda_lib/full_cumsum.cu:45-46, mapper/tile_mapper.py:183-190); here neither count ever leaves the device:
//
//   * visible count: not needed — nothing is compacted.  A culled gaussian keeps its row, carries depth 0 (and the
//     sort key CULLED_DEPTH_KEY on the pre-sort sequence), overlaps no tile and receives zero gradients;
//   * overlap total K: the kernels downstream of the scan read it from `counters` and run on capacity-sized grids.
//
// A caller may still LOOK at K (k_host / k_event) — after everything is enqueued — to grow its buffers.
#include <cstdlib>
#include <cstring>
#include "common.h"
#include "frame_internal.h"
#include "raster_common.h"

namespace ms {

struct FrameGeom {
  int w_pad, h_pad, tiles_wide, tiles_high, num_tiles, tile_bits, row_begin, row_end;
  size_t es;            // element size of the float type
};

static FrameGeom frame_geom(const ms_frame_desc* d) {
  FrameGeom g;
  const int ts = d->raster.tile_size;
  g.tiles_wide = (d->image_w + ts - 1) / ts;
  g.tiles_high = (d->image_h + ts - 1) / ts;
  g.w_pad = g.tiles_wide * ts;
  g.h_pad = g.tiles_high * ts;
  g.num_tiles = g.tiles_wide * g.tiles_high;
  int bits = 1;
  while ((1ll << bits) < (long long)g.num_tiles) ++bits;
  g.tile_bits = bits;
  g.row_begin = d->tile_row_begin < 0 ? 0 : d->tile_row_begin;
  g.row_end = d->tile_row_end > g.tiles_high ? g.tiles_high : d->tile_row_end;
  g.es = d->dtype == MS_F64 ? 8 : 4;
  return g;
}

static int check_desc(const ms_frame_desc* d, const char* who) {
  if (!d) { set_error("%s: desc is null", who); return MS_ERR_BAD_ARG; }
  // the caller's header: another struct size or ABI generation means its fields are not where this library reads them
  if (d->struct_size != sizeof(ms_frame_desc) || MS_ABI_GENERATION(d->abi_version) != MS_ABI_GENERATION(MS_VERSION)) {
    set_error("%s: ms_frame_desc of another ABI (struct_size %u, abi_version %u; this library: %u, %d) — set desc.struct_size = "
              "sizeof(ms_frame_desc) and desc.abi_version = MS_VERSION of the include/mi355_splat.h you compile against",
              who, d->struct_size, d->abi_version, (unsigned)sizeof(ms_frame_desc), MS_VERSION);
    return MS_ERR_ABI;
  }
  if (d->split_long_runs < 0 || d->split_seg_len < 0) { set_error("%s: negative split parameter", who); return MS_ERR_BAD_ARG; }
  if (d->n < 0 || d->k_capacity < 0) { set_error("%s: negative size", who); return MS_ERR_BAD_ARG; }
  if (d->n >= (1ll << 31) || d->k_capacity >= (1ll << 31)) { set_error("%s: sizes are int32 indexes (< 2^31)", who); return MS_ERR_BAD_ARG; }
  if (d->image_w <= 0 || d->image_h <= 0) { set_error("%s: bad image size", who); return MS_ERR_BAD_ARG; }
  if (d->dtype != MS_F32 && d->dtype != MS_F64) { set_error("%s: dtype must be MS_F32 or MS_F64", who); return MS_ERR_BAD_ARG; }
  const int ts = d->raster.tile_size;
  if (ts != 8 && ts != 16 && ts != 32) { set_error("%s: tile_size must be 8, 16 or 32 (got %d)", who, ts); return MS_ERR_UNSUPPORTED; }
  if (d->f < 1 || d->f > 4) { set_error("%s: 1..4 colour channels (got %d)", who, d->f); return MS_ERR_UNSUPPORTED; }
  if (d->sh_degree < -1 || d->sh_degree > 3) { set_error("%s: SH degree must be in [0, 3]", who); return MS_ERR_BAD_ARG; }
  if (d->mapper != MS_MAPPER_DIRECT && d->mapper != MS_MAPPER_PRESORT) { set_error("%s: mapper must be MS_MAPPER_DIRECT or MS_MAPPER_PRESORT (got %d)", who, d->mapper); return MS_ERR_BAD_ARG; }
  if (d->projected_input && d->sh_degree >= 0) { set_error("%s: projected input carries colours, not SH", who); return MS_ERR_BAD_ARG; }
  if (d->depth16) {
    const FrameGeom g = frame_geom(d);
    if (g.num_tiles > 65536) { set_error("%s: use_depth16 keys hold a 16 bit tile id: too many tiles", who); return MS_ERR_BAD_ARG; }
  }
  return 0;
}

// Splat rows (common.h) inside the frame executor: OFF unless MS_SPLAT_ROWS=1 is in the environment (read once).  The
// raster kernels gain 0.057 + 0.030 ms on config D from the one-line gathers, but FILLING the table costs the frame
// more than that: the projection kernel's 32-byte pieces and the SH kernel's 16-byte pieces are partial writes of
// 64-byte sectors, which HBM serves at ~1.4 TB/s (projection 0.088 -> 0.230 ms, SH 0.243 -> 0.337 ms; frame 3.16 ->
// 3.37 ms, profiles/r05_splat_rows.txt), and one kernel writing whole rows has to move 0.58 GB more than the frame
// does today (~0.10 ms).  The table pays where rows arrive complete or are rasterized more than once — the C-ABI
// entry points ms_splat_rows_pack / ms_raster_fwd_rows / ms_raster_bwd_moments_rows — and stays here as a switch for
// measurements.
static bool frame_uses_rows(const ms_frame_desc* d) {
  static const bool on = [] { const char* e = getenv("MS_SPLAT_ROWS"); return e && e[0] == '1'; }();
  return on && !d->projected_input && d->n > 0 && raster_uses_splat_rows(&d->raster, d->f, d->dtype);
}

// Long tile runs cut into segments (raster_common.h): float32 RGB frames on the product kernels
static bool frame_uses_split(const ms_frame_desc* d) {
  return d->split_long_runs != 0 && d->n > 0 && d->k_capacity > 0 && raster_uses_splat_rows(&d->raster, d->f, d->dtype);
}
// split_long_runs = 1: the default threshold; > 1: the threshold itself; split_seg_len = 0: the default segment length
static SplitParams frame_split_params(const ms_frame_desc* d) {
  return split_params(d->raster.tile_size, d->split_long_runs, d->split_seg_len);
}

static int check_inputs(const ms_frame_inputs* in, const char* who) {
  if (!in) { set_error("%s: inputs are null", who); return MS_ERR_BAD_ARG; }
  if (in->struct_size != sizeof(ms_frame_inputs)) {
    set_error("%s: ms_frame_inputs of another ABI (struct_size %u, this library: %u)", who, in->struct_size, (unsigned)sizeof(ms_frame_inputs));
    return MS_ERR_ABI;
  }
  return 0;
}

static int check_grads(const ms_frame_grads* g, const char* who) {
  if (!g) { set_error("%s: grads are null", who); return MS_ERR_BAD_ARG; }
  if (g->struct_size != sizeof(ms_frame_grads)) {
    set_error("%s: ms_frame_grads of another ABI (struct_size %u, this library: %u)", who, g->struct_size, (unsigned)sizeof(ms_frame_grads));
    return MS_ERR_ABI;
  }
  return 0;
}

struct Carve {
  size_t off = 0;
  size_t take(size_t bytes) { const size_t o = off; off += align_up(bytes > 0 ? bytes : 1, 256); return o; }
};

static void frame_layout(const ms_frame_desc* d, ms_frame_layout* L) {
  const FrameGeom g = frame_geom(d);
  const size_t n = (size_t)d->n, k = (size_t)d->k_capacity;
  const bool own_points = !d->projected_input;
  Carve keep_n, scratch_n, keep_k, scratch_k;
  L->points7 = keep_n.take(own_points ? n * 7 * g.es : 0);
  L->depth = keep_n.take(own_points ? n * g.es : 0);
  L->colours = keep_n.take(d->sh_degree >= 0 ? n * d->f * g.es : 0);
  L->camera_position = keep_n.take(4 * g.es);
  L->counters = keep_n.take(8 * sizeof(int32_t));
  L->tile_ranges = keep_n.take((size_t)g.num_tiles * 2 * sizeof(int32_t));
  L->splat_rows = keep_n.take(frame_uses_rows(d) ? n * SPLAT_ROW * sizeof(float) : 0);
  L->keep_n_bytes = keep_n.off;

  L->sorted_keys = scratch_n.take(n * 4);
  L->order = scratch_n.take(n * 4);
  L->counts = scratch_n.take(n * 4);
  L->cum = scratch_n.take((n + 1) * 4);
  L->ordered_points = scratch_n.take(n * 7 * 4);
  L->points7_f32 = scratch_n.take(d->dtype == MS_F64 ? n * 7 * 4 : 0);
  const size_t t1 = sort_tmp_size(d->n, 4, true), t2 = scan_tmp_size(d->n);
  L->tmp_n = scratch_n.take(t1 > t2 ? t1 : t2);
  L->scratch_n_bytes = scratch_n.off;

  L->overlap_to_point = keep_k.take(k * 4);
  L->split_scratch = keep_k.take(frame_uses_split(d) ? split_scratch_bytes(d->k_capacity, d->raster.tile_size, frame_split_params(d)) : 0);
  L->keep_k_bytes = keep_k.off;

  // 8 byte keys (tile << 32 | depth key) of the direct-order mapper; the pre-sort path (MS_MAPPER=presort) uses half
  L->keys = scratch_k.take(k * 8);
  L->values = scratch_k.take(k * 4);
  L->keys_sorted = scratch_k.take(k * 8);
  L->tmp_k = scratch_k.take(sort_tmp_size(d->k_capacity, 8));
  L->scratch_k_bytes = scratch_k.off;
}

// What ms_frame_map_raster needs before its first real kernel, in ONE launch (a launch boundary costs ~5 us on this
// chip whatever the kernel does; K itself is written to counters[0] and the caller's pinned word by the last block
// of the overlap scan):
//   * live K against the capacity of this call's overlap buffers: 0 (nothing is emitted, sorted or ranged) on
//     overflow — an int32 total that wrapped negative counts as overflow;
//   * the camera position for the SH colours (cam_out != NULL);
//   * the zero fill of the tile ranges (empty tiles stay [0, 0), tile_mapper.py:93-112).
template <typename T>
__global__ void __launch_bounds__(256)
frame_prepare_kernel(int32_t* __restrict__ counters, int32_t capacity, const T* __restrict__ Tcw, T* __restrict__ cam_out,
                     int32_t* __restrict__ ranges, int64_t range_words) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < range_words) ranges[i] = 0;
  if (i == 0) {
    const int32_t k = counters[0];
    const bool ok = k >= 0 && k <= capacity;
    counters[1] = ok ? k : 0;
    counters[2] = ok ? 0 : 1;
    counters[3] = 0;                 // longest run of the per-tile sort's one-workgroup path, and the ticket of its
    counters[4] = 0;                 // workgroups, "a run was declined" (tile_sort.hip)
    counters[5] = 0;
    if (cam_out) camera_position_solve(Tcw, cam_out);
  }
}

__global__ void __launch_bounds__(256)
f64_to_f32_kernel(const double* __restrict__ in, float* __restrict__ out, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = (float)in[i];
}

// Which of the two mapper sequences a frame runs (include/mi355_splat.h: MS_MAPPER_DIRECT / MS_MAPPER_PRESORT; both
// produce the same overlap_to_point / tile_ranges).  MS_MAPPER=direct|presort in the environment overrides the
// descriptor for the whole process; read once.
static bool mapper_presort(const ms_frame_desc& d) {
  static const int forced = [] {
    const char* e = getenv("MS_MAPPER");
    if (e && strcmp(e, "presort") == 0) return MS_MAPPER_PRESORT;
    if (e && strcmp(e, "direct") == 0) return MS_MAPPER_DIRECT;
    return -1;
  }();
  return (forced >= 0 ? forced : d.mapper) == MS_MAPPER_PRESORT;
}

static bool frame_uses_moments(const ms_frame_desc* d, int deterministic) {
  // raster_bwd_scan.hip: float32 RGB, plain pdf, every tile size (rasterizer/function.py::_use_moments_backward)
  (void)deterministic;
  return d->dtype == MS_F32 && d->f == 3 && !d->raster.antialias && d->raster.use_alpha_blending;
}

}  // namespace ms

using namespace ms;

extern "C" int ms_frame_layout_query(const ms_frame_desc* desc, ms_frame_layout* out) {
  int rc = check_desc(desc, "ms_frame_layout_query");
  if (rc) return rc;
  MS_CHECK_ARG(out != nullptr, "out is null");
  frame_layout(desc, out);
  return 0;
}

extern "C" int ms_frame_uses_moments(const ms_frame_desc* desc, int deterministic) {
  return desc && frame_uses_moments(desc, deterministic) ? 1 : 0;
}

#define MS_TRY(expr)            \
  do {                          \
    const int rc__ = (expr);    \
    if (rc__ != 0) return rc__; \
  } while (0)

// per-gaussian forward stage into keep_n: camera position + projection (+ SH colours)
static int frame_project_impl(const ms_frame_desc* desc, const ms_frame_inputs* in, void* keep_n, bool projection,
                              bool colours, void* stream, const char* who) {
  const ms_frame_desc& d = *desc;
  if (d.projected_input) { set_error("%s: projected input has no per-gaussian stage", who); return MS_ERR_BAD_ARG; }
  if (d.n == 0) return 0;
  ms_frame_layout L;
  frame_layout(desc, &L);
  char* kn = (char*)keep_n;
  if (!(in->position && in->log_scaling && in->rotation && in->alpha_logit && in->T_camera_world && in->projection)) {
    set_error("%s: null gaussian / camera input", who); return MS_ERR_BAD_ARG;
  }
  if (!in->feature) { set_error("%s: feature is null", who); return MS_ERR_BAD_ARG; }
  float* rows = frame_uses_rows(desc) ? (float*)(kn + L.splat_rows) : nullptr;
  if (projection) {
    // (projection alone = ms_frame_project_count: the camera position is then made by ms_frame_map_raster's prepare
    // kernel, next to the K limit)
    if (d.sh_degree >= 0 && colours)
      MS_TRY(ms_camera_position(in->T_camera_world, kn + L.camera_position, d.dtype, stream));
    // (without SH the features ARE the colours: the projection kernel copies them into the rows)
    MS_TRY(project_fwd_launch(in->position, in->log_scaling, in->rotation, in->alpha_logit, in->T_camera_world, in->projection,
                              d.image_w, d.image_h, d.near_plane, d.far_plane, d.blur_cov, d.clamp_margin,
                              d.raster.alpha_threshold, d.n, kn + L.points7, kn + L.depth, nullptr, d.dtype, stream, rows,
                              rows && d.sh_degree < 0 ? in->feature : nullptr));
  }
  if (colours && d.sh_degree >= 0)
    MS_TRY(sh_fwd_inplace_launch(in->feature, in->position, kn + L.depth, kn + L.camera_position, d.n, d.f,
                                 d.sh_degree, kn + L.colours, d.dtype, (hipStream_t)stream, rows));
  return 0;
}

extern "C" int ms_frame_project(const ms_frame_desc* desc, const ms_frame_inputs* in, void* keep_n, void* stream) {
  MS_TRY(check_desc(desc, "ms_frame_project"));
  MS_TRY(check_inputs(in, "ms_frame_project"));
  MS_CHECK_ARG(in && keep_n, "null pointer");
  return frame_project_impl(desc, in, keep_n, true, true, stream, "ms_frame_project");
}

extern "C" int ms_frame_sh_colours(const ms_frame_desc* desc, const ms_frame_inputs* in, void* keep_n, void* stream) {
  MS_TRY(check_desc(desc, "ms_frame_sh_colours"));
  MS_TRY(check_inputs(in, "ms_frame_sh_colours"));
  MS_CHECK_ARG(in && keep_n, "null pointer");
  const ms_frame_desc& d = *desc;
  MS_CHECK_ARG(!d.projected_input && d.sh_degree >= 0, "a frame that evaluates SH colours itself");
  MS_CHECK_ARG(!frame_uses_rows(desc), "the splat-row table is filled on the frame's own stream");
  if (d.n == 0) return 0;
  ms_frame_layout L;
  frame_layout(desc, &L);
  MS_CHECK_ARG(in->T_camera_world != nullptr, "T_camera_world is null");
  MS_TRY(ms_camera_position(in->T_camera_world, (char*)keep_n + L.camera_position, d.dtype, stream));
  return frame_project_impl(desc, in, keep_n, false, true, stream, "ms_frame_sh_colours");
}

extern "C" int ms_frame_project_count(const ms_frame_desc* desc, const ms_frame_inputs* in, void* keep_n,
                                      void* scratch_n, int32_t* k_host, void* k_event, void* stream) {
  MS_TRY(check_desc(desc, "ms_frame_project_count"));
  MS_TRY(check_inputs(in, "ms_frame_project_count"));
  MS_CHECK_ARG(in && keep_n && scratch_n, "null pointer");
  const ms_frame_desc& d = *desc;
  const FrameGeom g = frame_geom(desc);
  ms_frame_layout L;
  frame_layout(desc, &L);
  hipStream_t s = (hipStream_t)stream;
  char* kn = (char*)keep_n;
  char* sn = (char*)scratch_n;
  int32_t* counters = (int32_t*)(kn + L.counters);

  if (d.n == 0) {
    MS_CHECK_HIP(hipMemsetAsync(counters, 0, 8 * sizeof(int32_t), s));
    if (k_host) *k_host = 0;
    if (k_event) MS_CHECK_HIP(hipEventRecord((hipEvent_t)k_event, s));
    return 0;
  }

  const void* points7;
  const void* depth;
  if (!d.projected_input) {
    // projection only: the SH colours are not needed before the raster forward and are evaluated by
    // ms_frame_map_raster — AFTER the kernels that produce K, so that a host that looks at K (eager mode) finds the
    // long SH pass still queued behind it instead of in front of it
    MS_TRY(frame_project_impl(desc, in, keep_n, true, false, stream, "ms_frame_project_count"));
    points7 = kn + L.points7;
    depth = kn + L.depth;
  } else {
    MS_CHECK_ARG(in->points7 && in->depth && in->colours, "null projected input");
    points7 = in->points7;
    depth = in->depth;
  }

  // the overlap test runs in float32 like the reference (Gaussian2D from taichi_lib.f32)
  const float* points_f32 = (const float*)points7;
  if (d.dtype == MS_F64) {
    float* copy = (float*)(sn + L.points7_f32);
    f64_to_f32_kernel<<<dim3((unsigned)div_up(d.n * 7, 256)), dim3(256), 0, s>>>((const double*)points7, copy, d.n * 7);
    points_f32 = copy;
  }

  uint32_t* sorted_keys = (uint32_t*)(sn + L.sorted_keys);
  int32_t* order = (int32_t*)(sn + L.order);
  int32_t* counts = (int32_t*)(sn + L.counts);
  int32_t* cum = (int32_t*)(sn + L.cum);
  const int cull = d.projected_input ? 0 : 1;
  if (mapper_presort(d)) {
    // ndc depth (renderer.py:67) is fused into the key generation when a near plane is given
    depth_argsort_launch(depth, d.n, d.depth16, d.near_plane > 0.0 ? d.near_plane : 0.0, d.far_plane, d.dtype, cull,
                         sorted_keys, order, sn + L.tmp_n, s);
    tile_count_launch(points_f32, order, cull ? sorted_keys : nullptr, d.n, g.w_pad, g.h_pad, d.raster.tile_size,
                      (float)d.raster.alpha_threshold, g.row_begin, g.row_end, counts, (float*)(sn + L.ordered_points), s);
  } else {
    tile_count_direct_launch(points_f32, cull ? depth : nullptr, d.dtype, d.n, g.w_pad, g.h_pad, d.raster.tile_size,
                             (float)d.raster.alpha_threshold, g.row_begin, g.row_end, counts, s);
  }
  exclusive_scan_launch(counts, d.n, cum, k_host, sn + L.tmp_n, s, counters);      // K -> counters[0] and *k_host
  MS_CHECK_LAUNCH();
  if (k_event) MS_CHECK_HIP(hipEventRecord((hipEvent_t)k_event, s));
  return 0;
}

extern "C" int ms_frame_map_raster(const ms_frame_desc* desc, const ms_frame_inputs* in, void* keep_n, void* scratch_n,
                                   void* keep_k, void* scratch_k, void* out_image, void* out_alpha,
                                   void* out_visibility, void* stream) {
  MS_TRY(check_desc(desc, "ms_frame_map_raster"));
  MS_TRY(check_inputs(in, "ms_frame_map_raster"));
  MS_CHECK_ARG(in && keep_n && scratch_n && out_image && out_alpha, "null pointer");
  const ms_frame_desc& d = *desc;
  const FrameGeom g = frame_geom(desc);
  ms_frame_layout L;
  frame_layout(desc, &L);
  hipStream_t s = (hipStream_t)stream;
  char* kn = (char*)keep_n;
  char* sn = (char*)scratch_n;
  char* kk = (char*)keep_k;
  char* sk = (char*)scratch_k;
  int32_t* counters = (int32_t*)(kn + L.counters);
  int32_t* ranges = (int32_t*)(kn + L.tile_ranges);
  int32_t* o2p = (int32_t*)(kk + L.overlap_to_point);
  // a frame whose SH colours are evaluated by ms_frame_sh_colours on ANOTHER stream, beside the mapper's launches: this
  // call then neither evaluates them nor makes the camera position, and waits for the event in front of the raster forward
  const bool colours_elsewhere = !d.projected_input && d.sh_degree >= 0 && in->colours_ready_event != nullptr;

  {
    const bool want_cam = !d.projected_input && d.sh_degree >= 0 && d.n > 0 && !colours_elsewhere;
    if (want_cam) MS_CHECK_ARG(in->T_camera_world != nullptr, "T_camera_world is null");
    const int64_t words = (int64_t)g.num_tiles * 2;
    const dim3 grid((unsigned)(words > 0 ? div_up(words, 256) : 1)), block(256);
    if (d.dtype == MS_F64)
      frame_prepare_kernel<double><<<grid, block, 0, s>>>(counters, (int32_t)d.k_capacity, (const double*)in->T_camera_world,
                                                          want_cam ? (double*)(kn + L.camera_position) : nullptr, ranges, words);
    else
      frame_prepare_kernel<float><<<grid, block, 0, s>>>(counters, (int32_t)d.k_capacity, (const float*)in->T_camera_world,
                                                         want_cam ? (float*)(kn + L.camera_position) : nullptr, ranges, words);
  }
  if (!d.projected_input && !colours_elsewhere) MS_TRY(frame_project_impl(desc, in, keep_n, false, true, stream, "ms_frame_map_raster"));
  if (d.n > 0 && d.k_capacity > 0) {
    MS_CHECK_ARG(keep_k && scratch_k, "null overlap buffers");
    int32_t* values = (int32_t*)(sk + L.values);
    if (mapper_presort(d)) {
      uint32_t* keys = (uint32_t*)(sk + L.keys);
      uint32_t* keys_sorted = (uint32_t*)(sk + L.keys_sorted);
      tile_emit_ordered_launch((const float*)(sn + L.ordered_points), (const int32_t*)(sn + L.order),
                               (const int32_t*)(sn + L.cum), d.n, g.w_pad, g.h_pad, d.raster.tile_size,
                               (float)d.raster.alpha_threshold, g.row_begin, g.row_end, counters + 1, keys, values, s);
      sort_pairs_u32_dev_launch(keys, values, keys_sorted, o2p, d.k_capacity, counters + 1, g.tile_bits, sk + L.tmp_k, s);
      MS_TRY(find_ranges_dev_launch(keys_sorted, d.k_capacity, counters + 1, g.num_tiles, ranges, s, true));
    } else {
      uint64_t* keys = (uint64_t*)(sk + L.keys);
      uint64_t* keys_sorted = (uint64_t*)(sk + L.keys_sorted);
      const void* depth = d.projected_input ? in->depth : (const void*)(kn + L.depth);
      const float* points_f32 = d.dtype == MS_F64 ? (const float*)(sn + L.points7_f32)
                                                  : (const float*)(d.projected_input ? in->points7 : (const void*)(kn + L.points7));
      tile_emit_direct_launch(points_f32, depth, d.dtype, (const int32_t*)(sn + L.cum), d.n, g.w_pad, g.h_pad,
                              d.raster.tile_size, (float)d.raster.alpha_threshold, g.row_begin, g.row_end, d.depth16,
                              d.near_plane > 0.0 ? d.near_plane : 0.0, d.far_plane, counters + 1, keys, values, s);
      sort_pairs_u64_dev_launch(keys, values, keys_sorted, o2p, d.k_capacity, counters + 1, 32, 32 + g.tile_bits,
                                sk + L.tmp_k, s);
      MS_TRY(find_ranges_u64_dev_launch(keys_sorted, d.k_capacity, counters + 1, g.num_tiles, ranges, s));
      // only the strip's tile rows have runs (a rank of 8 would launch 57 344 workgroups that find an empty range)
      const int64_t first_tile = (int64_t)g.row_begin * g.tiles_wide;
      const int64_t strip_tiles = (int64_t)(g.row_end > g.row_begin ? g.row_end - g.row_begin : 0) * g.tiles_wide;
      tile_depth_sort_launch(ranges + 2 * first_tile, strip_tiles, keys_sorted, o2p, keys, s, counters + 3, in->longest_run_host);
    }
  }
  MS_CHECK_LAUNCH();

  const void* points7 = d.projected_input ? in->points7 : (const void*)(kn + L.points7);
  const void* colours = d.sh_degree >= 0 ? (const void*)(kn + L.colours) : (d.projected_input ? in->colours : in->feature);
  MS_CHECK_ARG(d.n == 0 || colours != nullptr, "colours are null");
  SplitScratch split{};
  const bool cut = frame_uses_split(desc) && keep_k != nullptr;
  if (cut) split = split_scratch_carve(kk + L.split_scratch, d.k_capacity, d.raster.tile_size, frame_split_params(desc));
  // the colours of a projected-input frame may still be arriving (a rank step's second forward collective)
  if (in->colours_ready_event) MS_CHECK_HIP(hipStreamWaitEvent(s, (hipEvent_t)in->colours_ready_event, 0));
  return raster_fwd_launch(points7, colours, frame_uses_rows(desc) ? (const float*)(kn + L.splat_rows) : nullptr, ranges, o2p,
                           d.image_w, d.image_h, d.f, &d.raster, out_image, out_alpha, out_visibility, g.row_begin, g.row_end,
                           d.dtype, stream, cut ? &split : nullptr, in->longest_run_host);
}

extern "C" int ms_frame_backward(const ms_frame_desc* desc, const ms_frame_inputs* in, void* keep_n, void* keep_k,
                                 const ms_frame_grads* gr, void* stream) {
  MS_TRY(check_desc(desc, "ms_frame_backward"));
  MS_TRY(check_inputs(in, "ms_frame_backward"));
  MS_TRY(check_grads(gr, "ms_frame_backward"));
  MS_CHECK_ARG(in && keep_n && gr, "null pointer");
  MS_CHECK_ARG(gr->stage == MS_BACKWARD_GAUSSIANS || (gr->image && gr->grad_image), "null image / grad_image");
  const ms_frame_desc& d = *desc;
  const FrameGeom g = frame_geom(desc);
  ms_frame_layout L;
  frame_layout(desc, &L);
  hipStream_t s = (hipStream_t)stream;
  char* kn = (char*)keep_n;
  char* kk = (char*)keep_k;
  const int32_t* ranges = (const int32_t*)(kn + L.tile_ranges);
  const int32_t* o2p = (const int32_t*)(kk + L.overlap_to_point);
  const void* points7 = d.projected_input ? in->points7 : (const void*)(kn + L.points7);
  const void* depth = d.projected_input ? in->depth : (const void*)(kn + L.depth);
  const void* colours = d.sh_degree >= 0 ? (const void*)(kn + L.colours) : (d.projected_input ? in->colours : in->feature);
  if (d.n == 0) return 0;

  // MS_BACKWARD_GAUSSIANS: the 2D-boundary gradients in grad_points7 / grad_colours come from elsewhere (multi-GPU:
  // summed over the strips and sent home) — only the per-gaussian pass runs.  MS_BACKWARD_RASTER: the other half
  const bool given = gr->stage == MS_BACKWARD_GAUSSIANS;
  const bool raster_only = gr->stage == MS_BACKWARD_RASTER || d.projected_input;
  const bool moments = !given && frame_uses_moments(desc, gr->deterministic);
  MS_CHECK_ARG(!gr->grad_image_broadcast || (!given && moments), "grad_image_broadcast: moments path only (float32 RGB, plain pdf)");
  MS_CHECK_ARG(gr->boundary_form == MS_BOUNDARY_AXIS_SIGMA || gr->boundary_form == MS_BOUNDARY_COVARIANCE, "boundary_form");
  MS_CHECK_ARG(gr->boundary_form == 0 || given || (moments && gr->stage == MS_BACKWARD_RASTER),
               "MS_BOUNDARY_COVARIANCE rows: MS_BACKWARD_GAUSSIANS, or MS_BACKWARD_RASTER on the moments path");
  if (gr->boundary_stride != 0) {
    MS_CHECK_ARG(gr->boundary_stride >= 7 + d.f && d.dtype == MS_F32, "boundary_stride: rows of >= 7 + f floats, float32 frames");
    MS_CHECK_ARG(given || (moments && raster_only), "boundary_stride: MS_BACKWARD_GAUSSIANS, or MS_BACKWARD_RASTER on the moments path");
  }
  if (given) {
    MS_CHECK_ARG(!d.projected_input, "projected input has no per-gaussian backward");
    if (gr->gather_world > 0)
      MS_CHECK_ARG(gr->gather_rows && gr->gather_slots && gr->gather_route && gr->boundary_stride >= 7 + d.f && d.dtype == MS_F32,
                   "gathered boundary gradients: rows, slots, route, boundary_stride >= 7 + f, float32");
    else
      MS_CHECK_ARG(gr->grad_points7 != nullptr, "grad_points7 is null");
  } else if (moments) {
    MS_CHECK_ARG(gr->moments != nullptr, "moments is null");
    // the segments of long tile runs start from the states the forward of THIS frame left in keep_k
    SplitScratch split{};
    const bool cut = frame_uses_split(desc) && keep_k != nullptr;
    if (cut) split = split_scratch_carve(kk + L.split_scratch, d.k_capacity, d.raster.tile_size, frame_split_params(desc));
    MS_TRY(raster_bwd_moments_launch(points7, colours, ranges, o2p, gr->image, gr->grad_image, d.image_w, d.image_h, &d.raster,
                                     (float*)gr->moments, gr->deterministic, gr->fixed_exp, g.row_begin, g.row_end,
                                     gr->grad_image_broadcast, s,
                                     frame_uses_rows(desc) ? (const float*)(kn + L.splat_rows) : nullptr,
                                     cut ? &split : nullptr));
    if (raster_only)
      return moments_finalize_rezero_launch((const float*)points7, (float*)gr->moments, gr->deterministic, gr->fixed_exp,
                                            d.n, (float*)gr->grad_points7, (float*)gr->grad_colours,
                                            d.raster.compute_point_heuristic ? (float*)gr->point_heuristic : nullptr, s,
                                            gr->boundary_stride, d.projected_input && gr->stage != MS_BACKWARD_RASTER ? 0 : gr->boundary_form);
  } else {
    MS_CHECK_ARG(gr->grad_points7 || gr->grad_colours, "no gradient accumulator");
    MS_TRY(ms_raster_bwd(points7, colours, ranges, o2p, gr->image, gr->grad_image, d.image_w, d.image_h, d.f, &d.raster,
                         gr->grad_points7, gr->grad_colours, d.raster.compute_point_heuristic ? gr->point_heuristic : nullptr,
                         g.row_begin, g.row_end, d.dtype, stream));
    if (raster_only) return 0;
  }

  GaussianBwdArgs a{};
  a.dtype = d.dtype;
  a.n = d.n;
  a.position = in->position; a.log_scaling = in->log_scaling; a.rotation = in->rotation; a.alpha_logit = in->alpha_logit;
  a.T_camera_world = in->T_camera_world; a.projection = in->projection;
  a.image_w = d.image_w; a.image_h = d.image_h;
  a.blur_cov = d.blur_cov; a.clamp_margin = d.clamp_margin;
  a.depth = depth;
  if (moments) {
    a.moments = gr->moments; a.deterministic = gr->deterministic; a.fixed_exp = gr->fixed_exp;
    a.store_points7 = gr->grad_points7; a.store_colours = gr->grad_colours;
    a.point_heuristic = d.raster.compute_point_heuristic ? gr->point_heuristic : nullptr;
    a.point_visibility = d.raster.compute_point_heuristic ? gr->point_visibility : nullptr;
  } else {
    a.grad_points7 = gr->grad_points7; a.grad_colours = gr->grad_colours;
    a.boundary_stride = gr->boundary_stride;
    a.boundary_cov = given ? gr->boundary_form : 0;
    if (given && gr->gather_world > 0) {
      a.gather_world = gr->gather_world; a.gather_rows = gr->gather_rows;
      a.gather_slots = gr->gather_slots; a.gather_route = gr->gather_route;
    }
  }
  a.extra_points7 = gr->extra_points7; a.extra_depth = gr->extra_depth; a.extra_colours = gr->extra_colours;
  a.sh_degree = d.sh_degree; a.f = d.f;
  a.camera_position = kn + L.camera_position; a.colours = colours;
  a.grad_position = gr->grad_position; a.grad_log_scaling = gr->grad_log_scaling; a.grad_rotation = gr->grad_rotation;
  a.grad_alpha_logit = gr->grad_alpha_logit; a.grad_feature = gr->grad_feature; a.grad_camera = gr->grad_camera;
  return gaussian_bwd_launch(a, s);
}
