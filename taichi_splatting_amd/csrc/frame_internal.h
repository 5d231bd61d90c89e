// frame_internal.h — C++ launchers shared between translation units (not part of the C-ABI): the frame executor
// (frame.hip) strings the stages of a frame together without going back to the host, so it needs the stage
// launchers with DEVICE-SIDE counts: grids are sized for a capacity and every kernel reads the live count from
// memory (`*n_dev`), exits early above it, and never writes past the capacity.
#pragma once
#include "common.h"

namespace ms {

// key of a culled gaussian in the depth pre-sort (frame executor: no compaction, culled gaussians stay in place):
// sorts behind every real depth (float bits of a non-negative depth are < 0x7f800000) and tells the overlap
// count / emit kernels to skip the row
constexpr uint32_t CULLED_DEPTH_KEY = 0xffffffffu;

// Translation column of inverse(T_camera_world) (reference perspective/params.py:62-65 computes
// torch.inverse(T)[0:3, 3]): Gauss-Jordan with partial pivoting in double, one thread.  Shared by ms_camera_position
// (projection.hip) and the frame executor's prepare kernel (frame.hip); contraction is off inside so that both
// translation units round identically.
template <typename T>
__device__ inline void camera_position_solve(const T* __restrict__ m, T* __restrict__ out) {
#pragma clang fp contract(off)
  double a[4][5];
  for (int r = 0; r < 4; ++r) {
    for (int c = 0; c < 4; ++c) a[r][c] = (double)m[r * 4 + c];
    a[r][4] = r == 3 ? 1.0 : 0.0;                     // solve T x = e_3: x = 4th column of the inverse
  }
  for (int col = 0; col < 4; ++col) {
    int piv = col;
    for (int r = col + 1; r < 4; ++r)
      if (fabs(a[r][col]) > fabs(a[piv][col])) piv = r;
    for (int c = 0; c < 5; ++c) { const double t = a[col][c]; a[col][c] = a[piv][c]; a[piv][c] = t; }
    const double inv = 1.0 / a[col][col];
    for (int c = 0; c < 5; ++c) a[col][c] *= inv;
    for (int r = 0; r < 4; ++r) {
      if (r == col) continue;
      const double fct = a[r][col];
      for (int c = 0; c < 5; ++c) a[r][c] -= fct * a[col][c];
    }
  }
  for (int k = 0; k < 3; ++k) out[k] = (T)a[k][4];
}

// ---- scan_sort.hip ----------------------------------------------------------------------------------------------
size_t scan_tmp_size(int64_t n);
size_t sort_tmp_size(int64_t n, int key_bytes, bool adaptive = false);   // adaptive: depth_argsort_launch's layout
// exclusive scan of n int32 into out[0..n] (out[n] = total); the total also goes to *total_host (pinned) and
// *total_copy (device) when given
void exclusive_scan_launch(const int32_t* in, int64_t n, int32_t* out, int32_t* total_host, void* tmp, hipStream_t s,
                           int32_t* total_copy = nullptr);
// ms_depth_argsort with `cull`: depth <= 0 marks a culled gaussian (key CULLED_DEPTH_KEY)
void depth_argsort_launch(const void* depth, int64_t n, int depth16, double ndc_near, double ndc_far, int dtype,
                          int cull, uint32_t* out_sorted_keys, int32_t* out_order, char* tmp, hipStream_t s);
// stable radix sort of u32 keys / int32 values on bits [0, end_bit): `capacity` sizes the grid and the scratch,
// the live count is *n_dev (<= capacity)
void sort_pairs_u32_dev_launch(const uint32_t* keys_in, const int32_t* vals_in, uint32_t* keys_out, int32_t* vals_out,
                               int64_t capacity, const int32_t* n_dev, int end_bit, char* tmp, hipStream_t s);
// per-tile [first, last + 1) ranges of the sorted tile ids; writes every entry (zero fill included unless the caller
// has `zeroed` the array on the stream already)
int find_ranges_dev_launch(const uint32_t* sorted_keys, int64_t capacity, const int32_t* k_dev, int64_t num_tiles,
                           int32_t* out_ranges, hipStream_t s, bool zeroed = false);

// the same on u64 keys, bits [begin_bit, end_bit) (direct-order mapper: tile id in bits 32.., depth key below)
void sort_pairs_u64_dev_launch(const uint64_t* keys_in, const int32_t* vals_in, uint64_t* keys_out, int32_t* vals_out,
                               int64_t capacity, const int32_t* n_dev, int begin_bit, int end_bit, char* tmp, hipStream_t s);
// ranges of keys >> 32; the caller has zeroed out_ranges on the stream
int find_ranges_u64_dev_launch(const uint64_t* sorted_keys, int64_t capacity, const int32_t* k_dev, int64_t num_tiles,
                               int32_t* out_ranges, hipStream_t s);

// ---- mapper.hip -------------------------------------------------------------------------------------------------
void tile_count_launch(const float* points7, const int32_t* order, const uint32_t* cull_keys, int64_t v, int image_w,
                       int image_h, int tile_size, float alpha_threshold, int row_begin, int row_end,
                       int32_t* out_counts, float* out_ordered, hipStream_t s);
// key_mode 2 emission in depth order from the ordered copy; nothing is written when *k_limit_dev == 0 (overflow of
// the caller's capacity: see frame.hip)
void tile_emit_ordered_launch(const float* ordered_points7, const int32_t* order, const int32_t* cum, int64_t v,
                              int image_w, int image_h, int tile_size, float alpha_threshold, int row_begin,
                              int row_end, const int32_t* k_limit_dev, uint32_t* out_keys, int32_t* out_values,
                              hipStream_t s);

// direct-order mapper of the frame executor (no depth pre-sort): overlap counts in storage order (cull_depth: rows
// with depth <= 0 overlap nothing), then u64 keys tile << 32 | depth_sort_key and the point index as value
void tile_count_direct_launch(const float* points7, const void* cull_depth, int dtype, int64_t v, int image_w, int image_h,
                              int tile_size, float alpha_threshold, int row_begin, int row_end, int32_t* out_counts,
                              hipStream_t s);
void tile_emit_direct_launch(const float* points7, const void* depth, int dtype, const int32_t* cum, int64_t v, int image_w,
                             int image_h, int tile_size, float alpha_threshold, int row_begin, int row_end, int depth16,
                             double ndc_near, double ndc_far, const int32_t* k_limit_dev, uint64_t* out_keys,
                             int32_t* out_values, hipStream_t s);

// ---- tile_sort.hip ----------------------------------------------------------------------------------------------
// Sorts every tile's run of `sorted_keys` (tile << 32 | depth key, already grouped by tile with the point indices in
// `overlap_to_point` ascending inside each run) by (depth key, point index); only overlap_to_point is rewritten.
// `scratch`: K u64 words the large-tile path may use (the unsorted key buffer is free by then).
// run_stats (device, three zeroed words, or NULL): [2] = "a per-tile kernel declined a run" (NULL: the long-run kernel
// reads every run's mark; [0], [1] unused).  run_host (pinned, may be NULL) receives the length of a run above 16384
// entries if there is one.
void tile_depth_sort_launch(const int32_t* tile_ranges, int64_t num_tiles, uint64_t* sorted_keys, int32_t* overlap_to_point,
                            uint64_t* scratch, hipStream_t s, int32_t* run_stats = nullptr, int32_t* run_host = nullptr);

// ---- projection.hip ---------------------------------------------------------------------------------------------
// ms_project_fwd; splat_rows (float32): also fills words 0..7 of each gaussian's splat row (common.h) and, when colours_in
// (n x 3) is given, words 8..11
int project_fwd_launch(const void* position, const void* log_scaling, const void* rotation, const void* alpha_logit,
                       const void* T_camera_world, const void* projection, int image_w, int image_h, double near_plane,
                       double far_plane, double blur_cov, double clamp_margin, double alpha_threshold, int64_t n,
                       void* out_points7, void* out_depth, int32_t* out_flag, int dtype, void* stream, float* splat_rows,
                       const void* colours_in);

// ---- sh.hip -----------------------------------------------------------------------------------------------------
// SH colours of ALL n gaussians in place (identity index list); rows with depth[i] <= 0 (culled) get zeros and
// their 4 F D parameter bytes are not read
// splat_rows (float32 RGB only): the colours also go into words 8..10 of each gaussian's splat row (raster_common.h)
int sh_fwd_inplace_launch(const void* params, const void* positions, const void* depth, const void* cam_pos,
                          int64_t n, int f, int degree, void* out, int dtype, hipStream_t s, float* splat_rows = nullptr);

// ---- raster.hip -------------------------------------------------------------------------------------------------
// Splat rows (raster_common.h: SPLAT_ROW floats per gaussian, [points7 | depth | colour | 0]): a side table the frame
// executor's projection and SH kernels fill for the product raster kernels, which then gather one 128-byte line per
// splat.  The dense arrays stay what every other consumer (mapper, per-gaussian backward, outputs) reads.
bool raster_uses_splat_rows(const ms_raster_config* cfg, int f, int dtype);
// ms_raster_fwd; splat_rows != NULL: the product kernels read the table instead of points7 / features
int raster_fwd_launch(const void* points7, const void* features, const float* splat_rows, const int32_t* tile_ranges,
                      const int32_t* overlap_to_point, int image_w, int image_h, int f, const ms_raster_config* cfg,
                      void* out_image, void* out_alpha, void* out_visibility, int tile_row_begin, int tile_row_end,
                      int dtype, void* stream, const struct SplitScratch* split = nullptr, int32_t* long_run_word = nullptr);

// ---- raster_bwd_scan.hip ----------------------------------------------------------------------------------------
// ms_raster_bwd_moments with grad_broadcast: dL/dimage given as ONE pixel's f values (ms_frame_grads.grad_image_broadcast)
int raster_bwd_moments_launch(const void* points7, const void* features, const int32_t* tile_ranges,
                              const int32_t* overlap_to_point, const void* image, const void* grad_image, int image_w,
                              int image_h, const ms_raster_config* cfg, float* moments, int deterministic,
                              const int32_t* fixed_exp, int tile_row_begin, int tile_row_end, int grad_broadcast,
                              hipStream_t s, const float* splat_rows = nullptr, const struct SplitScratch* split = nullptr);
// ms_raster_moments_finalize that also clears the rows it reads (persistent moments buffer)
// row_stride > 0: grad_points7 / grad_features are columns of one row-major array with that many floats per row
int moments_finalize_rezero_launch(const float* points7, float* moments, int deterministic, const int32_t* fixed_exp,
                                   int64_t n, float* grad_points7, float* grad_features, float* point_heuristic,
                                   hipStream_t s, int row_stride = 0, int covariance_form = 0);

// ---- tools/experiments/raster_bwd_rows.hip (only with -DMS_WITH_ROWS_KERNEL, tools/build_variant.sh) ------------
// the round-4 experiment (2x2 quad lists packed by DPP rows): tile 8 / 16; false = not served (tile 32, or
// MS_RASTER_BWD=scan), the caller launches raster_bwd_scan_kernel
struct FastParams;
bool launch_rows_backward(const float* points7, const float* features, const int32_t* tile_ranges,
                          const int32_t* overlap_to_point, const float* image, const float* grad_image,
                          const FastParams& rp, int tile_size, bool heuristics, float* moments, const int32_t* fixed_exp,
                          hipStream_t s);

// ---- gaussian_bwd.hip -------------------------------------------------------------------------------------------
// One pass over the gaussians for the whole per-gaussian backward of a frame: 2D-boundary gradients (from the raster
// backward's moment rows, which it re-zeroes, or from gradient arrays) -> projection backward -> SH backward.
struct GaussianBwdArgs {
  int dtype;                 // MS_F32 / MS_F64
  int64_t n;
  const void *position, *log_scaling, *rotation, *alpha_logit, *T_camera_world, *projection;
  int image_w, image_h;
  double blur_cov, clamp_margin;
  const void* depth;         // (n) forward depth, <= 0: culled (every gradient row of it is zero)
  // source of d(packed 2D gaussian), d(colour): moments (n, MS_MOMENT_ROW) float (int64 when deterministic), or arrays
  void* moments;
  int deterministic;
  const int32_t* fixed_exp;  // deterministic: binary exponents of the fixed-point scales (raster_bwd_scan.hip)
  const void* grad_points7;  // (n, 7) or NULL
  const void* grad_colours;  // (n, f) or NULL
  int boundary_cov;          // rows hold [d mean | dL/d(a, b, c) | 0 | d alpha] (MS_BOUNDARY_COVARIANCE) instead of d(packed 2D)
  int boundary_stride;       // 0: the two arrays above are dense; > 0: floats per row of the array both are columns of
  // gather_world > 0: d(packed 2D gaussian), d(colour) of gaussian i = sum of rows gather_slots[i * gather_world + c],
  // c < gather_route[i] >> 16, of gather_rows (boundary_stride floats per row)
  int gather_world;
  const void* gather_rows;
  const int32_t *gather_slots, *gather_route;
  // gradients arriving at the frame's own per-gaussian outputs (loss terms on gaussians2d / depth / colours)
  const void *extra_points7, *extra_depth, *extra_colours;
  // SH (degree >= 0): d(colour) -> d(sh params) through the clamp mask of the forward colours
  int sh_degree, f;
  const void *camera_position, *colours;
  // outputs (each may be NULL)
  void *grad_position, *grad_log_scaling, *grad_rotation, *grad_alpha_logit;
  void* grad_feature;        // (n, f, D) for SH, (n, f) for plain colours (moments source only)
  void* grad_camera;         // 16 values, accumulated
  void *store_points7, *store_colours;   // the summed 2D-boundary gradients (gaussians2d.grad / features.grad)
  void* point_heuristic;     // (n, 2) from the moment rows
  void* point_visibility;    // (n,) from column 11 of the moment rows (raster backward run with heuristics), or NULL
};
int gaussian_bwd_launch(const GaussianBwdArgs& a, hipStream_t s);

}  // namespace ms
