/* mi355_splat.h — C-ABI of libmi355_splat.so: the gfx950 (MI355X) back end of the
 * taichi_splatting render path (projection -> SH colour -> tile mapper -> alpha composite).
 *
 * Conventions
 *   - every function returns 0 on success, a positive hipError_t on a HIP failure or a negative
 *     MS_ERR_* code on an argument error; ms_last_error_string() describes the last failure.
 *   - all pointers are DEVICE pointers unless the name ends in _host; tensors are contiguous,
 *     row-major; image_size is (W, H), images are (H, W, C).
 *   - no function allocates device memory or synchronises (callers own every buffer, including
 *     scratch: functions taking (tmp, tmp_bytes) report the required size in *tmp_bytes when
 *     tmp == NULL and do nothing else).
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *   - dtype selects the float (MS_F32) or double (MS_F64) instantiation, the way the reference
 *     specialises its Taichi kernels on the tensor dtype (rasterizer/function.py:126).
 *
 * Each entry point names the reference interface it replaces (paths relative to
 * /root/reference/taichi_splatting).
 */
#ifndef MI355_SPLAT_H
#define MI355_SPLAT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 0.5.0 (round 6).  The hundreds digit is the ABI generation: it changes whenever a struct below changes layout or a
 * function changes its signature.  500: ms_frame_desc / ms_frame_inputs / ms_frame_grads start with their own size (and
 * ms_frame_desc with the caller's MS_VERSION) — every ms_frame_* call rejects a struct of another size or generation
 * with MS_ERR_ABI instead of reading fields that are not where it expects them (round 5 grew ms_frame_desc and
 * ms_frame_layout under version 400); split_min_run / split_seg_len parameters of the ms_raster_*_split functions,
 * ms_frame_desc.split_seg_len; ms_optim_* (fused optimiser groups). */
#define MS_VERSION 500
#define MS_ABI_GENERATION(version) ((version) / 100)

enum { MS_F32 = 0, MS_F64 = 1 };

enum {
  MS_ERR_BAD_ARG = -1,      /* null pointer / negative size / unsupported enum */
  MS_ERR_UNSUPPORTED = -2,  /* valid request with no compiled instantiation (e.g. F > 4) */
  MS_ERR_TMP_TOO_SMALL = -3,
  MS_ERR_ABI = -4           /* a struct's leading struct_size / abi_version is not this library's */
};

/* Compile-time constants of the reference's RasterConfig (data_types.py:17-47) that the raster
 * kernels consume at run time. */
typedef struct ms_raster_config {
  int32_t tile_size;               /* 8, 16 or 32 */
  int32_t antialias;               /* gaussian_pdf_antialias instead of gaussian_pdf */
  int32_t use_alpha_blending;      /* 0: quantile ("median") render, forward only */
  int32_t compute_visibility;
  int32_t compute_point_heuristic;
  int32_t reserved;
  double clamp_max_alpha;
  double alpha_threshold;
  double saturate_threshold;
} ms_raster_config;

int ms_version(void);
const char* ms_last_error_string(void);

/* ---- perspective projection ------------------------------------------------------------------
 * ms_project_fwd replaces project_kernel (perspective/projection.py:33-81): per gaussian, packed
 * 2D gaussian [mean2, axis2, sigma2, alpha] into out_points7[i], camera depth into out_depth[i]
 * (0 when culled) and the in-view flag (0/1) into out_flag[i].  T_camera_world is the (4,4)
 * row-major matrix (rows 0..2 used), projection = [fx, fy, cx, cy]; both are device pointers so
 * that no host read-back of camera tensors is needed (the reference replicates them per point,
 * projection.py:215-216). */
int ms_project_fwd(const void* position, const void* log_scaling, const void* rotation,
                   const void* alpha_logit, const void* T_camera_world, const void* projection,
                   int image_w, int image_h, double near_plane, double far_plane,
                   double blur_cov, double clamp_margin, double alpha_threshold,
                   int64_t n, void* out_points7, void* out_depth, int32_t* out_flag,
                   int dtype, void* stream);

/* Compaction replacing torch.nonzero + 2 gathers (projection.py:147-150).  scan = exclusive scan
 * of the flags (n+1 entries, ms_exclusive_scan_i32).  Writes the V visible rows: out_points7
 * (V,7), out_depth (V), optional out_ndc_depth (V) = 1-(1/d-1/far)/(1/near-1/far)
 * (torch_lib/projection.py:120-123, fused) and out_indexes (V) int64. */
int ms_project_gather(const void* points7, const void* depth, const int32_t* flag,
                      const int32_t* scan, int64_t n, double near_plane, double far_plane,
                      void* out_points7, void* out_depth, void* out_ndc_depth,
                      int64_t* out_indexes, int dtype, void* stream);

/* ms_project_bwd replaces indexed_project_kernel.grad (projection.py:85-119,167-188): hand-derived
 * reverse mode.  Rows indexes[i] of the N-sized outputs are WRITTEN (rows of culled gaussians are
 * left untouched: pre-zero them); grad_camera (16 values: dT[3][4] row-major then d[fx,fy,cx,cy])
 * is ACCUMULATED atomically and may be NULL; grad_depth may be NULL (no gradient reaches the depth output). */
int ms_project_bwd(const void* position, const void* log_scaling, const void* rotation,
                   const void* alpha_logit, const void* T_camera_world, const void* projection,
                   int image_w, int image_h, double blur_cov, double clamp_margin,
                   const int64_t* indexes, int64_t v, const void* grad_points7,
                   const void* grad_depth, void* grad_position, void* grad_log_scaling,
                   void* grad_rotation, void* grad_alpha_logit, void* grad_camera,
                   int dtype, void* stream);

/* ---- spherical harmonics ---------------------------------------------------------------------
 * evaluate_sh_at_kernel (indexed_spherical_harmonics.py:119-134): out[i,c] =
 * clamp(sum_d Y_d(normalise(pos[idx]-cam)) * params[idx,c,d] + 0.5, 0, 1); params is (M, F, D),
 * D = (degree+1)^2, degree in [0,3]. */
int ms_sh_fwd(const void* params, const void* positions, const int64_t* indexes,
              const void* camera_pos, int64_t v, int f, int degree, void* out,
              int dtype, void* stream);

/* evaluate_sh_at_kernel.grad (indexed_spherical_harmonics.py:153-160).  grad_params (M,F,D),
 * grad_positions (M,3) and grad_camera_pos (3) are ACCUMULATED atomically (indexes may repeat);
 * any of them may be NULL.  out = the saved forward output (V,F) or NULL: with it, and when only
 * grad_params is requested (the renderer detaches positions, renderer.py:53), a streaming kernel
 * with coalesced row writes is used; unique_indexes != 0 additionally promises that indexes holds
 * no repeats (true for the projection's compaction list), so rows are WRITTEN with plain stores
 * (rows not listed stay untouched: pre-zero grad_params). */
int ms_sh_bwd(const void* params, const void* positions, const int64_t* indexes,
              const void* camera_pos, int64_t v, int f, int degree, const void* out,
              const void* grad_out, void* grad_params, void* grad_positions,
              void* grad_camera_pos, int unique_indexes, int dtype, void* stream);

/* ---- tile mapper -----------------------------------------------------------------------------
 * ms_tile_count replaces tile_overlaps_kernel (mapper/tile_mapper.py:76-86): counts[j] = number
 * of tiles (for gaussian order[j], or j when order is NULL) of the (image_w x image_h, already padded to the tile size) grid that pass the
 * OBB-vs-tile test, restricted to tile rows [tile_row_begin, tile_row_end) (multi-GPU strips;
 * pass 0 and a huge value for the whole image).  points7 is float. */
int ms_tile_count(const float* points7, const int32_t* order, int64_t v, int image_w, int image_h,
                  int tile_size, float alpha_threshold, int tile_row_begin, int tile_row_end,
                  int32_t* out_counts, float* out_ordered_points7 /* (v,7) copy in visiting order, or NULL */,
                  void* stream);

/* Depth pre-sort keys (new design, replaces 2/3 of the reference's 48-bit key sort,
 * tile_mapper.py:156): out_keys[i] = float_bits(depth[i]) (or the 16 bit quantisation of
 * make_sort_key, tile_mapper.py:55-61, when depth16 != 0), out_values[i] = i.  Stable-sorting these
 * pairs yields `order`, the depth order with ties by point index; ms_tile_count / ms_tile_emit then
 * visit the gaussians in that order (counts[j] / keys of gaussian order[j]) so that a stable sort on
 * the TILE ID alone (key_mode 2) reproduces the reference's (tile, depth, point) order.
 * ndc_near > 0 applies ndc_depth (torch_lib/projection.py:120-123) to the camera depth first (in
 * double, from the depth's own dtype), fusing renderer.py:67 into the key generation. */
int ms_depth_sort_keys(const void* depth, int64_t v, int depth16, double ndc_near, double ndc_far,
                       uint32_t* out_keys, int32_t* out_values, int dtype, void* stream);

/* ms_depth_sort_keys + ms_radix_sort_pairs (32 bit keys, all 32 or 16 bits) in one call: the first radix pass makes
 * its keys from `depth` on the fly and takes the item index as value, so no key / value arrays exist before the
 * first scatter.  out_order[j] = index of the j-th gaussian in (depth, index) order, out_sorted_keys its key.
 * tmp: call with tmp == NULL for *tmp_bytes (same size as ms_radix_sort_pairs for v 4-byte keys). */
int ms_depth_argsort(const void* depth, int64_t v, int depth16, double ndc_near, double ndc_far, int dtype,
                     uint32_t* out_sorted_keys, int32_t* out_order, void* tmp, size_t* tmp_bytes, void* stream);

/* cuda_lib.full_cumsum (cuda_lib/full_cumsum.cu:17-67): exclusive scan of n int32 into out[0..n],
 * out[n] = total.  If total_host is not NULL it must be pinned, device-visible host memory and
 * receives the total as well (valid after the stream is synchronised). */
int ms_exclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, int32_t* total_host,
                          void* tmp, size_t* tmp_bytes, void* stream);

/* generate_sort_keys_kernel + make_sort_key (tile_mapper.py:36-66,115-146): for every passing
 * tile of gaussian p = order[j] (or j) writes key and value = p at cum[j]++.
 * key_mode 0: u64 (tile_id << 32) | float_bits(depth[p]);  key_mode 1 (use_depth16): u32
 * (tile_id << 16) | u16(clamp(depth[p],0,1)*65535);  key_mode 2: u32 tile_id only (depth may be
 * NULL; requires `order` = depth order for a depth-sorted result).
 * tile_id = tx + ty * (image_w / tile_size) — NOT limited to 16 bits for modes 0 and 2. */
int ms_tile_emit(const float* points7, const float* depth, const int32_t* order, const int32_t* cum,
                 int64_t v, int image_w, int image_h, int tile_size, float alpha_threshold,
                 int tile_row_begin, int tile_row_end, int key_mode,
                 int points_are_ordered /* points7 is ms_tile_count's ordered copy */,
                 void* out_keys, int32_t* out_values, void* stream);

/* The same emission in storage order (no `order`) with the sort key of ms_depth_sort_keys fused in:
 * out_keys[o] = (tile_id << 32) | key32(depth[p]) — float bits of the depth, of its ndc value when ndc_near > 0
 * (torch_lib/projection.py:120-123), or the 16 bit quantisation when depth16 != 0 — and out_values[o] = p.  depth
 * is float or double (depth_dtype).  Feeds the direct-order mapper: ms_radix_sort_pairs on bits
 * [32, 32 + tile bits), ms_find_ranges(key_bytes 8, tile_shift 32), ms_tile_depth_sort. */
int ms_tile_emit_keys64(const float* points7, const void* depth, int depth_dtype, const int32_t* cum, int64_t v,
                        int image_w, int image_h, int tile_size, float alpha_threshold, int tile_row_begin,
                        int tile_row_end, int depth16, double ndc_near, double ndc_far, uint64_t* out_keys,
                        int32_t* out_values, void* stream);

/* cuda_lib.radix_sort_pairs (cuda_lib/radix_sort_pairs.cu:8-70 = cub::DeviceRadixSort::SortPairs):
 * stable LSD radix sort of n (key, int32 value) pairs on key bits [begin_bit, end_bit),
 * out-of-place, inputs preserved.  key_bytes is 4 or 8 (unsigned order). */
int ms_radix_sort_pairs(const void* keys_in, const int32_t* values_in, void* keys_out,
                        int32_t* values_out, int64_t n, int key_bytes, int begin_bit, int end_bit,
                        void* tmp, size_t* tmp_bytes, void* stream);

/* cuda_lib.segmented_sort_pairs (cuda_lib/segmented_sort_pairs.cu:9-73): stable sort of each
 * segment [start_offsets[s], end_offsets[s]) by int32 key, values int32. */
int ms_segmented_sort_pairs(const int32_t* keys_in, const int32_t* values_in, int32_t* keys_out,
                            int32_t* values_out, int64_t n, const int64_t* start_offsets,
                            const int64_t* end_offsets, int64_t num_segments, void* stream);

/* find_ranges_kernel (tile_mapper.py:93-112): out_ranges (num_tiles, 2) int32 = [first, last+1)
 * of each tile id (= key >> tile_shift) in the sorted key list; empty tiles get [0, 0).  The
 * function zero-fills out_ranges itself. */
int ms_find_ranges(const void* sorted_keys, int64_t k, int key_bytes, int tile_shift,
                   int64_t num_tiles, int32_t* out_ranges, void* stream);

/* Per-tile depth sort behind a tile-only sort (the frame executor's mapper, csrc/tile_sort.hip; together with a
 * stable sort of `tile << 32 | depth key` pairs on bits [32, 32 + tile_bits) it replaces the reference's 6-pass sort
 * of the whole 64 bit key, tile_mapper.py:148-170, and gives the same order).  sorted_keys (K) u64, grouped by tile
 * (bits 32..) with a 32 bit depth key below; overlap_to_point (K) the point indices, ascending inside every tile's run
 * — what a stable sort of a storage-order emission leaves; tile_ranges (num_tiles, 2) from
 * ms_find_ranges(key_bytes 8, tile_shift 32).  On return every run of overlap_to_point is in (depth key, point index)
 * order.  sorted_keys is scratch afterwards (long runs are sorted through it); `scratch`: K more u64 words. */
int ms_tile_depth_sort(const int32_t* tile_ranges, int64_t num_tiles, uint64_t* sorted_keys,
                       int32_t* overlap_to_point, uint64_t* scratch, void* stream);

/* ---- rasterizer ------------------------------------------------------------------------------
 * _forward_kernel (rasterizer/forward.py:23-135).  points7 (V,7), features (V,F), tile_ranges
 * (T,2) int32 indexed by tile id = tx + ty * ceil(W/tile), overlap_to_point (K) int32.
 * out_image (H,W,F), out_alpha (H,W), out_visibility (V) (pre-zeroed, used when
 * cfg->compute_visibility) or NULL.  Only tile rows [tile_row_begin, tile_row_end) are rendered. */
int ms_raster_fwd(const void* points7, const void* features, const int32_t* tile_ranges,
                  const int32_t* overlap_to_point, int image_w, int image_h, int f,
                  const ms_raster_config* cfg, void* out_image, void* out_alpha,
                  void* out_visibility, int tile_row_begin, int tile_row_end,
                  int dtype, void* stream);

/* _backward_kernel (rasterizer/backward.py:51-224).  image = forward output, grad_image =
 * dL/dimage.  grad_points7 (V,7), grad_features (V,F) and point_heuristic (V,2) are ACCUMULATED
 * atomically (pre-zero them); each may be NULL to skip that output. */
int ms_raster_bwd(const void* points7, const void* features, const int32_t* tile_ranges,
                  const int32_t* overlap_to_point, const void* image, const void* grad_image,
                  int image_w, int image_h, int f, const ms_raster_config* cfg,
                  void* grad_points7, void* grad_features, void* point_heuristic,
                  int tile_row_begin, int tile_row_end, int dtype, void* stream);

/* Product-path backward (float32, F = 3, plain pdf, alpha blending) in two steps — same semantics as
 * ms_raster_bwd / rasterizer/backward.py:97-224, different organisation (csrc/raster_bwd_scan.hip):
 *
 * ms_raster_bwd_moments ACCUMULATES, per point, one 64-byte row of MS_MOMENT_ROW floats into `moments`
 * (V, MS_MOMENT_ROW), pre-zeroed by the caller, 64-byte aligned:
 *   [0..5]  sum over contributing pixels of q * (1, X', Y', X'^2, X'Y', Y'^2),  q = alpha_pt g dL/dalpha
 *           (backward.py:171-188), (X', Y') = sqrt(log2(e)/2) * the pixel in the splat's normalised frame
 *           (generic.py:311-317)
 *   [6..8]  dL/dfeature = sum w G_c (backward.py:197)
 *   [9..10] with cfg->compute_point_heuristic: sum (dL/dalpha)^2 and sqrt-scaled sum |dL/dmean|_1
 *           (backward.py:190-194)
 *   [11]    with cfg->compute_point_heuristic: sum of the blend weights w = alpha T of the pairs visited (not the ones
 *           backward.py:154 drops behind a pixel's saturation point)
 * Every geometric gradient of gaussian_pdf_with_grad (generic.py:321-336) is a per-splat linear map of these
 * sums; ms_raster_moments_finalize applies it once per point and STORES grad_points7 (V,7), grad_features
 * (V,3) and point_heuristic (V,2) (each may be NULL).
 *
 * deterministic != 0: `moments` is (V, MS_MOMENT_ROW) INT64 (128 bytes per point, pre-zeroed) and the per-patch
 * sums are committed in fixed point with integer atomics, which makes the gradients bitwise reproducible from run
 * to run (float atomics add in arrival order).  The fixed-point unit is chosen per launch from the size of
 * dL/dimage: ms_fixed_point_exponents turns max |grad_image| (one float in device memory, e.g. torch's amax) into
 * the two binary exponents `fixed_exp` (device, int32[2]: sums linear in dL/dimage; the squared prune-cost sum)
 * that both functions then read — no host round trip.  Pass the same flag and exponents to both functions.
 * ms_raster_moments_finalize writes exact zeros for splats that can never blend (alpha or a sigma of 0). */
#define MS_MOMENT_ROW 16
int ms_fixed_point_exponents(const float* amax_dev, int32_t* out_exp2, void* stream);
int ms_raster_bwd_moments(const void* points7, const void* features, const int32_t* tile_ranges,
                          const int32_t* overlap_to_point, const void* image, const void* grad_image,
                          int image_w, int image_h, const ms_raster_config* cfg, float* moments,
                          int deterministic, const int32_t* fixed_exp, int tile_row_begin, int tile_row_end,
                          void* stream);
int ms_raster_moments_finalize(const void* points7, const float* moments, int deterministic,
                               const int32_t* fixed_exp, int64_t n,
                               float* grad_points7, float* grad_features, float* point_heuristic,
                               void* stream);

/* Splat rows (round 5; no counterpart in the reference, whose kernels index Gaussian2D.vec and the feature array
 * separately — rasterizer/forward.py:85-90, backward.py:118-126).  The product raster kernels (float32, F = 3, plain
 * pdf) can gather each splat from ONE table of MS_SPLAT_ROW floats per point, 64-byte aligned:
 *   [0..6] the points7 row, [7] free (the frame executor stores the depth), [8..10] the colour, [11..15] unused
 * — one 128-byte line per gathered splat instead of two or three (config D: backward -6 %, forward -5 %).  A caller
 * packs the table with ms_splat_rows_pack (depth may be NULL; 0.1 ms for 6 M points) and rasterizes it any number of
 * times: it pays from the second pass over the same projected gaussians on.  (The frame executor, ms_frame_*, whose
 * gaussians change every frame, keeps to the dense arrays: filling the table costs it more than the kernels gain;
 * MS_SPLAT_ROWS=1 in the environment makes it do so for measurements.)  Results
 * are those of ms_raster_fwd / ms_raster_bwd_moments on the same values: the forward bit for bit, the moments up to
 * the order of their float atomics (bit for bit with `deterministic`).  ms_raster_moments_finalize is unchanged (it
 * reads points7).  MS_ERR_UNSUPPORTED for configurations the product kernels do not serve (antialias, no alpha
 * blending for the forward). */
/* Long tile runs (round 5; no counterpart in the reference, whose kernels also walk a tile's list with one thread block,
 * rasterizer/forward.py:77-110).  A tile whose run exceeds 16 384 entries (a zoomed-out view piles the scene onto a few
 * tiles) is cut into segments that separate workgroups blend; one pass per such tile composes the segments front to back
 * (the forward blend is affine in (colour, transmittance)) and leaves, per segment and pixel, the state the backward
 * starts from.  ms_raster_fwd_split = ms_raster_fwd for float32 RGB, plain pdf, alpha blending (else MS_ERR_UNSUPPORTED;
 * with cfg->compute_visibility and out_visibility the segments are walked a second time, from their true start), results equal to it up to the rounding of the re-associated products (1e-7 relative).
 * `split_scratch`: ms_raster_split_scratch_bytes(k_capacity, tile_size) bytes, 256-byte aligned, k_capacity >= the number
 * of overlaps; it carries the plan and the start states to ms_raster_bwd_moments_split (= ms_raster_bwd_moments), which
 * must follow the forward on the same lists.  Scenes without long runs pay three near-empty launches (~10 us). */
/* split_min_run / split_seg_len (round 6): a run is cut when it has MORE than split_min_run entries, into segments of at
 * least split_seg_len entries (rounded up to a multiple of 256, at most 256 segments per tile).  0 = the defaults (16 384;
 * 1024 entries, 4096 at tile 32); values below 256 are raised to 256.  The three calls of a (scratch, forward, backward)
 * group must be given the same pair.  If k_capacity is below the real overlap count the plan may not fit its capacities:
 * it is then dropped as a whole (scratch word [2] = 1) and every tile is rendered by its own workgroup — slower, same
 * results. */
size_t ms_raster_split_scratch_bytes(int64_t k_capacity, int tile_size, int split_min_run, int split_seg_len);
int ms_raster_fwd_split(const float* points7, const float* features, const int32_t* tile_ranges,
                        const int32_t* overlap_to_point, int64_t k_capacity, int image_w, int image_h,
                        const ms_raster_config* cfg, float* out_image, float* out_alpha, float* out_visibility,
                        void* split_scratch, int split_min_run, int split_seg_len, int tile_row_begin, int tile_row_end,
                        void* stream);
int ms_raster_bwd_moments_split(const float* points7, const float* features, const int32_t* tile_ranges,
                                const int32_t* overlap_to_point, int64_t k_capacity, const float* image,
                                const float* grad_image, int image_w, int image_h, const ms_raster_config* cfg,
                                float* moments, int deterministic, const int32_t* fixed_exp, const void* split_scratch,
                                int split_min_run, int split_seg_len, int tile_row_begin, int tile_row_end, void* stream);

/* Measurement hook: the NEXT launch of the product raster backward (ms_raster_bwd_moments*, ms_frame_backward on the
 * moments path) records the two hipEvent_t around its per-tile kernel on the stream it is launched on, then disarms.
 * bench.py times the dominant kernel INSIDE whole frames with it (roofline.kernel_ms); NULL, NULL disarms. */
int ms_probe_raster_bwd(void* start_event, void* stop_event);

#define MS_SPLAT_ROW 16
int ms_splat_rows_pack(const float* points7, const float* depth, const float* colours3, int64_t n, float* rows,
                       void* stream);
int ms_raster_fwd_rows(const float* rows, const int32_t* tile_ranges, const int32_t* overlap_to_point, int image_w,
                       int image_h, const ms_raster_config* cfg, float* out_image, float* out_alpha,
                       float* out_visibility, int tile_row_begin, int tile_row_end, void* stream);
int ms_raster_bwd_moments_rows(const float* rows, const int32_t* tile_ranges, const int32_t* overlap_to_point,
                               const float* image, const float* grad_image, int image_w, int image_h,
                               const ms_raster_config* cfg, float* moments, int deterministic,
                               const int32_t* fixed_exp, int tile_row_begin, int tile_row_end, void* stream);

/* ---- frame executor --------------------------------------------------------------------------------------------
 * One frame of render_gaussians (renderer.py:23-108) or rasterize (rasterizer/function.py:133-165) as a FIXED launch
 * sequence without host round trips, so that a frame can be enqueued ahead of the GPU and captured in a hipGraph:
 *
 *   - no compaction: the reference compacts the visible gaussians (torch.nonzero + gathers and a host read of V,
 *     perspective/projection.py:147-150).  Here all n gaussians stay in place; a culled one has depth 0, sorts last
 *     in the depth pre-sort, overlaps no tile and gets zero gradient rows.  The tile order (tile, depth bits, point
 *     index) is unchanged because compaction preserves the index order; overlap_to_point holds indexes into the n
 *     input rows.
 *   - the overlap total K stays on the device (the reference reads it back, cuda_lib/full_cumsum.cu:45-46): the
 *     overlap list has a caller-chosen CAPACITY, the sort / range kernels read the live count from `counters`, and
 *     when K exceeds the capacity nothing is emitted (every tile range empty, image = background) and
 *     counters[2] = 1, so a caller that checks can re-run ms_frame_map_raster with larger buffers.
 *
 * ms_frame_layout reports the sizes of the four caller-owned blocks and the offsets of the arrays inside them:
 *   keep_n / keep_k   per-gaussian / per-overlap arrays that the backward pass (and the caller) read
 *   scratch_n / scratch_k   dead once the forward calls returned (stream order)
 * counters (int32[8] at lay.counters in keep_n): [0] K, [1] live K (0 on overflow), [2] overflow flag.
 *
 * ms_frame_project:        the per-gaussian stage alone (camera position, projection, SH colours) into keep_n — what a
 *     rank of the gaussian-sharded multi-GPU step runs on its shard before the exchange.
 * ms_frame_project_count:  camera position, projection, depth pre-sort, overlap count, scan, K (the SH colours are
 *     evaluated by ms_frame_map_raster: nothing before the raster forward needs them, and a host that looks at K
 *     then finds the long SH pass queued behind the K kernel, not in front of it).
 *     projected_input != 0 (2-D path, splats received for a multi-GPU strip): in->points7 / depth / colours are
 *     used as they are and nothing is culled by depth.  k_host (pinned host int32, may be NULL) receives K;
 *     k_event (hipEvent_t, may be NULL) is recorded right after the kernel that writes it.
 * ms_frame_map_raster:     SH colours, emission, stable sort on the tile id, tile ranges, raster forward.
 * ms_frame_backward:       raster backward + ONE pass over the gaussians (moment rows -> 2D gradients -> projection
 *     backward -> SH backward; csrc/gaussian_bwd.hip).  ms_frame_uses_moments(desc) != 0: the product raster
 *     backward runs and g->moments (n, MS_MOMENT_ROW) must be zero on entry and is zero again on return;
 *     otherwise g->grad_points7 / grad_colours are zero-initialised accumulators of ms_raster_bwd. */
/* Two launch sequences build the same overlap_to_point / tile_ranges (same order, ties included):
 *   MS_MAPPER_DIRECT   overlaps emitted in storage order with keys tile << 32 | depth key, stable sort on the tile bits,
 *                      per-tile depth sort (ms_tile_depth_sort).  92 bytes moved per overlap, nothing per gaussian
 *                      beyond the count / emit passes: the faster one up to about 3.5 overlaps per gaussian.
 *   MS_MAPPER_PRESORT  gaussians sorted by depth first (4 passes over n pairs), overlaps counted and emitted in that
 *                      order, stable sort of (tile id, point) on the tile bits: 52 bytes per overlap.
 * The environment variable MS_MAPPER=direct|presort overrides the field for a whole process (A/B runs). */
#define MS_MAPPER_DIRECT 0
#define MS_MAPPER_PRESORT 1

typedef struct ms_frame_desc {
  uint32_t struct_size;            /* sizeof(ms_frame_desc) of the header the caller was compiled against ... */
  uint32_t abi_version;            /* ... and its MS_VERSION: another size or ABI generation is refused (MS_ERR_ABI) */
  int64_t n;                       /* gaussians = rows of every per-gaussian array */
  int64_t k_capacity;              /* rows of the overlap list */
  int32_t image_w, image_h;
  int32_t dtype;                   /* MS_F32 / MS_F64 */
  int32_t f;                       /* colour channels of the rasterizer (1..4) */
  int32_t sh_degree;               /* -1: `feature` / `colours` are (n, f) colours; 0..3: `feature` is (n, f, (deg+1)^2) */
  int32_t depth16;                 /* use_depth16 sort keys */
  int32_t tile_row_begin, tile_row_end;   /* multi-GPU strip (0, INT32_MAX: whole image) */
  int32_t projected_input;
  int32_t mapper;                  /* MS_MAPPER_DIRECT / MS_MAPPER_PRESORT: the same in every call of a frame */
  int32_t split_long_runs;         /* != 0: long tile runs are cut into segments blended by separate workgroups
                                      (ms_raster_fwd_split; float32 RGB product kernels, ignored otherwise); the same
                                      in every call of a frame.  1: runs above 16 384 entries; > 1: runs above that
                                      many entries (ms_raster_fwd_split's split_min_run).  Costs three near-empty
                                      launches when there is no such run: set it for scene shapes that showed one
                                      (ms_frame_inputs.longest_run_host) */
  int32_t split_seg_len;           /* 0: the default segment length; else ms_raster_fwd_split's split_seg_len */
  double near_plane, far_plane, blur_cov, clamp_margin;
  ms_raster_config raster;
} ms_frame_desc;

typedef struct ms_frame_layout {
  size_t keep_n_bytes, scratch_n_bytes, keep_k_bytes, scratch_k_bytes;
  /* keep_n */
  size_t points7, depth, colours, points7_f32, camera_position, counters, tile_ranges;
  /* scratch_n */
  size_t sorted_keys, order, counts, cum, ordered_points, tmp_n;
  /* keep_k */
  size_t overlap_to_point;
  /* scratch_k */
  size_t keys, values, keys_sorted, tmp_k;
  /* keep_n, behind tile_ranges: the splat-row table (64 bytes per gaussian) of a process started with MS_SPLAT_ROWS=1;
   * 256 unused bytes otherwise */
  size_t splat_rows;
  /* keep_k, behind overlap_to_point: plan and start states of the long-run segments (ms_frame_desc.split_long_runs);
   * 256 unused bytes otherwise */
  size_t split_scratch;
} ms_frame_layout;

typedef struct ms_frame_inputs {
  uint32_t struct_size;            /* sizeof(ms_frame_inputs) */
  uint32_t reserved;
  const void *position, *log_scaling, *rotation, *alpha_logit, *feature, *T_camera_world, *projection;
  const void *points7, *depth, *colours;       /* projected_input */
  /* optional (pinned host int32, device-visible): when a tile's run exceeds 16384 entries, ms_frame_map_raster writes
   * its length there (one of them if there are several; never written otherwise) — the per-tile sort of
   * MS_MAPPER_DIRECT and the product raster forward do.  Such a run is sorted by ONE workgroup at ~12 ns per entry and
   * rasterized by one workgroup: a caller that finds the word set switches the scene shape to MS_MAPPER_PRESORT, whose
   * cost does not depend on how the overlaps are spread over the tiles, and sets split_long_runs
   * (taichi_splatting_amd/frame.py does). */
  int32_t* longest_run_host;
  /* optional hipEvent_t: ms_frame_map_raster makes `stream` wait for it right before the raster forward — the first kernel
   * of the frame that reads `colours` (projected_input frames).  A multi-GPU rank step records it behind the collective
   * that delivers the colours, so the mapper overlaps that transfer.  NULL: no wait. */
  void* colours_ready_event;
} ms_frame_inputs;

enum { MS_BACKWARD_ALL = 0, MS_BACKWARD_GAUSSIANS = 1, MS_BACKWARD_RASTER = 2 };

typedef struct ms_frame_grads {
  uint32_t struct_size;            /* sizeof(ms_frame_grads) */
  uint32_t reserved;
  const void* image;               /* forward image (H, W, f) */
  const void* grad_image;
  const void *extra_points7, *extra_depth, *extra_colours;   /* dL/d(frame's own per-gaussian outputs), may be NULL */
  void* moments;
  int32_t deterministic;
  int32_t stage;                   /* MS_BACKWARD_ALL, or one half of it around a multi-GPU gradient collective:
                                      MS_BACKWARD_RASTER stops after the raster backward and STORES the 2D-boundary
                                      gradients in grad_points7 / grad_colours; MS_BACKWARD_GAUSSIANS runs only the
                                      per-gaussian pass on the gradients GIVEN in those two arrays */
  const int32_t* fixed_exp;
  void *grad_points7, *grad_colours;   /* moments path: optional stores of the summed 2D-boundary gradients */
  void *grad_position, *grad_log_scaling, *grad_rotation, *grad_alpha_logit, *grad_feature, *grad_camera;
  void* point_heuristic;           /* (n, 2), written (moments path) or accumulated (zero-initialised) */
  void* point_visibility;          /* (n,) or NULL.  Moments path with raster.compute_point_heuristic, MS_BACKWARD_ALL: for
                                      every gaussian that passed the projection's culling, WRITES the sum of the blend weights
                                      of the (pixel, splat) pairs the raster backward visited: the visibility of
                                      forward.py:127-128 WITHOUT the pairs behind a pixel's saturation point (backward.py:154
                                      drops them, the forward keeps adding them: at most 1 - saturate_threshold per pixel).
                                      A training loop that accepts that difference can run its forward without
                                      out_visibility.  Ignored on the other paths (ms_frame_uses_moments() tells which runs) */
  int32_t boundary_stride;         /* 0: grad_points7 (n, 7) and grad_colours (n, f) are two dense arrays.  > 0: both are
                                      columns of ONE row-major array with this many floats per row (the return buffer of a
                                      multi-GPU rank step: grad_colours = grad_points7 + 7, stride 7 + f) — float32 frames,
                                      MS_BACKWARD_RASTER on the moments path (stores) and MS_BACKWARD_GAUSSIANS (reads) only */
  int32_t gather_world;            /* MS_BACKWARD_GAUSSIANS, > 0: the 2D-boundary gradient of gaussian i is the SUM of the rows
                                      gather_slots[i * gather_world + c], c < (gather_route[i] >> 16), of gather_rows
                                      (rows of boundary_stride floats: [d packed 2D | d colour]; slot < 0: no row) — the
                                      receive buffer of a rank step's reverse exchange read in place: no return pass, no
                                      home array.  grad_points7 / grad_colours are then unused */
  const void* gather_rows;
  const int32_t* gather_slots;     /* out_slots of ms_strip_route_pack_slots */
  const int32_t* gather_route;     /* out_route of ms_strip_route_count */
  int32_t boundary_form;           /* form of the 2D-boundary gradient rows that MS_BACKWARD_RASTER stores (moments path) and
                                      MS_BACKWARD_GAUSSIANS reads.  0 = MS_BOUNDARY_AXIS_SIGMA: d(packed 2D gaussian) as the
                                      reference's rasterizer returns it, [d mean (2) | d axis (2) | d sigma (2) | d alpha].
                                      1 = MS_BOUNDARY_COVARIANCE: [d mean (2) | dL/d(a, b, c) of the 2D covariance
                                      [[a, b], [b, c]] (3) | 0 | d alpha] — the same 7 columns.  The plain gaussian pdf depends
                                      on (axis, sigma) only through the covariance; handed over in this form the gradient
                                      never passes through the derivative of the eigen-decomposition (no division by
                                      l1 - l2: float32 rows of nearly isotropic splats keep their digits), and rows of
                                      one gaussian from several strips / ranks simply add.  Multi-GPU rank steps use 1. */
  int32_t grad_image_broadcast;    /* != 0 (moments path only): grad_image points to ONE pixel's f values, used for every pixel —
                                      dL/dimage of a sum / mean loss is an expanded scalar, which the caller need not
                                      materialise as an (H, W, f) array (50 MB at 2048^2 written and read back otherwise) */
} ms_frame_grads;

enum { MS_BOUNDARY_AXIS_SIGMA = 0, MS_BOUNDARY_COVARIANCE = 1 };

int ms_frame_layout_query(const ms_frame_desc* desc, ms_frame_layout* out);
int ms_frame_uses_moments(const ms_frame_desc* desc, int deterministic);
int ms_frame_project(const ms_frame_desc* desc, const ms_frame_inputs* in, void* keep_n, void* stream);
int ms_frame_project_count(const ms_frame_desc* desc, const ms_frame_inputs* in, void* keep_n, void* scratch_n,
                           int32_t* k_host, void* k_event, void* stream);
/* The SH colours of a frame (the second half of ms_frame_project: camera position = inverse(T_camera_world)[:3, 3],
 * perspective/params.py:78-80, then evaluate_sh_at, indexed_spherical_harmonics.py:119-134, into keep_n) as a call of its
 * own, so that a caller can enqueue it on ANOTHER stream behind ms_frame_project_count (it reads the depths the projection
 * wrote) and beside the mapper's launches of ms_frame_map_raster: that call, given ms_frame_inputs.colours_ready_event
 * (an event recorded behind this one), skips the colours and waits for the event in front of the raster forward.  Frames
 * with sh_degree >= 0 and their own projection only; not with the splat-row table. */
int ms_frame_sh_colours(const ms_frame_desc* desc, const ms_frame_inputs* in, void* keep_n, void* stream);
int ms_frame_map_raster(const ms_frame_desc* desc, const ms_frame_inputs* in, void* keep_n, void* scratch_n,
                        void* keep_k, void* scratch_k, void* out_image, void* out_alpha, void* out_visibility,
                        void* stream);
int ms_frame_backward(const ms_frame_desc* desc, const ms_frame_inputs* in, void* keep_n, void* keep_k,
                      const ms_frame_grads* g, void* stream);

/* ---- optimiser step ------------------------------------------------------------
 * Moment update of the fractional (visibility-weighted) Adam (kind 0, optim/fractional_adam.py:8-86)
 * and LaProp (kind 1, optim/fractional_laprop.py:8-86) for the m_count visible points listed in
 * indexes: reads grad[idx] (N,D), the per-point step weight (m_count) and total_weight[idx] (N),
 * updates m (N,D) and v ((N,D), or (N,) when vector != 0: one second moment per point from the
 * squared gradient norm) in place and writes lr_step (m_count, D), float32. */
int ms_fractional_step(int kind, int vector, float* lr_step, const int64_t* indexes,
                       const float* weight, float* m, float* v, const float* total_weight,
                       const float* grad, int64_t m_count, int d, float lr, float beta1,
                       float beta2, float eps, int bias_correction, void* stream);

/* Fused per-group update (optim/fractional.py:108-156,176-195 + the gradient pre-scaling of
 * optim/visibility_aware.py:95-104) for the m_count visible rows listed in indexes, in one pass:
 *   g = grad[idx] * grad_scale[i];  local_vector (group_type 2, d = 2 or 3): g = basis[i]^-1 g;
 *   moments m / v updated as in ms_fractional_step (group_type 0: v is (N,D); 1, 2: v is (N,));
 *   step clamped to +-lr*clip (clip < 0: off); local_vector: step = basis[i] step; *= mask_lr[j];
 *   *= point_lr[idx]; non-finite -> 0;  param[idx] -= step * (1 - exp(-2 weight[i])).
 * grad_scale (m_count), basis (m_count, d, d), mask_lr (d), point_lr (N) may be NULL.  float32, d <= 256. */
int ms_fractional_update(int kind, int group_type, float* param, const float* grad, float* m, float* v,
                         const int64_t* indexes, const float* weight, const float* total_weight,
                         const float* grad_scale, const float* basis, const float* mask_lr,
                         const float* point_lr, int64_t m_count, int d, float lr, float beta1,
                         float beta2, float eps, float clip, int bias_correction, void* stream);

/* All parameter groups of one optimiser step in ONE launch (round 6; the reference loops over its groups on the host,
 * optim/fractional.py:176-195, optim/visibility_aware.py:86-104).  Each group is what ms_fractional_update takes;
 * indexes / weight / total_weight / grad_scale are shared by the groups, as they are in the reference's step().
 * Rows whose length is a multiple of 4 floats (16-byte aligned arrays) move as 16-byte pieces.  A row with
 * weight[i] < 0 is skipped; indexes == NULL means rows 0 .. m_count - 1 (both: the dense mode of
 * ms_optim_visibility_weights).  `groups` is a HOST array. */
typedef struct ms_optim_group {
  uint32_t struct_size;            /* sizeof(ms_optim_group) */
  int32_t group_type;              /* 0 scalar, 1 vector, 2 local_vector */
  float* param;                    /* (N, d) */
  const float* grad;               /* (N, d) */
  float* m;                        /* (N, d) first moment */
  float* v;                        /* second moment: (N, d) for scalar groups, (N,) otherwise */
  const float* basis;              /* local_vector: (m_count, d, d), else NULL */
  const float* mask_lr;            /* (d) or NULL */
  const float* point_lr;           /* (N) or NULL */
  int32_t d;
  int32_t bias_correction;
  float lr, beta1, beta2, eps;
  float clip;                      /* < 0: off */
  float reserved;
} ms_optim_group;
int ms_optim_step_groups(int kind, const ms_optim_group* groups, int num_groups, const int64_t* indexes,
                         const float* weight, const float* total_weight, const float* grad_scale, int64_t m_count,
                         void* stream);

/* Step weights of the visibility-aware optimisers in one pass (optim/visibility_aware.py:35-52 update_visibility and
 * :86-104): for the m_count rows listed in indexes, running_vis <- ((1 - beta) v^4 + beta running_vis^4)^(1/4),
 * out_weight = v / max(running_vis, floor_eps), total_weight += out_weight, out_grad_scale = 1 / (v + vis_smooth)
 * (out_grad_scale may be NULL).  indexes == NULL (dense mode): row i is point i, and a point with
 * visibility <= skip_threshold is left untouched and gets out_weight = -1 (which ms_optim_step_groups /
 * ms_fractional_update skip) — the reference's `visible = (visibility > 1e-8).nonzero()` without the host
 * synchronisation (examples/fit_image_gaussians.py:118-119). */
int ms_optim_visibility_weights(const int64_t* indexes, const float* visibility, int64_t m_count, float vis_beta,
                                float vis_smooth, float floor_eps, float skip_threshold, float* running_vis,
                                float* total_weight, float* out_weight, float* out_grad_scale, void* stream);

/* ---- Morton codes --------------------------------------------------------------------
 * out_codes[i] = 63-bit Z-order code of points3[i] (N, 3 float32) on the grid of cell size inc3_host anchored at
 * lower3_host (3 floats each, host memory) with `size` cells per axis (<= 2^21): cell = clamp((p - lower) / inc,
 * 0, size - 1) truncated; x, y, z bits interleaved from bit 0 (misc/morton_sort.py:25-33,56-70,94-99).
 * Sort with ms_radix_sort_pairs (key_bytes 8) for misc/morton_sort.py:113-119 argsort. */
int ms_morton_codes64(const float* points3, int64_t n, const float* lower3_host, const float* inc3_host,
                      uint32_t size, uint64_t* out_codes, void* stream);

/* ---- camera position ----------------------------------------------------------------------------------
 * out_position3 = inverse(T_camera_world)[0:3, 3] for a row-major 4x4 (perspective/params.py:62-65), one
 * launch instead of a device-side LU. */
int ms_camera_position(const void* t_camera_world, void* out_position3, int dtype, void* stream);

/* ---- multi-GPU: routing projected splats to tile-row strips -----------------------------
 * New design, no reference counterpart (the reference renders on one GPU).  Rank r renders the tile rows
 * [bounds[r], bounds[r+1]); a projected splat goes to every rank whose strip meets the tile-row span of
 * the mapper's grid query (taichi_lib/grid_query.py:10-40), i.e. a stable multi-way split by destination.
 * float32 only.
 *
 * ms_strip_route_blocks: number of 256-splat blocks B of the split (scratch: out_block_counts is
 *   world x B int32).
 * ms_strip_route_count: out_route[i] = first_rank | copies << 16; out_block_counts = per-(rank, block)
 *   exclusive slot offsets inside the rank's bucket; out_send_counts[r] (int64, device) = splats routed
 *   to rank r = the all-to-all split sizes.  bounds_host: world + 1 ints in host memory.
 * ms_strip_route_pack: writes the send buffer, buckets in rank order and local index order inside a
 *   bucket: out_rows (S, 9 + f) = [packed 2D (7) | colour (f) | depth | global id as int32 bits], global
 *   id = (ids ? ids[i] : i) + index_offset; out_send_index[slot] = i (to sum the returned gradients).
 * ms_strip_unpack: splits received rows into the arrays the mapper / rasterizer consume.
 * ms_strip_return_grads: backward of the exchange on the sending side: back_rows (S, 7 + f) =
 *   [d packed 2D | d colour] of every slot of the send buffer, summed into the zero-initialised
 *   grad_points7 (V, 7) / grad_features (V, f) of the local splats (grad[send_index[slot]] += row);
 *   route = out_route of ms_strip_route_count (splats with one copy are stored, not accumulated); slots with
 *   send_index < 0 (unused rows of fixed buckets) are skipped. */
int ms_strip_route_blocks(int v);
int ms_strip_route_count(const float* points7, const float* depth /* NULL, or (v): rows with depth <= 0 are culled */,
                         int v, int image_h, int tile_size, float alpha_threshold,
                         const int32_t* bounds_host, int world, int32_t* out_route,
                         int32_t* out_block_counts, int64_t* out_send_counts, void* stream);
/* bucket_capacity > 0: FIXED buckets — destination r owns rows [r * capacity, (r + 1) * capacity) of out_rows, so
 * the all-to-all has equal splits the host knows without reading the counts back (sync-free rank step, HIP-graph
 * capture).  The caller zero-fills out_rows (an all-zero row is a splat with alpha 0: it overlaps no tile and gets
 * no gradient) and fills out_send_index with -1 first; rows beyond a bucket's capacity are dropped and
 * *overflow_flag (device int32, may be NULL) is set.  bucket_capacity == 0: buckets of exactly send_counts[r] rows. */
int ms_strip_route_pack(const float* points7, const float* features, const float* depths,
                        const int64_t* ids, int f, int v, int world, int64_t index_offset,
                        const int32_t* route, const int32_t* block_offsets, const int64_t* send_counts,
                        int64_t bucket_capacity, int32_t* overflow_flag,
                        float* out_rows, int64_t* out_send_index, void* stream);
/* the same, and out_slots[i * world + c] (int32, may be NULL) = row of out_rows that holds copy c of local splat i
 * (c < copies of out_route), -1 for a copy dropped by a full bucket: what ms_frame_grads.gather_slots reads */
int ms_strip_route_pack_slots(const float* points7, const float* features, const float* depths,
                              const int64_t* ids, int f, int v, int world, int64_t index_offset,
                              const int32_t* route, const int32_t* block_offsets, const int64_t* send_counts,
                              int64_t bucket_capacity, int32_t* overflow_flag,
                              float* out_rows, int64_t* out_send_index, int32_t* out_slots, void* stream);
/* the same with the colours in a buffer of their own (round 6: the forward exchange as TWO collectives, so that the
 * receiving strip's mapper — which reads geometry only — runs while the colours are still on the links):
 * out_geometry_rows (S, 9) = [packed 2D (7) | depth | global id bits], out_colour_rows (S, f); both zero-filled by the
 * caller for fixed buckets.  ms_strip_unpack(rows9, m, 0, points7, NULL, depths, ids) splits the geometry rows; the
 * colour rows ARE the (m, f) colour array the rasterizer reads (ms_frame_inputs.colours +
 * ms_frame_inputs.colours_ready_event). */
int ms_strip_route_pack_split(const float* points7, const float* features, const float* depths,
                              const int64_t* ids, int f, int v, int world, int64_t index_offset,
                              const int32_t* route, const int32_t* block_offsets, const int64_t* send_counts,
                              int64_t bucket_capacity, int32_t* overflow_flag, float* out_geometry_rows,
                              float* out_colour_rows, int64_t* out_send_index, int32_t* out_slots, void* stream);
int ms_strip_unpack(const float* rows, int64_t m, int f, float* out_points7, float* out_features,
                    float* out_depths, int64_t* out_ids, void* stream);
int ms_strip_return_grads(const float* back_rows, const int64_t* send_index, const int32_t* route,
                          int f, int64_t s, float* grad_points7, float* grad_features, void* stream);
/* the same into ONE zero-initialised (V, 7 + f) array of rows [d packed 2D | d colour] (one line per splat instead of
 * two; ms_frame_grads.boundary_stride = 7 + f hands it to the per-gaussian pass) */
int ms_strip_return_rows(const float* back_rows, const int64_t* send_index, const int32_t* route,
                         int f, int64_t s, float* grad_rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MI355_SPLAT_H */
