"""Differentiable tile rasterizer: same operator surface as reference
``rasterizer/function.py`` (``rasterize`` :133, ``rasterize_with_tiles`` :100, ``RasterOut`` :19,
autograd glue :28-97), launching the gfx950 kernels of csrc/raster.hip through the C-ABI.
"""
from __future__ import annotations

from dataclasses import replace
import os
from numbers import Integral
from typing import NamedTuple, Optional, Tuple

import torch

from .. import _lib
from ..data_types import RasterConfig
from ..mapper.tile_mapper import map_to_tiles

RasterOut = NamedTuple('RasterOut', [
  ('image', torch.Tensor),
  ('image_weight', torch.Tensor),
  ('point_heuristic', Optional[torch.Tensor]),
  ('visibility', Optional[torch.Tensor])
])

MAX_KERNEL_FEATURES = 4   # csrc/raster.hip instantiates F = 1..4; wider features are chunked
WIDE_KERNEL_FEATURES = (8, 16)   # backward with point heuristics: instantiated too (zero-padded up to these widths)


# Bitwise reproducible gradients for the product path (float32 RGB): fixed-point integer commits instead of float
# atomics (csrc/raster_bwd_scan.hip).  Set the module attribute or MS_DETERMINISTIC=1; costs a 128 B instead of
# 64 B accumulator row per gaussian.  RasterConfig stays the reference's dataclass, hence no field there.
DETERMINISTIC_BACKWARD = os.environ.get('MS_DETERMINISTIC', '0') not in ('0', '')


def _use_moments_backward(config: RasterConfig, dtype, f: int) -> bool:
  """Product path (float32 RGB, plain pdf): the splat-per-lane scan kernel of csrc/raster_bwd_scan.hip at every tile
  size (config D: 1.65 + 0.14 ms at tile 8, 1.43 + 0.14 at tile 16, 1.93 + 0.14 at tile 32 with one 1024-thread
  workgroup per tile; the pixel-per-lane kernel of raster_fast.hip takes 3.0 ms there).  ``MS_RASTER_BWD=patch``
  forces the pixel-per-lane kernels (A/B measurements)."""
  return dtype == torch.float32 and f == 3 and not config.antialias and os.environ.get('MS_RASTER_BWD', 'scan') != 'patch'


def _tile_rows(config: RasterConfig, image_size, tile_rows):
  tiles_high = (image_size[1] + config.tile_size - 1) // config.tile_size
  if tile_rows is None:
    return 0, tiles_high
  return max(0, int(tile_rows[0])), min(tiles_high, int(tile_rows[1]))


def _strip_pixels(rows, tile_size, h):
  """Pixel rows [y0, y1) covered by the tile rows ``rows``."""
  return min(rows[0] * tile_size, h), min(rows[1] * tile_size, h)


def _forward_chunk(lib, gaussians, features, ranges, o2p, image_size, cfg_c, visibility, rows, stream, crop):
  w, h = image_size
  f = features.shape[1]
  dtype = gaussians.dtype
  y0, y1 = _strip_pixels(rows, cfg_c.tile_size, h) if crop else (0, h)
  # cropped: only the strip's pixel rows exist; the kernels address absolute rows, so they get the
  # address row 0 WOULD have (they touch rows [y0, y1) only)
  image = torch.empty((y1 - y0, w, f), dtype=dtype, device=gaussians.device)
  alpha = torch.empty((y1 - y0, w), dtype=dtype, device=gaussians.device)
  if not crop and rows != (0, (h + cfg_c.tile_size - 1) // cfg_c.tile_size):
    image.zero_(); alpha.zero_()   # rows outside the strip are not rendered
  if y1 > y0:
    _lib.check(lib.ms_raster_fwd(gaussians.data_ptr(), features.data_ptr(), ranges.data_ptr(), _lib.ptr(o2p),
                                 w, h, f, cfg_c, image.data_ptr() - y0 * w * f * image.element_size(),
                                 alpha.data_ptr() - y0 * w * alpha.element_size(), _lib.ptr(visibility),
                                 rows[0], rows[1], _lib.dtype_code(dtype), stream), "rasterize forward")
  return image, alpha


class _RasterFunction(torch.autograd.Function):
  """reference rasterizer/function.py:42-95"""

  @staticmethod
  def forward(ctx, gaussians, features, overlap_to_point, tile_overlap_ranges, image_size, config, tile_rows,
              crop_to_rows=False):
    lib = _lib.load()
    _lib.require_gpu(gaussians, features, overlap_to_point, tile_overlap_ranges)
    assert gaussians.ndim == 2 and gaussians.shape[1] == 7, f"gaussians2d must be (N, 7), got {gaussians.shape}"
    assert features.ndim == 2 and features.shape[0] == gaussians.shape[0], \
      f"features must be (N, F), got {features.shape} for {gaussians.shape[0]} gaussians"
    assert features.dtype == gaussians.dtype, f"dtype mismatch {features.dtype} != {gaussians.dtype}"

    gaussians_c = gaussians.detach().contiguous()
    features_c = features.detach().contiguous()
    o2p = overlap_to_point.contiguous()
    assert o2p.dtype == torch.int32 and tile_overlap_ranges.dtype == torch.int32
    ranges = tile_overlap_ranges.contiguous().view(-1, 2)

    w, h = int(image_size[0]), int(image_size[1])
    ts = config.tile_size
    assert ranges.shape[0] == ((w + ts - 1) // ts) * ((h + ts - 1) // ts), \
      f"tile_overlap_ranges has {ranges.shape[0]} tiles, image {w}x{h} with tile_size {ts} needs " \
      f"{((w + ts - 1) // ts) * ((h + ts - 1) // ts)}"

    device, dtype = gaussians.device, gaussians.dtype
    n, f = features_c.shape
    rows = _tile_rows(config, (w, h), tile_rows)
    stream = _lib.current_stream(device)
    cfg_c = _lib.raster_config_c(config)

    # compute_visibility with use_alpha_blending=False (quantile render): forward.py:114-126 keeps summing the blend
    # weights into `visibility`; here over EVERY gated splat of every pixel (the generic kernel has no early exit),
    # which is what oracle/raster.py restates.  The reference stops a warp once its 32 pixels are saturated
    # (forward.py:92-94), so its values are <= these and depend on its thread -> pixel map.
    if config.compute_point_heuristic and f > WIDE_KERNEL_FEATURES[-1]:
      raise NotImplementedError(f"compute_point_heuristic with {f} > {WIDE_KERNEL_FEATURES[-1]} feature channels: "
                                "prune_cost / split_score need dL/dalpha over ALL channels at once")
    if config.compute_point_heuristic:
      point_heuristic = torch.zeros((n, 2), dtype=dtype, device=device)
    else:
      point_heuristic = torch.empty((0, 2), dtype=dtype, device=device)
    if config.compute_visibility:
      visibility = torch.zeros((n,), dtype=dtype, device=device)
    else:
      visibility = torch.empty((0,), dtype=dtype, device=device)

    if f <= MAX_KERNEL_FEATURES:
      image, alpha = _forward_chunk(lib, gaussians_c, features_c, ranges, o2p, (w, h), cfg_c,
                                    visibility if config.compute_visibility else None, rows, stream, crop_to_rows)
    else:
      # channels are independent in the forward pass: render them MAX_KERNEL_FEATURES at a time
      images = []
      for c0 in range(0, f, MAX_KERNEL_FEATURES):
        chunk = features_c[:, c0:c0 + MAX_KERNEL_FEATURES].contiguous()
        vis = visibility if (config.compute_visibility and c0 == 0) else None
        cfg_chunk = cfg_c if c0 == 0 else _lib.raster_config_c(replace(config, compute_visibility=False))
        img, alpha_c = _forward_chunk(lib, gaussians_c, chunk, ranges, o2p, (w, h), cfg_chunk, vis, rows, stream,
                                      crop_to_rows)
        images.append(img)
        if c0 == 0:
          alpha = alpha_c
      image = torch.cat(images, dim=2)

    ctx.set_materialize_grads(False)      # no zero tensors for image_weight / heuristics / visibility gradients
    ctx.overlap_to_point = o2p
    ctx.tile_overlap_ranges = ranges
    ctx.image_size = (w, h)
    ctx.config = config
    ctx.rows = rows
    ctx.y0 = _strip_pixels(rows, ts, h)[0] if crop_to_rows else 0
    ctx.point_heuristic = point_heuristic
    ctx.mark_non_differentiable(alpha, point_heuristic, visibility)
    ctx.save_for_backward(gaussians_c, features_c, image)
    return image, alpha, point_heuristic, visibility

  @staticmethod
  def backward(ctx, grad_image, grad_alpha, grad_point_heuristic, grad_visibility):
    lib = _lib.load()
    gaussians, features, image = ctx.saved_tensors
    config = ctx.config
    w, h = ctx.image_size
    n, f = features.shape
    need_points, need_features = ctx.needs_input_grad[0], ctx.needs_input_grad[1]

    heuristic = ctx.point_heuristic if config.compute_point_heuristic else None
    if grad_image is None or not (need_points or need_features or heuristic is not None):
      return None, None, None, None, None, None, None, None
    moments_path = _use_moments_backward(config, gaussians.dtype, f) and image.shape[0] > 0 and n > 0
    alloc = torch.empty_like if moments_path else torch.zeros_like   # the finalize pass stores, the others accumulate
    grad_gaussians = alloc(gaussians) if need_points else None
    grad_features = alloc(features) if need_features else None

    grad_image = grad_image.contiguous()
    if image.shape[0] == 0:
      return grad_gaussians, grad_features, None, None, None, None, None, None
    row_bytes = ctx.y0 * w * image.element_size()      # cropped strip: address of the (absent) row 0
    stream = _lib.current_stream(gaussians.device)
    dtype_code = _lib.dtype_code(gaussians.dtype)
    cfg_c = _lib.raster_config_c(config)

    if moments_path:
      det = int(DETERMINISTIC_BACKWARD)
      moments = torch.zeros((n, _lib.MOMENT_ROW), dtype=torch.int64 if det else torch.float32, device=gaussians.device)
      fixed_exp = _lib.fixed_point_exponents(grad_image) if det else None
      _lib.check(lib.ms_raster_bwd_moments(gaussians.data_ptr(), features.data_ptr(), ctx.tile_overlap_ranges.data_ptr(),
                                           _lib.ptr(ctx.overlap_to_point), image.data_ptr() - row_bytes * f,
                                           grad_image.data_ptr() - row_bytes * f, w, h, cfg_c, moments.data_ptr(), det,
                                           _lib.ptr(fixed_exp), ctx.rows[0], ctx.rows[1], stream), "rasterize backward")
      _lib.check(lib.ms_raster_moments_finalize(gaussians.data_ptr(), moments.data_ptr(), det, _lib.ptr(fixed_exp), n, _lib.ptr(grad_gaussians),
                                                _lib.ptr(grad_features), _lib.ptr(heuristic), stream),
                 "rasterize backward (moments -> gradients)")
    elif f > MAX_KERNEL_FEATURES and heuristic is not None:
      # the heuristics square / take |.| of dL/dalpha summed over ALL channels (backward.py:171-194), so the
      # channels cannot be split: one launch of the F = 8 / 16 instantiation on zero-padded channels (a channel
      # whose dL/dimage is zero contributes nothing)
      fw = next(k for k in WIDE_KERNEL_FEATURES if k >= f)
      pad = lambda t: torch.nn.functional.pad(t, (0, fw - f)).contiguous()
      feat_w, img_w, gimg_w = pad(features), pad(image), pad(grad_image)
      gfeat_w = torch.zeros_like(feat_w) if need_features else None
      _lib.check(lib.ms_raster_bwd(gaussians.data_ptr(), feat_w.data_ptr(), ctx.tile_overlap_ranges.data_ptr(),
                                   _lib.ptr(ctx.overlap_to_point), img_w.data_ptr() - row_bytes * fw,
                                   gimg_w.data_ptr() - row_bytes * fw, w, h, fw, cfg_c, _lib.ptr(grad_gaussians),
                                   _lib.ptr(gfeat_w), _lib.ptr(heuristic), ctx.rows[0], ctx.rows[1], dtype_code, stream),
                 "rasterize backward")
      if need_features:
        grad_features.copy_(gfeat_w[:, :f])
    elif f <= MAX_KERNEL_FEATURES:
      _lib.check(lib.ms_raster_bwd(gaussians.data_ptr(), features.data_ptr(), ctx.tile_overlap_ranges.data_ptr(),
                                   _lib.ptr(ctx.overlap_to_point), image.data_ptr() - row_bytes * f,
                                   grad_image.data_ptr() - row_bytes * f, w, h, f, cfg_c, _lib.ptr(grad_gaussians), _lib.ptr(grad_features),
                                   _lib.ptr(heuristic), ctx.rows[0], ctx.rows[1], dtype_code, stream),
                 "rasterize backward")
    else:
      # d(alpha) = sum_c (...)_c * G_c is linear in the channels, so point gradients of channel
      # chunks add up exactly (no heuristics on this path: see above).
      for c0 in range(0, f, MAX_KERNEL_FEATURES):
        sl = slice(c0, c0 + MAX_KERNEL_FEATURES)
        feat_c = features[:, sl].contiguous()
        img_c = image[:, :, sl].contiguous()
        gimg_c = grad_image[:, :, sl].contiguous()
        gfeat_c = torch.zeros_like(feat_c) if need_features else None
        _lib.check(lib.ms_raster_bwd(gaussians.data_ptr(), feat_c.data_ptr(), ctx.tile_overlap_ranges.data_ptr(),
                                     _lib.ptr(ctx.overlap_to_point), img_c.data_ptr() - row_bytes * feat_c.shape[1],
                                     gimg_c.data_ptr() - row_bytes * feat_c.shape[1], w, h, feat_c.shape[1], cfg_c, _lib.ptr(grad_gaussians), _lib.ptr(gfeat_c),
                                     None, ctx.rows[0], ctx.rows[1],
                                     dtype_code, stream), "rasterize backward")
        if need_features:
          grad_features[:, sl] = gfeat_c

    return grad_gaussians, grad_features, None, None, None, None, None, None


def rasterize_with_tiles(gaussians2d: torch.Tensor, features: torch.Tensor,
                         overlap_to_point: torch.Tensor, tile_overlap_ranges: torch.Tensor,
                         image_size: Tuple[Integral, Integral], config: RasterConfig,
                         tile_rows: Optional[Tuple[int, int]] = None, crop_to_rows: bool = False) -> RasterOut:
  """Rasterize an image given 2d gaussians, features and tile overlap information.

  Parameters:
      gaussians2d: (N, 7) packed gaussians [mean2, axis2, sigma2, alpha]
      features: (N, F) features
      tile_overlap_ranges: (TH * TW, 2) tile index -> range of overlap indices
      overlap_to_point: (K,) overlap index -> point index
      image_size: (width, height)
      config: RasterConfig
      tile_rows: optional (begin, end) tile-row window (multi-GPU strips); rows outside are zero
      crop_to_rows: with tile_rows, return only the strip's pixel rows: image (y1 - y0, W, F)

  Returns RasterOut(image (H, W, F), image_weight (H, W), point_heuristic (N, 2)|(0, 2),
  visibility (N,)|(0,)).  Differentiable w.r.t. gaussians2d and features.
  """
  image, image_weight, point_heuristic, visibility = _RasterFunction.apply(
    gaussians2d, features, overlap_to_point, tile_overlap_ranges, image_size, config, tile_rows, crop_to_rows)
  return RasterOut(image, image_weight, point_heuristic, visibility)


def rasterize(gaussians2d: torch.Tensor, depth: torch.Tensor, features: torch.Tensor,
              image_size: Tuple[Integral, Integral], config: RasterConfig,
              use_depth16: bool = False) -> RasterOut:
  """Tile-map then rasterize (reference rasterizer/function.py:133-165)."""
  assert gaussians2d.shape[0] == depth.shape[0] == features.shape[0], \
    f"Size mismatch: got {gaussians2d.shape}, {depth.shape}, {features.shape}"

  from .. import frame
  if (frame.USE_FRAME and features.ndim == 2 and 1 <= features.shape[1] <= MAX_KERNEL_FEATURES and gaussians2d.is_cuda
      and depth.dtype in (torch.float32, torch.float64)):
    # one node on the frame executor (projected input): no host read of the overlap total before the frame is enqueued
    return RasterOut(*frame.rasterize_frame(gaussians2d, depth, features, image_size, config, use_depth16))

  overlap_to_point, tile_overlap_ranges = map_to_tiles(
    gaussians2d, depth, image_size=image_size, config=config, use_depth16=use_depth16)

  return rasterize_with_tiles(
    gaussians2d, features,
    tile_overlap_ranges=tile_overlap_ranges.view(-1, 2),
    overlap_to_point=overlap_to_point,
    image_size=image_size,
    config=config)
