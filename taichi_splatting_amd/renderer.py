"""``render_gaussians``: project -> SH colour -> tile mapper -> rasterize.

Same interface as reference ``renderer.py:23-108`` (``render_gaussians``, ``render_projected``,
``viewspace_gradient``).
"""
from __future__ import annotations

from dataclasses import replace
from typing import Optional, Tuple

import torch

from .data_types import Gaussians3D, RasterConfig
from .mapper.tile_mapper import map_to_tiles_strip
from .perspective import CameraParams
from .perspective.projection import project_to_image
from .rasterizer.function import rasterize_with_tiles
from .rendering import RenderedPoints, Rendering, ndc_depth
from .spherical_harmonics import evaluate_sh_at


def render_gaussians(
  gaussians: Gaussians3D,
  camera_params: CameraParams,
  config: RasterConfig = RasterConfig(),
  use_sh: bool = False,
  render_depth: bool = False,
  use_depth16: bool = False,
  render_median_depth: bool = False,
  tile_rows: Optional[Tuple[int, int]] = None,
) -> Rendering:
  """A complete renderer for 3D gaussians.

  Parameters:
    gaussians: Gaussians3D, feature (N, C) or spherical harmonics (N, 3, (D+1)**2)
    camera_params: CameraParams
    config: RasterConfig
    use_sh: evaluate ``feature`` as spherical harmonics of the view direction
    render_depth: accepted for compatibility (the reference ignores it too: renderer.py:84)
    use_depth16: 16 bit depth sort keys (otherwise 32 bit)
    render_median_depth: extra quantile pass producing ``median_depth_image``
    tile_rows: optional (begin, end) tile-row strip to render (multi-GPU sharding)
  """
  from . import frame
  if frame.USE_FRAME and frame.frame_supported(gaussians.feature, config, use_sh):
    # one autograd node on a fixed launch sequence, no host round trip for the visible / overlap counts (frame.py)
    return frame.render_frame(gaussians, camera_params, config, use_sh, use_depth16=use_depth16,
                              render_median_depth=render_median_depth, tile_rows=tile_rows)

  # modular composition (wide feature vectors, MS_FRAME=legacy): launched before the projection's host
  # synchronisation (visible count) so that the first kernel queued after it is the long SH pass
  camera_position = camera_params.camera_position if use_sh else None
  gaussians2d, depths, indexes = project_to_image(gaussians, camera_params, config)

  if use_sh:
    features = evaluate_sh_at(gaussians.feature, gaussians.position.detach(), indexes,
                              camera_position, unique_indexes=True)
  else:
    features = gaussians.feature[indexes]
    assert len(features.shape) == 2, f"Features must be (N, C) if use_sh=False, got {features.shape}"

  return render_projected(indexes, gaussians2d, features, depths, camera_params, config,
                          use_depth16=use_depth16, render_median_depth=render_median_depth,
                          tile_rows=tile_rows)


def render_projected(indexes: torch.Tensor, gaussians2d: torch.Tensor, features: torch.Tensor,
                     depths: torch.Tensor, camera_params: CameraParams, config: RasterConfig,
                     use_depth16: bool = False, render_median_depth: bool = False,
                     tile_rows: Optional[Tuple[int, int]] = None, crop_to_rows: bool = False) -> Rendering:
  # crop_to_rows: with tile_rows, the images hold only the strip's pixel rows (multi-GPU strips)
  # ndc depth (renderer.py:67) is computed inside the mapper's key kernel
  overlap_to_point, tile_overlap_ranges = map_to_tiles_strip(
    gaussians2d, depths.detach(), image_size=camera_params.image_size, config=config,
    use_depth16=use_depth16, tile_rows=tile_rows,
    ndc_range=(camera_params.near_plane, camera_params.far_plane))

  raster = rasterize_with_tiles(
    gaussians2d, features,
    tile_overlap_ranges=tile_overlap_ranges.view(-1, 2), overlap_to_point=overlap_to_point,
    image_size=camera_params.image_size, config=config, tile_rows=tile_rows, crop_to_rows=crop_to_rows)

  median_depth = None
  if render_median_depth:
    raster_depth = rasterize_with_tiles(
      gaussians2d, depths,
      tile_overlap_ranges=tile_overlap_ranges.view(-1, 2), overlap_to_point=overlap_to_point,
      image_size=camera_params.image_size,
      config=replace(config, use_alpha_blending=False, saturate_threshold=config.median_threshold,
                     compute_visibility=False, compute_point_heuristic=False),
      tile_rows=tile_rows, crop_to_rows=crop_to_rows)
    median_depth = raster_depth.image.squeeze(-1)

  points = RenderedPoints(
    idx=indexes,
    depths=depths,
    gaussians2d=gaussians2d,
    _visibility=raster.visibility if config.compute_visibility else None,
    _prune_cost=raster.point_heuristic[:, 0] if config.compute_point_heuristic else None,
    _split_score=raster.point_heuristic[:, 1] if config.compute_point_heuristic else None,
    features=features,
    attributes=None,
    batch_size=(depths.shape[0],))

  return Rendering(image=raster.image,
                   image_weight=raster.image_weight,
                   depth_image=None,
                   median_depth_image=median_depth,
                   points=points,
                   camera=camera_params,
                   config=config)


def viewspace_gradient(gaussians2d: torch.Tensor):
  assert gaussians2d.shape[1] == 7, f"Expected packed 2D gaussians (N, 7), got {gaussians2d.shape}"
  assert gaussians2d.grad is not None, \
    "Expected gradients on gaussians2d, run backward first with gaussians2d.retain_grad()"
  xy_grad = gaussians2d.grad[:, :2]
  return torch.norm(xy_grad, dim=1)
