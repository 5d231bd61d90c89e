"""Result containers of the renderer (reference ``rendering.py:17-157``)."""
from __future__ import annotations

from dataclasses import dataclass, fields
from functools import cached_property
from typing import Any, Optional, Tuple

import torch

from .data_types import RasterConfig
from .perspective.params import CameraParams
from .tensorclass import TensorClass


def ndc_depth(depth: torch.Tensor, near: float, far: float) -> torch.Tensor:
  """ndc from 0 (near) to 1 (far) (reference ``torch_lib/projection.py:120-123``)."""
  return 1 - (1. / depth - 1. / far) / (1. / near - 1. / far)


def inverse_ndc_depth(ndc: torch.Tensor, near: float, far: float) -> torch.Tensor:
  """reference ``torch_lib/projection.py:126-129``"""
  return 1.0 / ((1.0 - ndc) * (1 / near - 1 / far) + 1 / far)


def unpack(dc) -> dict:
  return {field.name: getattr(dc, field.name) for field in fields(dc)}


class Indexed(TensorClass):
  idx: torch.Tensor    # N, index of points in larger scene
  data: torch.Tensor   # N, K data of points

  def expanded(self, n: int) -> torch.Tensor:
    data = torch.zeros((n, *self.data.shape[1:]), dtype=self.data.dtype, device=self.data.device)
    data[self.idx] = self.data
    return data


class RenderedPoints(TensorClass):
  idx: torch.Tensor            # index of points in larger scene
  depths: torch.Tensor         # point depths
  gaussians2d: torch.Tensor    # 7, 2d gaussians after projection
  features: torch.Tensor       # rendered features of points e.g. colour

  _prune_cost: Optional[torch.Tensor] = None
  _split_score: Optional[torch.Tensor] = None
  _visibility: Optional[torch.Tensor] = None
  attributes: Optional[Any] = None

  @property
  def prune_cost(self):
    assert self._prune_cost is not None, \
      "No prune cost information available (render with config.compute_point_heuristic=True)"
    return self._prune_cost

  @property
  def split_score(self):
    assert self._split_score is not None, \
      "No split score information available (render with config.compute_point_heuristic=True)"
    return self._split_score

  @property
  def visibility(self):
    assert self._visibility is not None, \
      "No visibility information available (render with config.compute_visibility=True)"
    return self._visibility

  @property
  def screen_scale(self):
    return self.gaussians2d[:, 4:6]

  @property
  def opacity(self):
    return self.gaussians2d[:, 6]

  @property
  def visible_mask(self) -> torch.Tensor:
    return self.visibility > 0.0

  @property
  def visible(self) -> 'RenderedPoints':
    return self[self.visible_mask]

  @property
  def num_visible(self) -> int:
    return int(self.visible_mask.sum().item())

  @property
  def indexed_visibility(self) -> Indexed:
    return Indexed(idx=self.idx, data=self.visibility)

  def full_mask(self, n: int) -> torch.Tensor:
    mask = torch.zeros((n,), dtype=torch.bool, device=self.idx.device)
    mask[self.idx] = self.visible_mask
    return mask

  def full_visibility(self, n: int) -> torch.Tensor:
    vis = torch.zeros((n,), dtype=self.visibility.dtype, device=self.visibility.device)
    vis[self.idx] = self.visibility
    return vis

  def gaussian_scale(self, alpha_threshold: float = 1.0 / 255):
    """Factor of the gaussian bounds used for culling (3DGS uses a fixed 3.0)."""
    return torch.sqrt(2 * torch.log(self.opacity / alpha_threshold))

  def ndc(self, near: float, far: float):
    return ndc_depth(self.depths, near, far)


@dataclass(frozen=True)
class Rendering:
  """Collection of outputs from the renderer (reference ``rendering.py:106-157``)."""
  image: torch.Tensor                 # (H, W, C) rendered image
  image_weight: torch.Tensor          # (H, W) total alpha per pixel

  points: RenderedPoints              # (V,) rendered points which were in view
  camera: CameraParams
  config: RasterConfig

  depth_image: Optional[torch.Tensor] = None          # (H, W)
  median_depth_image: Optional[torch.Tensor] = None   # (H, W)
  glo_feature: Optional[torch.Tensor] = None

  def __getattribute__(self, name):
    # the frame executor (frame.py) does not compact the visible gaussians; `points` then holds a LazyPoints that
    # builds the (V, ...) arrays — with the host read of V the reference does in its projection — on first access
    value = object.__getattribute__(self, name)
    if name == 'points' and hasattr(value, 'materialise'):
      value = value.materialise()
      object.__setattr__(self, 'points', value)
    return value

  @cached_property
  def ndc_image(self) -> torch.Tensor:
    return ndc_depth(self.depth_image, self.camera.near_plane, self.camera.far_plane)

  @cached_property
  def median_ndc_image(self) -> torch.Tensor:
    return ndc_depth(self.median_depth_image, self.camera.near_plane, self.camera.far_plane)

  @property
  def visible_idx(self) -> torch.Tensor:
    return self.points.idx[self.points.visible_mask]

  @property
  def in_view_idx(self) -> torch.Tensor:
    return self.points.idx

  @property
  def visible_points(self) -> RenderedPoints:
    return self.points[self.points.visible_mask]

  @property
  def image_size(self) -> Tuple[int, int]:
    return self.camera.image_size

  def detach(self):
    return Rendering(**{k: x.detach() if hasattr(x, 'detach') else x
                        for k, x in unpack(self).items()})
