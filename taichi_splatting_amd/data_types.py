"""Data model of the renderer: same names, fields and defaults as the reference.

Mirrors reference ``taichi_splatting/data_types.py``: ``RasterConfig`` (:17-47),
``Gaussians3D`` (:57-115), ``Gaussians2D`` (:122-145).  tensordict / beartype / roma are not
required: containers derive from the in-repo :class:`TensorClass`.
"""
from __future__ import annotations

from dataclasses import dataclass
import math
from typing import List, Tuple

import torch

from .tensorclass import TensorClass


@dataclass(frozen=True, eq=True, kw_only=True)
class RasterConfig:
  """Rasterizer configuration (reference ``data_types.py:17-47``; identical defaults).

  Frozen + hashable so it can be used as a cache key and with ``dataclasses.replace``.
  """
  tile_size: int = 16

  # pixel tiling per thread in the backward pass.  The reference uses it to pick the
  # thread->pixel map; results never depend on it.  The gfx950 kernels always map one wave64
  # to an 8x8 pixel patch, so the value is validated and otherwise ignored.
  pixel_stride: Tuple[int, int] = (2, 2)

  # clamp position to within this margin of the image for the affine jacobian
  clamp_margin: float = 0.15

  antialias: bool = False       # use the anti-aliased pdf
  blur_cov: float = 0.3         # added to the diagonal of the projected covariance

  clamp_max_alpha: float = 0.99
  alpha_threshold: float = 1. / 255.

  saturate_threshold: float = 0.9999   # backward stops at this accumulated weight
  use_alpha_blending: bool = True      # False + saturate_threshold => quantile (median) render

  compute_point_heuristic: bool = False
  compute_visibility: bool = False

  median_threshold: float = 0.25

  def __post_init__(self):
    assert self.tile_size in (8, 16, 32), \
      f"tile_size must be 8, 16 or 32 (one wave64 per 8x8 pixel patch), got {self.tile_size}"
    sx, sy = self.pixel_stride
    assert self.tile_size % sx == 0 and self.tile_size % sy == 0, \
      f"pixel_stride {self.pixel_stride} must divide tile_size {self.tile_size}"
    # reference rasterizer/backward.py:32-33
    assert (self.tile_size * self.tile_size) // (sx * sy) >= 32, \
      f"pixel_stride {self.pixel_stride} and tile_size {self.tile_size} must allow at least one warp sized (32) tile"


def check_packed3d(packed_gaussians: torch.Tensor):
  assert len(packed_gaussians.shape) == 2 and packed_gaussians.shape[1] == 11, \
    f"Expected shape (N, 11), got {packed_gaussians.shape}"


def check_packed2d(packed_gaussians: torch.Tensor):
  # the packed 2D gaussian is 7 floats [mean.xy, axis.xy, sigma.xy, alpha] (taichi_lib/generic.py:30-58)
  assert len(packed_gaussians.shape) == 2 and packed_gaussians.shape[1] == 7, \
    f"Expected shape (N, 7), got {packed_gaussians.shape}"


def _quat_to_mat(q: torch.Tensor) -> torch.Tensor:
  x, y, z, w = q.unbind(-1)
  x2, y2, z2 = x * x, y * y, z * z
  return torch.stack([
    1 - 2 * y2 - 2 * z2, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y,
    2 * x * y + 2 * w * z, 1 - 2 * x2 - 2 * z2, 2 * y * z - 2 * w * x,
    2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x2 - 2 * y2], dim=-1).reshape(q.shape[:-1] + (3, 3))


def _mat_to_quat(m: torch.Tensor) -> torch.Tensor:
  """Rotation matrix -> unit quaternion (xyzw), numerically robust branch selection."""
  m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
  qw = torch.sqrt(torch.clamp_min(1 + m00 + m11 + m22, 0)) / 2
  qx = torch.sqrt(torch.clamp_min(1 + m00 - m11 - m22, 0)) / 2
  qy = torch.sqrt(torch.clamp_min(1 - m00 + m11 - m22, 0)) / 2
  qz = torch.sqrt(torch.clamp_min(1 - m00 - m11 + m22, 0)) / 2
  qx = torch.copysign(qx, m[..., 2, 1] - m[..., 1, 2])
  qy = torch.copysign(qy, m[..., 0, 2] - m[..., 2, 0])
  qz = torch.copysign(qz, m[..., 1, 0] - m[..., 0, 1])
  q = torch.stack([qx, qy, qz, qw], dim=-1)
  return q / torch.norm(q, dim=-1, keepdim=True)


class Gaussians3D(TensorClass):
  """3D gaussians (reference ``data_types.py:57``).  ``rotation`` is an xyzw quaternion."""
  position: torch.Tensor      # 3  - xyz
  log_scaling: torch.Tensor   # 3  - scale = exp(log_scaling)
  rotation: torch.Tensor      # 4  - quaternion xyzw (normalised inside the kernels)
  alpha_logit: torch.Tensor   # 1  - alpha = sigmoid(alpha_logit)
  feature: torch.Tensor       # (N, C) colours or (N, 3, (deg+1)^2) spherical harmonics

  def __post_init__(self):
    assert self.position.shape[1] == 3, f"Expected shape (N, 3), got {self.position.shape}"
    assert self.log_scaling.shape[1] == 3, f"Expected shape (N, 3), got {self.log_scaling.shape}"
    assert self.rotation.shape[1] == 4, f"Expected shape (N, 4), got {self.rotation.shape}"
    assert self.alpha_logit.shape[1] == 1, f"Expected shape (N, 1), got {self.alpha_logit.shape}"

  def packed(self):
    return torch.cat([self.position, self.log_scaling, self.rotation, self.alpha_logit], dim=-1)

  def shape_tensors(self):
    return (self.position, self.log_scaling, self.rotation, self.alpha_logit)

  def scaled(self, scale: float) -> 'Gaussians3D':
    return self.replace(position=self.position * scale,
                        log_scaling=math.log(scale) + self.log_scaling)

  def translated(self, translation: torch.Tensor) -> 'Gaussians3D':
    return self.replace(position=self.position + translation.view(1, 3))

  @property
  def scale(self):
    return torch.exp(self.log_scaling)

  @property
  def alpha(self):
    return torch.sigmoid(self.alpha_logit)

  def transform_rigid(self, m: torch.Tensor) -> 'Gaussians3D':
    """Transform the gaussians by a rigid 4x4 matrix (reference ``data_types.py:91-102``)."""
    assert m.shape == (4, 4), f"Expected shape (4, 4), got {m.shape}"
    r, t = m[:3, :3], m[:3, 3]
    position = self.position @ r.T + t
    q = self.rotation / torch.norm(self.rotation, dim=-1, keepdim=True)
    rotation = _mat_to_quat(r.unsqueeze(0) @ _quat_to_mat(q))
    return self.replace(position=position, rotation=rotation)

  @staticmethod
  def concat_batch(gaussians: List['Gaussians3D']) -> 'Gaussians3D':
    return Gaussians3D.cat(gaussians, dim=0)


def inverse_sigmoid(x: torch.Tensor):
  return torch.log(x / (1 - x))


class Gaussians2D(TensorClass):
  """2D gaussians used by the 2D harness (reference ``data_types.py:122``)."""
  position: torch.Tensor      # 2  - xy
  depths: torch.Tensor        # 1  - for sorting
  log_scaling: torch.Tensor   # 2
  rotation: torch.Tensor      # 2  - unit length complex number
  alpha_logit: torch.Tensor   # 1  - alpha = sigmoid(alpha_logit)
  feature: torch.Tensor       # N  - (any rgb, label etc)

  @property
  def opacity(self):
    return self.alpha_logit.sigmoid()

  @property
  def scaling(self):
    return torch.exp(self.log_scaling)

  def set_scaling(self, scaling) -> 'Gaussians2D':
    return self.replace(log_scaling=torch.log(scaling))
