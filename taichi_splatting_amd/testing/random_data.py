"""Synthetic scene generators with the same distributions AND the same torch RNG call sequence as
the reference's ``tests/random_data.py`` (``random_camera`` :15-45, ``random_3d_gaussians`` :48-75,
``random_2d_gaussians`` :78-103), so that a given ``torch.manual_seed`` yields identical inputs
(pinned by tests/golden/random_data_seed*.pt).  CPU tensors are returned, like the reference.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from ..data_types import Gaussians2D, Gaussians3D, _quat_to_mat
from ..perspective.params import CameraParams
from ..rendering import inverse_ndc_depth


def _join_rt(r, t):
  T = torch.eye(4, device=r.device, dtype=r.dtype)
  T[0:3, 0:3] = r
  T[0:3, 3] = t
  return T


def _inverse_sigmoid(x: torch.Tensor):
  return torch.log(x / (1 - x))


def _unproject_points(uv, depth, transform):
  # torch_lib/projection.py:56-60
  points = torch.concatenate([uv * depth, depth, torch.ones_like(depth)], axis=-1)
  transformed = (torch.inverse(transform).reshape(1, 4, 4) @ points.reshape(-1, 4, 1))[..., 0]
  return transformed[..., 0:3] / transformed[..., 3:4]


def random_camera(pos_scale: float = 1., image_size: Optional[Tuple[int, int]] = None,
                  image_size_range: Tuple[int, int] = (256, 1024), near_plane=0.1) -> CameraParams:
  assert near_plane > 0

  q = F.normalize(torch.randn((1, 4)))
  t = torch.randn((3)) * pos_scale

  T_world_camera = _join_rt(_quat_to_mat(q), t)
  T_camera_world = torch.inverse(T_world_camera)

  if image_size is None:
    min_size, max_size = image_size_range
    image_size = [x.item() for x in torch.randint(size=(2,), low=min_size, high=max_size)]

  w, h = image_size
  cx, cy = torch.tensor([w / 2, h / 2]) + torch.randn(2) * (w / 20)

  fov = torch.deg2rad(torch.rand(1) * 70 + 30)
  fx = w / (2 * torch.tan(fov / 2))
  fy = h / (2 * torch.tan(fov / 2))

  projection = torch.tensor([fx, fy, cx, cy], dtype=torch.float32)

  return CameraParams(
    T_camera_world=T_camera_world,
    projection=projection,
    image_size=(w, h),
    near_plane=near_plane,
    far_plane=near_plane * 1000.)


def random_3d_gaussians(n, camera_params: CameraParams, scale_factor: float = 1.0,
                        alpha_range=(0.1, 0.9), margin=0.0) -> Gaussians3D:
  w, h = camera_params.image_size
  uv_pos = (torch.rand(n, 2) * (1 + margin) - margin * 0.5) * torch.tensor([w, h], dtype=torch.float32).unsqueeze(0)

  depth = inverse_ndc_depth(torch.rand(n), camera_params.near_plane * 2, camera_params.far_plane)

  position = _unproject_points(uv_pos, depth.unsqueeze(1), camera_params.T_image_world)
  fx = camera_params.T_image_camera[0, 0]

  scale = (w / math.sqrt(n)) * (depth / fx) * scale_factor
  scaling = torch.randn(n, 3) * 0.5 + torch.log(scale).unsqueeze(1)

  rotation = torch.randn(n, 4)
  rotation = F.normalize(rotation, dim=1)

  low, high = alpha_range
  alpha = torch.rand(n) * (high - low) + low

  return Gaussians3D(
    position=position,
    log_scaling=scaling,
    rotation=rotation,
    alpha_logit=_inverse_sigmoid(alpha).unsqueeze(1),
    feature=torch.rand(n, 3),
    batch_size=(n,))


def random_2d_gaussians(n, image_size: Tuple[int, int], num_channels=3, scale_factor=1.0,
                        alpha_range=(0.1, 0.9), depth_range=(0.0, 1.0)) -> Gaussians2D:
  w, h = image_size

  position = torch.rand(n, 2) * torch.tensor([w, h], dtype=torch.float32).unsqueeze(0)
  depth = torch.rand((n, 1)) * (depth_range[1] - depth_range[0]) + depth_range[0]

  density_scale = scale_factor * w / (1 + math.sqrt(n))
  scaling = (torch.rand(n, 2) + 0.2) * density_scale

  rotation = torch.randn(n, 2)
  rotation = rotation / torch.norm(rotation, dim=1, keepdim=True)

  low, high = alpha_range
  alpha = torch.rand(n) * (high - low) + low

  return Gaussians2D(
    position=position,
    depths=depth,
    log_scaling=torch.log(scaling),
    rotation=rotation,
    alpha_logit=_inverse_sigmoid(alpha),
    feature=torch.rand(n, num_channels),
    batch_size=(n,))
