"""Seeded synthetic scenes for tests and benchmarks.

Contract: after ``torch.manual_seed(s)`` the three public generators return bit-for-bit what the
reference's ``tests/random_data.py`` returns for the same arguments (``random_camera`` :15-45,
``random_3d_gaussians`` :48-75, ``random_2d_gaussians`` :78-103), so fixtures and benchmark scenes are shared.  That
fixes two things and nothing else: the ORDER and SHAPES of the ``torch.rand`` / ``randn`` / ``randint`` draws — listed
in each function's docstring — and the floating-point expression each draw goes through.
``tests/test_oracle_golden.py::test_generators_reproduce_reference_streams`` holds both against
``tests/golden/random_data_seed*.pt``.  Everything is built on the CPU, like the reference.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import torch

from ..data_types import Gaussians2D, Gaussians3D, _quat_to_mat
from ..perspective.params import CameraParams
from ..rendering import inverse_ndc_depth


# ---- draws shared by the generators (each consumes exactly the RNG calls named) --------------------------------------

def _unit_rows(shape) -> torch.Tensor:
  """randn(shape), rows scaled to unit length (clamped like F.normalize)"""
  v = torch.randn(shape)
  return v / v.norm(dim=-1, keepdim=True).clamp_min(1e-12)


def _opacity(n: int, bounds: Sequence[float]) -> torch.Tensor:
  """rand(n) mapped onto [bounds[0], bounds[1])"""
  lo, hi = bounds
  return torch.rand(n) * (hi - lo) + lo


def _logit(p: torch.Tensor) -> torch.Tensor:
  return torch.log(p / (1 - p))


def _image_extent(w, h) -> torch.Tensor:
  return torch.tensor([w, h], dtype=torch.float32).unsqueeze(0)


def _lift_to_world(pixels: torch.Tensor, z: torch.Tensor, T_image_world: torch.Tensor) -> torch.Tensor:
  """Back-project pixel coordinates at depth z (n, 1): homogeneous image point (u z, v z, z, 1) through the inverse
  of the world -> image transform, then the perspective divide (torch_lib/projection.py:56-60)."""
  homogeneous = torch.cat([pixels * z, z, torch.ones_like(z)], dim=-1)
  world = torch.matmul(torch.inverse(T_image_world).reshape(1, 4, 4), homogeneous.reshape(-1, 4, 1))[..., 0]
  return world[..., 0:3] / world[..., 3:4]


# ---- public generators -----------------------------------------------------------------------------------------------

def random_camera(pos_scale: float = 1., image_size: Optional[Tuple[int, int]] = None,
                  image_size_range: Tuple[int, int] = (256, 1024), near_plane=0.1) -> CameraParams:
  """A pinhole camera with a random pose, principal point and field of view (30-100 degrees).

  Draws, in order: randn(1, 4) orientation quaternion; randn(3) position; randint(2) image size (only when none is
  given); randn(2) principal-point offset; rand(1) field of view."""
  assert near_plane > 0
  orientation = _unit_rows((1, 4))
  centre = torch.randn((3)) * pos_scale

  pose = torch.eye(4, dtype=orientation.dtype)            # camera -> world
  pose[0:3, 0:3] = _quat_to_mat(orientation)
  pose[0:3, 3] = centre

  if image_size is None:
    smallest, largest = image_size_range
    image_size = [side.item() for side in torch.randint(size=(2,), low=smallest, high=largest)]
  w, h = image_size

  principal = torch.tensor([w / 2, h / 2]) + torch.randn(2) * (w / 20)
  half_tan = torch.tan(torch.deg2rad(torch.rand(1) * 70 + 30) / 2)
  intrinsics = torch.tensor([w / (2 * half_tan), h / (2 * half_tan), principal[0], principal[1]], dtype=torch.float32)

  return CameraParams(T_camera_world=torch.inverse(pose), projection=intrinsics, image_size=(w, h),
                      near_plane=near_plane, far_plane=near_plane * 1000.)


def random_3d_gaussians(n, camera_params: CameraParams, scale_factor: float = 1.0,
                        alpha_range=(0.1, 0.9), margin=0.0) -> Gaussians3D:
  """n gaussians scattered through the camera's frustum, sized so that their footprints tile the image about once
  (screen-space sigma ~ w / sqrt(n) pixels at any depth) with a log-normal spread of 0.5 per axis.

  Draws, in order: rand(n, 2) pixel position; rand(n) ndc depth; randn(n, 3) log-scale jitter; randn(n, 4) rotation;
  rand(n) opacity; rand(n, 3) colour."""
  w, h = camera_params.image_size
  pixels = (torch.rand(n, 2) * (1 + margin) - margin * 0.5) * _image_extent(w, h)
  z = inverse_ndc_depth(torch.rand(n), camera_params.near_plane * 2, camera_params.far_plane)
  centres = _lift_to_world(pixels, z.unsqueeze(1), camera_params.T_image_world)

  focal = camera_params.T_image_camera[0, 0]
  world_size = (w / math.sqrt(n)) * (z / focal) * scale_factor        # one image-tiling footprint, seen at depth z
  log_scales = torch.randn(n, 3) * 0.5 + torch.log(world_size).unsqueeze(1)
  quaternions = _unit_rows((n, 4))
  opacity = _opacity(n, alpha_range)

  return Gaussians3D(position=centres, log_scaling=log_scales, rotation=quaternions,
                     alpha_logit=_logit(opacity).unsqueeze(1), feature=torch.rand(n, 3), batch_size=(n,))


def random_2d_gaussians(n, image_size: Tuple[int, int], num_channels=3, scale_factor=1.0,
                        alpha_range=(0.1, 0.9), depth_range=(0.0, 1.0)) -> Gaussians2D:
  """n screen-space gaussians, uniform over the image, sigma uniform in [0.2, 1.2) x scale_factor w / (1 + sqrt(n)).

  Draws, in order: rand(n, 2) position; rand(n, 1) depth; rand(n, 2) sigma; randn(n, 2) axis; rand(n) opacity;
  rand(n, num_channels) colour."""
  w, h = image_size
  centres = torch.rand(n, 2) * _image_extent(w, h)
  near, far = depth_range
  z = torch.rand((n, 1)) * (far - near) + near

  footprint = scale_factor * w / (1 + math.sqrt(n))
  sigmas = (torch.rand(n, 2) + 0.2) * footprint
  axis = torch.randn(n, 2)
  axis = axis / torch.norm(axis, dim=1, keepdim=True)
  opacity = _opacity(n, alpha_range)

  return Gaussians2D(position=centres, depths=z, log_scaling=torch.log(sigmas), rotation=axis,
                     alpha_logit=_logit(opacity), feature=torch.rand(n, num_channels), batch_size=(n,))
