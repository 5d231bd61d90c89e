"""Camera parameters (reference ``perspective/params.py:11-105``; same fields and helpers)."""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Optional, Tuple

import torch


@dataclass
class CameraParams:
  projection: torch.Tensor        # (4) - [fx, fy, cx, cy]
  T_camera_world: torch.Tensor    # (4, 4) camera view matrix

  near_plane: float
  far_plane: float
  image_size: Tuple[int, int]     # (W, H)

  id: Optional[int] = None

  def __post_init__(self):
    assert self.projection.shape == (4,), f"Expected shape (4,), got {self.projection.shape}"
    assert self.T_camera_world.shape == (4, 4), f"Expected shape (4, 4), got {self.T_camera_world.shape}"
    assert len(self.image_size) == 2
    assert self.near_plane > 0
    assert self.far_plane > self.near_plane

  @property
  def depth_range(self):
    return (self.near_plane, self.far_plane)

  @property
  def device(self):
    return self.projection.device

  @property
  def dtype(self):
    return self.projection.dtype

  @property
  def T_image_camera(self):
    fx, fy, cx, cy = self.projection
    m = [[fx, 0, cx],
         [0, fy, cy],
         [0, 0, 1]]
    return torch.tensor(m, device=self.device, dtype=self.dtype)

  @property
  def focal_length(self):
    return self.projection[0:2]

  @property
  def principal_point(self):
    return self.projection[2:4]

  @property
  def T_image_world(self):
    T_image_camera = torch.eye(4, device=self.device, dtype=self.dtype)
    T_image_camera[0:3, 0:3] = self.T_image_camera
    return T_image_camera @ self.T_camera_world

  @property
  def camera_position(self):
    T = self.T_camera_world
    if T.is_cuda and not T.requires_grad and T.dtype in (torch.float32, torch.float64) and tuple(T.shape) == (4, 4):
      # one tiny kernel instead of a device-side LU (~10 launches): ms_camera_position
      from .. import _lib
      lib = _lib.load()
      Tc = T.contiguous()
      out = torch.empty(3, dtype=T.dtype, device=T.device)
      _lib.check(lib.ms_camera_position(_lib.ptr(Tc), _lib.ptr(out), _lib.dtype_code(T.dtype),
                                        _lib.current_stream(T.device)), 'ms_camera_position')
      return out
    T_world_camera = torch.inverse(T)
    return T_world_camera[0:3, 3]

  def transformed(self, t: torch.Tensor) -> 'CameraParams':
    return replace(self, T_camera_world=t @ self.T_camera_world)

  def requires_grad_(self, requires_grad: bool = True):
    self.projection.requires_grad_(requires_grad)
    self.T_camera_world.requires_grad_(requires_grad)
    return self

  def detach(self):
    return replace(self, projection=self.projection.detach(),
                   T_camera_world=self.T_camera_world.detach())

  def scale_image(self, scale: float):
    image_size = (int(self.image_size[0] * scale), int(self.image_size[1] * scale))
    return replace(self, image_size=image_size, projection=self.projection * scale)

  def to(self, device=None, dtype=None):
    return CameraParams(
      id=self.id,
      projection=self.projection.to(device=device, dtype=dtype),
      T_camera_world=self.T_camera_world.to(device=device, dtype=dtype),
      near_plane=self.near_plane,
      far_plane=self.far_plane,
      image_size=self.image_size)

  def __repr__(self):
    w, h = self.image_size
    fx, fy, cx, cy = self.projection.detach().cpu().numpy()
    pos_str = ", ".join([f"{x:.3f}" for x in self.camera_position.detach().cpu()])
    return (f"CameraParams(id={self.id}, {w}x{h}, fx={fx:.4f}, fy={fy:.4f}, cx={cx:.4f}, cy={cy:.4f}, "
            f"clipping={self.near_plane:.4f}-{self.far_plane:.4f}, position=({pos_str}))")
