"""Pinhole camera of a frame: intrinsics ``[fx, fy, cx, cy]``, the world -> camera matrix, clip planes and image size.

Field names, properties and methods are the ones callers of reference ``perspective/params.py:11-105`` use
(SURVEY.md appendix D); the bodies are this package's.  ``camera_position`` runs one 4x4 Gauss-Jordan kernel
(``ms_camera_position``) instead of a device-side LU when no gradient is asked of the pose.
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Optional, Tuple

import torch


@dataclass
class CameraParams:
  projection: torch.Tensor        # (4,)   fx, fy, cx, cy in pixels
  T_camera_world: torch.Tensor    # (4, 4) world -> camera

  near_plane: float
  far_plane: float
  image_size: Tuple[int, int]     # (W, H)

  id: Optional[int] = None

  def __post_init__(self):
    if tuple(self.projection.shape) != (4,):
      raise AssertionError(f"Expected shape (4,), got {self.projection.shape}")
    if tuple(self.T_camera_world.shape) != (4, 4):
      raise AssertionError(f"Expected shape (4, 4), got {self.T_camera_world.shape}")
    assert len(self.image_size) == 2, f"image_size is (W, H), got {self.image_size}"
    assert 0 < self.near_plane < self.far_plane, f"clip planes must satisfy 0 < near < far, got {self.depth_range}"

  # ---- where the tensors live -----------------------------------------------------------------------------------
  @property
  def device(self):
    return self.projection.device

  @property
  def dtype(self):
    return self.projection.dtype

  def to(self, device=None, dtype=None) -> 'CameraParams':
    moved = {name: getattr(self, name).to(device=device, dtype=dtype) for name in ('projection', 'T_camera_world')}
    return replace(self, **moved)

  def detach(self) -> 'CameraParams':
    return replace(self, projection=self.projection.detach(), T_camera_world=self.T_camera_world.detach())

  def requires_grad_(self, requires_grad: bool = True) -> 'CameraParams':
    for t in (self.projection, self.T_camera_world):
      t.requires_grad_(requires_grad)
    return self

  # ---- intrinsics -----------------------------------------------------------------------------------------------
  @property
  def depth_range(self) -> Tuple[float, float]:
    return (self.near_plane, self.far_plane)

  @property
  def focal_length(self) -> torch.Tensor:
    return self.projection[:2]

  @property
  def principal_point(self) -> torch.Tensor:
    return self.projection[2:]

  @property
  def T_image_camera(self) -> torch.Tensor:
    """3x3 intrinsic matrix K (a fresh tensor: not differentiable w.r.t. ``projection``, as in the reference)"""
    K = torch.eye(3, device=self.device, dtype=self.dtype)
    K[0, 0], K[1, 1] = self.projection[0], self.projection[1]
    K[0, 2], K[1, 2] = self.projection[2], self.projection[3]
    return K

  @property
  def T_image_world(self) -> torch.Tensor:
    """4x4 world -> homogeneous image coordinates: [K 0; 0 1] @ T_camera_world"""
    K4 = torch.eye(4, device=self.device, dtype=self.dtype)
    K4[:3, :3] = self.T_image_camera
    return K4 @ self.T_camera_world

  def scale_image(self, scale: float) -> 'CameraParams':
    w, h = self.image_size
    return replace(self, image_size=(int(w * scale), int(h * scale)), projection=self.projection * scale)

  # ---- pose -----------------------------------------------------------------------------------------------------
  @property
  def camera_position(self) -> torch.Tensor:
    """translation of the camera -> world transform = inverse(T_camera_world)[:3, 3]"""
    T = self.T_camera_world
    differentiable = T.requires_grad and torch.is_grad_enabled()
    if T.is_cuda and not differentiable and T.dtype in (torch.float32, torch.float64):
      from .. import _lib
      position = torch.empty(3, dtype=T.dtype, device=T.device)
      _lib.check(_lib.load().ms_camera_position(_lib.ptr(T.detach().contiguous()), _lib.ptr(position),
                                                _lib.dtype_code(T.dtype), _lib.current_stream(T.device)),
                 'ms_camera_position')
      return position
    return torch.inverse(T)[:3, 3]

  def transformed(self, t: torch.Tensor) -> 'CameraParams':
    """the same camera after moving the world by ``t`` (left-multiplies the view matrix)"""
    return replace(self, T_camera_world=t @ self.T_camera_world)

  def __repr__(self):
    w, h = self.image_size
    fx, fy, cx, cy = (float(x) for x in self.projection.detach().cpu())
    x, y, z = (float(v) for v in self.camera_position.detach().cpu())
    return (f"CameraParams(id={self.id}, {w}x{h}, fx={fx:.4f}, fy={fy:.4f}, cx={cx:.4f}, cy={cy:.4f}, "
            f"clipping={self.near_plane:.4f}-{self.far_plane:.4f}, position=({x:.3f}, {y:.3f}, {z:.3f}))")
