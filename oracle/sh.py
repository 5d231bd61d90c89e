"""Oracle: spherical-harmonics colour (torch, differentiable).

Follows ``evaluate_sh_at_kernel`` (indexed_spherical_harmonics.py:119-134) and the real SH basis
``rsh_cart_0..3`` (indexed_spherical_harmonics.py:38-106).
"""
from __future__ import annotations

import math

import torch


def rsh_cart(xyz: torch.Tensor, degree: int) -> torch.Tensor:
  x, y, z = xyz[..., 0], xyz[..., 1], xyz[..., 2]
  out = [torch.full_like(x, 0.282094791773878)]
  if degree >= 1:
    out += [-0.48860251190292 * y, 0.48860251190292 * z, -0.48860251190292 * x]
  if degree >= 2:
    x2, y2, z2 = x * x, y * y, z * z
    xy, xz, yz = x * y, x * z, y * z
    out += [1.09254843059208 * xy, -1.09254843059208 * yz,
            0.94617469575756 * z2 - 0.31539156525252, -1.09254843059208 * xz,
            0.54627421529604 * x2 - 0.54627421529604 * y2]
  if degree >= 3:
    out += [-0.590043589926644 * y * (3.0 * x2 - y2),
            2.89061144264055 * xy * z,
            0.304697199642977 * y * (1.5 - 7.5 * z2),
            1.24392110863372 * z * (1.5 * z2 - 0.5) - 0.497568443453487 * z,
            0.304697199642977 * x * (1.5 - 7.5 * z2),
            1.44530572132028 * z * (x2 - y2),
            -0.590043589926644 * x * (x2 - 3.0 * y2)]
  return torch.stack(out, dim=-1)


def evaluate_sh_at(params: torch.Tensor, points: torch.Tensor, indexes: torch.Tensor,
                   camera_pos: torch.Tensor) -> torch.Tensor:
  """params (M, K, D), points (M, 3), indexes (V,), camera_pos (3,) -> (V, K)"""
  d = params.shape[2]
  n = int(math.sqrt(d))
  assert n * n == d, f"SH feature count must be square, got {d}"
  degree = n - 1
  assert 0 <= degree <= 3
  dirs = points[indexes] - camera_pos.unsqueeze(0)
  dirs = dirs / torch.sqrt((dirs * dirs).sum(-1, keepdim=True))
  coeffs = rsh_cart(dirs, degree)                               # (V, D)
  out = (coeffs.unsqueeze(1) * params[indexes]).sum(-1)         # (V, K)
  return torch.clamp(out + 0.5, 0., 1.)
