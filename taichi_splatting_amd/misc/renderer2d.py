"""2D harness: packing, rendering and the toy densification helpers of the image-fitting demo
(reference ``misc/renderer2d.py:17-148``, used by ``examples/fit_image_gaussians.py``).  Pure torch on top of
``rasterize``; no kernels of their own."""
from __future__ import annotations

from numbers import Integral
from typing import Tuple

import math
from typing import Optional

import torch

from ..data_types import Gaussians2D, RasterConfig
from ..rasterizer import rasterize


def project_gaussians2d(points: Gaussians2D) -> torch.Tensor:
  """Pack Gaussians2D parameters into the (N, 7) representation [mean2, axis2, sigma2, alpha]
  used by the tile mapper and the rasterizer."""
  alpha = torch.sigmoid(points.alpha_logit)
  sigma = points.scaling
  v1 = points.rotation / torch.norm(points.rotation, dim=1, keepdim=True)
  if alpha.ndim == 1:
    alpha = alpha.unsqueeze(1)
  return torch.cat([points.position, v1, sigma, alpha], dim=-1)


def render_gaussians(gaussians: Gaussians2D, image_size: Tuple[Integral, Integral],
                     raster_config: RasterConfig = RasterConfig()):
  gaussians2d = project_gaussians2d(gaussians)
  return rasterize(gaussians2d=gaussians2d, depth=gaussians.depths.clamp(0, 1),
                   features=gaussians.feature, image_size=image_size, config=raster_config)


# ---- densification helpers (reference misc/renderer2d.py:36-131) ---------------------------------------

def _unit_axes(points: Gaussians2D):
  v1 = points.rotation / torch.norm(points.rotation, dim=1, keepdim=True)
  v2 = torch.stack([-v1[:, 1], v1[:, 0]], dim=-1)
  return v1, v2


def point_rotation(points: Gaussians2D) -> torch.Tensor:
  """(N, 2, 2) rotation with the major axis and its perpendicular as rows (reference :47-52)."""
  return torch.stack(_unit_axes(points), dim=1)


def point_basis(points: Gaussians2D, eps: float = 1e-4) -> torch.Tensor:
  """(N, 2, 2) matrix whose COLUMNS are the two axes scaled by their (clamped) sigmas, so that
  ``basis @ z`` maps a unit-normal sample z to an offset distributed like the gaussian (reference :36-43)."""
  v1, v2 = _unit_axes(points)
  sigma = torch.clamp_min(points.scaling, eps)
  return torch.stack([v1 * sigma[:, 0:1], v2 * sigma[:, 1:2]], dim=2)


def point_covariance(points: Gaussians2D) -> torch.Tensor:
  basis = point_basis(points)
  return basis @ basis.transpose(1, 2)


def sample_gaussians(points: Gaussians2D) -> torch.Tensor:
  """One offset per gaussian drawn from it (reference :96-98)."""
  z = torch.randn_like(points.position)
  return (point_basis(points) @ z.unsqueeze(2)).squeeze(2)


def repeat_sample_gaussians(samples: torch.Tensor, points: Gaussians2D, n: int = 2) -> torch.Tensor:
  """Map unit-space samples (N, n, 2) through each gaussian's basis: offsets (N, n, 2) (reference :101-103)."""
  basis = point_basis(points)
  return torch.einsum('nij,nkj->nki', basis, samples.reshape(-1, n, 2))


def split_with_offsets(points: Gaussians2D, offsets: torch.Tensor, depth_noise: float = 1e-2) -> Gaussians2D:
  """Replace every gaussian by n copies displaced by ``offsets`` (N, n, 2), depths jittered (reference :56-67)."""
  num_points, n, _ = offsets.shape
  copies = points.apply(lambda t: torch.repeat_interleave(t, repeats=n, dim=0), batch_size=[num_points * n])
  depths = copies.depths + torch.randn_like(copies.depths) * depth_noise
  return copies.replace(position=copies.position + offsets.reshape(-1, 2), depths=torch.clamp_min(depths, 1e-6))


def split_gaussians2d(points: Gaussians2D, n: int = 2, scaling: Optional[float] = None,
                      depth_noise: float = 1e-2) -> Gaussians2D:
  """Random split: n children sampled at half the parent's spread, scaled by ``scaling`` (default
  1/sqrt(n)) (reference :70-93)."""
  z = 0.5 * torch.randn((points.position.shape[0], n, 2), device=points.position.device, dtype=points.position.dtype)
  offsets = repeat_sample_gaussians(z, points, n)
  if scaling is None:
    scaling = 1.0 / math.sqrt(n)
  shrunk = points.replace(log_scaling=points.log_scaling + math.log(scaling))
  return split_with_offsets(shrunk, offsets, depth_noise)


def uniform_split_gaussians2d(points: Gaussians2D, n: int = 2, scaling: Optional[float] = None,
                              depth_noise: float = 1e-2, sep: float = 0.7, random_axis: bool = False,
                              eps: float = 1e-6) -> Gaussians2D:
  """Deterministic split: n children evenly spaced in [-sep, sep] sigmas along ONE axis (the longer
  one, or one drawn with probability proportional to the sigmas), that axis shrunk by ``scaling``
  (default sqrt(n)/n) (reference :106-131)."""
  if random_axis:
    probs = torch.nn.functional.normalize(points.scaling + eps, p=1, dim=1)
    axis = torch.multinomial(probs, num_samples=1).squeeze(1)
  else:
    axis = torch.argmax(points.log_scaling, dim=1)
  onehot = torch.nn.functional.one_hot(axis, num_classes=2).to(points.position.dtype)
  steps = torch.linspace(-sep, sep, n, device=points.position.device, dtype=points.position.dtype)
  z = steps.view(1, n, 1) * onehot.view(-1, 1, 2)
  offsets = repeat_sample_gaussians(z, points, n)      # offsets use the parent's sigmas
  if scaling is None:
    scaling = math.sqrt(n) / n
  shrunk = points.set_scaling(points.scaling * (onehot * scaling + (1 - onehot)))
  return split_with_offsets(shrunk, offsets, depth_noise)
