"""2D harness helpers (reference ``misc/renderer2d.py:17-33,134-148``)."""
from __future__ import annotations

from numbers import Integral
from typing import Tuple

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
