"""Multi-GPU rendering of ONE frame by screen-tile strips (new design; the reference is single GPU).

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  Gaussians are
replicated; rank r owns a contiguous strip of tile rows, maps/sorts/rasterizes only the overlaps of
its strip, and evaluates its part of the loss on its rows of the image.  Tiles are independent in
both raster passes, so the only exchange is the per-gaussian gradient sum: ONE all-reduce of the
2D-boundary gradient [d packed-2D (7) | d colour (F)] = 40 B x V for RGB, after which the
projection / SH backward runs replicated (payload 6x smaller than reducing the 3D/SH gradients).
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .data_types import Gaussians3D, RasterConfig
from .perspective import CameraParams


def strip_rows(tiles_high: int, world_size: int, rank: int,
               row_weights: Optional[Sequence[float]] = None) -> Tuple[int, int]:
  """Tile-row window [begin, end) of ``rank``.  With ``row_weights`` (e.g. overlaps per tile row,
  replicated on every rank, so no communication is needed to agree) the boundaries equalise the
  cumulative weight; otherwise rows are split evenly.  Strips are contiguous, disjoint and cover
  [0, tiles_high)."""
  assert 0 <= rank < world_size
  if row_weights is None:
    bounds = [(tiles_high * r) // world_size for r in range(world_size + 1)]
  else:
    w = torch.as_tensor(row_weights, dtype=torch.float64)
    assert w.shape[0] == tiles_high
    cum = torch.cumsum(w, 0)
    total = float(cum[-1]) if tiles_high > 0 else 0.0
    bounds = [0]
    for r in range(1, world_size):
      target = total * r / world_size
      b = int(torch.searchsorted(cum, torch.tensor(target, dtype=torch.float64)).item()) + 1 if total > 0 else (tiles_high * r) // world_size
      bounds.append(min(max(b, bounds[-1]), tiles_high))
    bounds.append(tiles_high)
  return bounds[rank], bounds[rank + 1]


def allreduce_boundary_grads(grad_points: torch.Tensor, grad_features: torch.Tensor,
                             group=None) -> Tuple[torch.Tensor, torch.Tensor]:
  """Sum the per-gaussian 2D-boundary gradients over all strips with a single collective."""
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
    return grad_points, grad_features
  buf = torch.cat([grad_points, grad_features], dim=1).contiguous()
  dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
  return buf[:, :grad_points.shape[1]], buf[:, grad_points.shape[1]:]


def render_strip_step(gaussians: Gaussians3D, camera_params: CameraParams, config: RasterConfig,
                      loss_fn: Callable[[torch.Tensor, Tuple[int, int]], torch.Tensor],
                      use_sh: bool = False, rank: Optional[int] = None, world_size: Optional[int] = None,
                      group=None, backward: bool = True):
  """One forward(+backward) step of the strip-sharded renderer on this rank.

  ``loss_fn(image, (row_begin_px, row_end_px))`` must return this rank's share of the loss computed
  from its rows of ``image`` (rows outside the strip are zero).  After the call, ``.grad`` of the
  leaf tensors in ``gaussians`` holds the FULL gradient (identical on every rank).
  Returns (Rendering of the strip, loss value of the strip).
  """
  from .perspective.projection import project_to_image
  from .renderer import render_projected
  from .spherical_harmonics import evaluate_sh_at

  if rank is None:
    rank = dist.get_rank(group) if dist.is_initialized() else 0
  if world_size is None:
    world_size = dist.get_world_size(group) if dist.is_initialized() else 1

  # per-gaussian stages: replicated
  gaussians2d, depths, indexes = project_to_image(gaussians, camera_params, config)
  if use_sh:
    features = evaluate_sh_at(gaussians.feature, gaussians.position.detach(), indexes, camera_params.camera_position, unique_indexes=True)
  else:
    features = gaussians.feature[indexes]

  # cut the autograd graph at the 2D boundary so the strip gradients can be reduced there
  g2 = gaussians2d.detach().requires_grad_(gaussians2d.requires_grad)
  f2 = features.detach().requires_grad_(features.requires_grad)

  ts = config.tile_size
  tiles_high = (camera_params.image_size[1] + ts - 1) // ts
  rows = strip_rows(tiles_high, world_size, rank)
  rendering = render_projected(indexes, g2, f2, depths.detach(), camera_params, config, tile_rows=rows)

  px_rows = (rows[0] * ts, min(rows[1] * ts, camera_params.image_size[1]))
  loss = loss_fn(rendering.image, px_rows)
  if backward and (g2.requires_grad or f2.requires_grad):
    loss.backward()
    gp = g2.grad if g2.grad is not None else torch.zeros_like(g2)
    gf = f2.grad if f2.grad is not None else torch.zeros_like(f2)
    gp, gf = allreduce_boundary_grads(gp, gf, group)
    tensors, grads = [], []
    if gaussians2d.requires_grad:
      tensors.append(gaussians2d); grads.append(gp)
    if features.requires_grad:
      tensors.append(features); grads.append(gf)
    torch.autograd.backward(tensors, grads)
  return rendering, loss.detach()
