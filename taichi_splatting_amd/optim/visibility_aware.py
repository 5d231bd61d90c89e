"""Visibility-aware Adam / LaProp (reference ``optim/visibility_aware.py:55-126``): the step weight of
a point is its visibility relative to a running (power-mean) visibility, gradients are normalised by
the visibility."""
from __future__ import annotations

from dataclasses import replace
from typing import Optional

import torch

from .fractional import ADAM, LAPROP, PointState, fused_update, make_group, saturate, weighted_step  # noqa: F401


def lerp(t, a, b):
  return a + (b - a) * t


def power_lerp(t, a, b, k=2):
  return lerp(t, a ** k, b ** k) ** (1 / k)


def update_visibility(running_vis: torch.Tensor, visibility: torch.Tensor, indexes: torch.Tensor,
                      total_weight: torch.Tensor, beta: float = 0.9, eps: float = 1e-12):
  updated_vis = power_lerp(beta, visibility, running_vis[indexes], k=4)
  running_vis[indexes] = updated_vis
  return visibility / torch.clamp_min(updated_vis, eps)


def set_indexes(target: torch.Tensor, values: torch.Tensor, indexes: torch.Tensor):
  result = torch.zeros_like(target)
  result[indexes] = values
  return result


class VisibilityOptimizer(torch.optim.Optimizer):
  def __init__(self, kind: int, params, lr=0.001, betas=(0.9, 0.999), eps=1e-16, vis_beta=0.9,
               vis_smooth: float = 0.01, bias_correction=True, grad_clip: Optional[float] = None):
    assert lr > 0, f"Invalid learning rate: {lr}"
    assert eps > 0, f"Invalid epsilon: {eps}"
    assert 0.0 <= betas[0] < 1.0, f"Invalid beta1: {betas[0]}"
    assert 0.0 <= betas[1] < 1.0, f"Invalid beta2: {betas[1]}"
    assert 0.0 <= vis_beta < 1.0, f"Invalid visibility beta: {vis_beta}"
    defaults = dict(lr=lr, betas=betas, eps=eps, mask_lr=None, point_lr=None, type="scalar",
                    bias_correction=bias_correction, clip=grad_clip)
    self.vis_beta = vis_beta
    self.vis_smooth = vis_smooth
    self.kind = kind
    super().__init__(params, defaults)

  @torch.no_grad()
  def step(self, indexes: torch.Tensor, visibility: torch.Tensor, basis: Optional[torch.Tensor] = None):
    assert visibility.shape == indexes.shape, f"shape mismatch {visibility.shape} != {indexes.shape}"
    groups = [make_group(group, self.state) for group in self.param_groups]
    n = groups[0].num_points

    shared = PointState(groups[0].state)
    total_weight = shared.per_point('total_weight', n, visibility.device)
    running_vis = shared.per_point('running_vis', n, visibility.device)

    weight = update_visibility(running_vis, visibility, indexes, total_weight, self.vis_beta)
    total_weight[indexes] += weight

    grad_scale = 1.0 / (visibility + self.vis_smooth)
    for group in groups:
      if group.grad is None:
        continue
      assert group.num_points == n, f"param shape {group.num_points} != {n}"
      # gradients are normalised by the point's visibility (reference :95-104): fused as a row scale
      fused_update(group, weight, indexes, total_weight, self.kind, basis, grad_scale=grad_scale)


class VisibilityAwareAdam(VisibilityOptimizer):
  def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-16, vis_beta=0.5,
               vis_smooth: float = 0.01, bias_correction=True, grad_clip: Optional[float] = None):
    super().__init__(ADAM, params, lr=lr, betas=betas, eps=eps, vis_beta=vis_beta, vis_smooth=vis_smooth,
                     bias_correction=bias_correction, grad_clip=grad_clip)


class VisibilityAwareLaProp(VisibilityOptimizer):
  def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-16, vis_beta=0.5,
               vis_smooth: float = 0.01, bias_correction=True, grad_clip: Optional[float] = None):
    super().__init__(LAPROP, params, lr=lr, betas=betas, eps=eps, vis_beta=vis_beta, vis_smooth=vis_smooth,
                     bias_correction=bias_correction, grad_clip=grad_clip)
