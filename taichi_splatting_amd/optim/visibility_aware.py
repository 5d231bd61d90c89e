"""Visibility-aware Adam / LaProp (reference ``optim/visibility_aware.py:55-126``): the step weight of
a point is its visibility relative to a running (power-mean) visibility, gradients are normalised by
the visibility."""
from __future__ import annotations

from dataclasses import replace
from typing import Optional

import torch

from .. import _lib
from .fractional import (ADAM, LAPROP, PointState, fused_update, fused_update_groups, make_group, saturate,  # noqa: F401
                         weighted_step)


def _track_visibility(running: torch.Tensor, seen: torch.Tensor, indexes: torch.Tensor, beta: float,
                      order: int = 4, floor: float = 1e-12) -> torch.Tensor:
  """Fold this step's visibilities into the running ones and return each point's step weight.

  The running value is an exponential moving POWER mean (order 4, i.e. biased towards a point's better views):
  ``r <- ((1 - beta) v^4 + beta r^4)^(1/4)`` for the points listed in ``indexes``; the weight is the visibility
  relative to it, ``v / max(r, floor)`` (reference optim/visibility_aware.py:25-41)."""
  previous = running[indexes]
  mixed = (seen ** order + (previous ** order - seen ** order) * beta) ** (1 / order)
  running[indexes] = mixed
  return seen / torch.clamp_min(mixed, floor)


# The reference's module-level helpers (optim/visibility_aware.py:11-52): same names, argument order and results, as
# thin wrappers, so that ``from taichi_splatting.optim.visibility_aware import update_visibility`` keeps working.
def get_running_vis(state: dict, n: int, device: torch.device) -> torch.Tensor:
  return PointState(state).per_point('running_vis', n, device)


def exp_lerp(t, a, b):
  """``log(lerp(t, exp(a), exp(b)))`` without overflow: interpolation of two log-domain values in the linear domain
  (reference optim/visibility_aware.py:19-22; plain eager torch here, the reference compiles it)."""
  top = torch.maximum(a, b)
  return top + torch.log(torch.lerp(torch.exp(a - top), torch.exp(b - top), t))


def lerp(t, a, b):
  """Value a fraction ``t`` of the way from ``a`` to ``b``."""
  return torch.lerp(torch.as_tensor(a), torch.as_tensor(b), t) if torch.is_tensor(t) else a + t * (b - a)


def max_decaying(t, a, b):
  return torch.maximum(a, lerp(t, a, b))


def power_lerp(t, a, b, k=2):
  """Interpolation of the k-th powers, back at the first power: a power mean of the end points."""
  return lerp(t, a ** k, b ** k) ** (1 / k)


def update_visibility(running_vis: torch.Tensor, visibility: torch.Tensor, indexes: torch.Tensor,
                      total_weight: torch.Tensor, beta: float = 0.9, eps: float = 1e-12) -> torch.Tensor:
  """Updates ``running_vis[indexes]`` in place and returns the step weights (``total_weight`` is unused, as in the
  reference)."""
  return _track_visibility(running_vis, visibility, indexes, beta, floor=eps)


def set_indexes(target: torch.Tensor, values: torch.Tensor, indexes: torch.Tensor) -> torch.Tensor:
  """A zero tensor shaped like ``target`` carrying ``values`` at ``indexes``."""
  return torch.zeros_like(target).index_put_((indexes,), values)


class VisibilityOptimizer(torch.optim.Optimizer):
  def __init__(self, kind: int, params, lr=0.001, betas=(0.9, 0.999), eps=1e-16, vis_beta=0.9,
               vis_smooth: float = 0.01, bias_correction=True, grad_clip: Optional[float] = None):
    for name, value, ok in (("lr", lr, lr > 0), ("eps", eps, eps > 0), ("betas[0]", betas[0], 0.0 <= betas[0] < 1.0),
                            ("betas[1]", betas[1], 0.0 <= betas[1] < 1.0), ("vis_beta", vis_beta, 0.0 <= vis_beta < 1.0)):
      if not ok:
        raise AssertionError(f"VisibilityOptimizer: {name} = {value} is out of range")
    defaults = dict(lr=lr, betas=betas, eps=eps, mask_lr=None, point_lr=None, type="scalar",
                    bias_correction=bias_correction, clip=grad_clip)
    self.vis_beta = vis_beta
    self.vis_smooth = vis_smooth
    self.kind = kind
    super().__init__(params, defaults)

  @torch.no_grad()
  def step(self, indexes: Optional[torch.Tensor], visibility: torch.Tensor, basis: Optional[torch.Tensor] = None,
           visible_threshold: float = 1e-8):
    """Reference ``step(indexes, visibility, basis)`` (optim/visibility_aware.py:78-104): two launches — the step weights
    (running visibility, total weight, gradient scale: ``ms_optim_visibility_weights``) and all parameter groups
    (``ms_optim_step_groups``).

    ``indexes=None`` is this library's DENSE mode: ``visibility`` then holds one value per point, and the points with
    ``visibility <= visible_threshold`` are skipped on the device — same update as
    ``visible = (visibility > 1e-8).nonzero().squeeze(1); step(visible, visibility[visible])`` of the reference's
    training loop (examples/fit_image_gaussians.py:118-125) without its host synchronisation.  (``basis`` rows are
    per point in that mode.)"""
    groups = [make_group(group, self.state) for group in self.param_groups]
    n = groups[0].num_points
    if indexes is not None:
      assert visibility.shape == indexes.shape, f"shape mismatch {visibility.shape} != {indexes.shape}"
      assert indexes.dtype == torch.int64
      indexes = indexes.contiguous()
    else:
      assert visibility.shape == (n,), f"dense mode: one visibility per point expected, got {tuple(visibility.shape)}"

    shared = PointState(groups[0].state)
    total_weight = shared.per_point('total_weight', n, visibility.device)
    running_vis = shared.per_point('running_vis', n, visibility.device)
    for group in groups:
      assert group.grad is None or group.num_points == n, f"param shape {group.num_points} != {n}"

    _lib.require_gpu(visibility, indexes)
    vis = visibility.detach().to(torch.float32).contiguous()
    count = vis.shape[0]
    weight, grad_scale = torch.empty_like(vis), torch.empty_like(vis)
    # gradients are normalised by the point's visibility (reference :95-104): a row scale inside the group kernel
    _lib.check(_lib.load().ms_optim_visibility_weights(_lib.ptr(indexes), _lib.ptr(vis), count, float(self.vis_beta),
                                                       float(self.vis_smooth), 1e-12, float(visible_threshold),
                                                       _lib.ptr(running_vis), _lib.ptr(total_weight), _lib.ptr(weight),
                                                       _lib.ptr(grad_scale), _lib.current_stream(vis.device)),
               "visibility-aware step weights")
    fused_update_groups(groups, weight, indexes, total_weight, self.kind, basis, grad_scale=grad_scale)


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
