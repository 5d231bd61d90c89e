"""Oracle: per-point moment update of the fractional Adam / LaProp optimisers (torch, CPU).

Follows the Taichi kernels of optim/fractional_adam.py:8-86 and optim/fractional_laprop.py:8-86
(``lerp(t, a, b) = a t + b (1 - t)``, taichi_lib/generic.py:488-490) and the host logic of
optim/fractional.py:108-156,176-195.  The reference has no test for these kernels (parity unpinned by
reference data); the restatement is pinned by the identity "weight 1 on every point == torch.optim.Adam
step scaled by saturate(1) = 1 - exp(-2)" (tests/test_optim.py).
"""
from __future__ import annotations

import torch


def lerp(t, a, b):
  return a * t + b * (1.0 - t)


def fractional_step(kind, vector, indexes, weight, m, v, total_weight, grad, lr, betas, eps, bias_correction):
  """Returns lr_step (M, D); updates m, v in place.  kind 0 = Adam, 1 = LaProp."""
  beta1, beta2 = betas
  w = weight.unsqueeze(1)
  tw = total_weight[indexes].unsqueeze(1)
  g = grad[indexes]
  bias1 = 1.0 - beta1 ** tw if bias_correction else torch.ones_like(tw)
  bias2 = 1.0 - beta2 ** tw if bias_correction else torch.ones_like(tw)
  if vector:
    norm = (g * g).sum(1, keepdim=True)
    v_new = lerp(beta2 ** w, v[indexes].unsqueeze(1), norm)
  else:
    v_new = lerp(beta2 ** w, v[indexes], g * g)
  if kind == 0:
    m_new = lerp(beta1 ** w, m[indexes], g)
    bias_factor = torch.sqrt(bias2) / bias1 if bias_correction else torch.ones_like(tw)
    step = m_new / torch.clamp_min(torch.sqrt(v_new), eps) * bias_factor * lr
  else:
    m_new = lerp(beta1 ** w, m[indexes], g / torch.clamp_min(torch.sqrt(v_new / bias2), eps))
    step = m_new * lr / bias1
  m[indexes] = m_new
  v[indexes] = v_new.squeeze(1) if vector else v_new
  return step


def group_update(kind, group_type, param, grad, m, v, indexes, weight, total_weight, lr, betas, eps,
                 bias_correction, grad_scale=None, basis=None, clip=None, mask_lr=None, point_lr=None):
  """One parameter group's step on the visible rows, host logic of optim/fractional.py:108-156,190-195
  (``weighted_step`` + ``param[indexes] -= lr_step * saturate(weight)``) with the gradient pre-scaling of
  optim/visibility_aware.py:95-104 (``grad_scale`` = 1 / (visibility + vis_smooth)).  Updates param, m, v
  in place.  group_type: 'scalar' | 'vector' | 'local_vector'."""
  grad = grad.clone()
  if grad_scale is not None:
    grad[indexes] = grad[indexes] * grad_scale.unsqueeze(1)
  if group_type == 'local_vector':
    inv_basis = torch.linalg.inv(basis)
    grad[indexes] = torch.einsum('bij,bj->bi', inv_basis, grad[indexes])
  step = fractional_step(kind, group_type != 'scalar', indexes, weight, m, v, total_weight, grad, lr, betas, eps,
                         bias_correction)
  if clip is not None:
    step = step.clamp(-lr * clip, lr * clip)
  if group_type == 'local_vector':
    step = torch.einsum('bij,bj->bi', basis, step)
  if mask_lr is not None:
    step = step * mask_lr.view(-1).unsqueeze(0)
  if point_lr is not None:
    step = step * point_lr[indexes].unsqueeze(1)
  step = torch.where(torch.isfinite(step), step, torch.zeros_like(step))
  param[indexes] -= step * (1 - 1 / torch.exp(2 * weight)).unsqueeze(1)
  return step


def visibility_weights(running_vis, visibility, indexes, total_weight, beta, vis_smooth, eps=1e-12):
  """optim/visibility_aware.py:35-52 (update_visibility: power_lerp of order 4 between the new visibilities and the
  running ones) and :86-97 of the step: returns (weight, grad_scale) and updates running_vis / total_weight in place."""
  k = 4
  a, b = visibility, running_vis[indexes]
  updated = (a ** k + (b ** k - a ** k) * beta) ** (1 / k)          # lerp(t, x, y) = x + (y - x) t, t = beta
  running_vis[indexes] = updated
  weight = visibility / torch.clamp_min(updated, eps)
  total_weight[indexes] += weight
  return weight, 1.0 / (visibility + vis_smooth)
