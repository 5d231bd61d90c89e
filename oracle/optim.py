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
