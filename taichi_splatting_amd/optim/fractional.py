"""Fractional (per-point weighted) Adam / LaProp and their sparse variants.

Same classes, arguments and update rule as reference ``optim/fractional.py:113-230``; the per-point
moment update (``fractional_adam.py`` / ``fractional_laprop.py`` Taichi kernels) is the gfx950 kernel
``ms_fractional_step`` (csrc/optim.hip) reached through the C-ABI.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from .. import _lib


class PointState:
  """Lazily created per-point optimiser state inside a torch ``Optimizer.state`` entry.

  Keys and shapes are the reference's (``optim/util.py``), including its naming quirk, so state dicts stay
  interchangeable: ``'v'`` is the (N, D) FIRST moment, ``'m'`` the second moment — (N, D) for scalar groups,
  (N,) for vector groups (one second moment per point from the squared gradient norm); ``'total_weight'`` and
  ``'running_vis'`` are (N,) float32 and live in the FIRST parameter group's state."""

  def __init__(self, state: dict):
    self.state = state

  def _get(self, key: str, make):
    if key not in self.state:
      self.state[key] = make()
    return self.state[key]

  def moments(self, param: torch.Tensor, per_point_second_moment: bool):
    # the pair is created TOGETHER, keyed on the first moment alone (reference optim/util.py:5-18): a state dict
    # restored with only 'm' gets both reset, one restored with only 'v' fails on the missing 'm' as it does there
    if 'v' not in self.state:
      rows = param.view(param.shape[0], -1)
      self.state['v'] = torch.zeros_like(rows)
      self.state['m'] = rows.new_zeros((rows.shape[0],)) if per_point_second_moment else torch.zeros_like(rows)
    return self.state['v'], self.state['m']

  def per_point(self, key: str, shape, device) -> torch.Tensor:
    dims = tuple(shape) if isinstance(shape, (tuple, list, torch.Size)) else (int(shape),)
    return self._get(key, lambda: torch.zeros(dims, dtype=torch.float32, device=device))


class restore_grad:
  """``with restore_grad(a, b): ...`` — inside the block the tensors that require grad start from a zero
  ``.grad``; on exit every tensor gets back the ``.grad`` it had before (reference ``optim/autograd.py``: used to
  take a side gradient, e.g. of a regulariser, without disturbing the accumulated one)."""

  def __init__(self, *tensors: torch.Tensor):
    self.tensors = tensors
    self.saved = None

  def __enter__(self):
    self.saved = [t.grad for t in self.tensors]
    for t in self.tensors:
      if t.requires_grad:
        t.grad = torch.zeros_like(t)
    return self

  def __exit__(self, *exc):
    for t, g in zip(self.tensors, self.saved):
      t.grad = g
    return False

ADAM, LAPROP = 0, 1


@dataclass
class Group:
  name: str
  type: str
  param: torch.Tensor
  grad: Optional[torch.Tensor]
  state: dict
  lr: float
  betas: Tuple[float, float]
  eps: float
  bias_correction: bool
  clip: Optional[float]
  mask_lr: Optional[torch.Tensor]
  point_lr: Optional[torch.Tensor]

  @property
  def num_points(self):
    return self.param.shape[0]


def make_group(group, state) -> Group:
  n = len(group["params"])
  assert n == 1, f"expected 1 tensor in group {group['name']}, got {n}"
  params = group["params"][0]
  state = state[params]
  return Group(
    name=group["name"], type=group["type"],
    param=params.view(params.shape[0], -1),
    grad=params.grad.view(params.shape[0], -1) if params.grad is not None else None,
    state=state, lr=group["lr"], betas=group["betas"], eps=group["eps"],
    bias_correction=group["bias_correction"], clip=group.get("clip", None),
    mask_lr=group["mask_lr"], point_lr=group["point_lr"])


def fractional_step(kind: int, vector: bool, lr_step, indexes, weight, m, v, total_weight, grad, lr,
                    betas, eps, bias_correction):
  """Launch ``ms_fractional_step``: updates m, v in place, writes lr_step (M, D)."""
  lib = _lib.load()
  _lib.require_gpu(lr_step, indexes, weight, m, v, total_weight, grad)
  for t in (lr_step, weight, m, v, total_weight, grad):
    assert t.dtype == torch.float32 and t.is_contiguous(), "fractional optimisers run in contiguous float32"
  assert indexes.dtype == torch.int64 and indexes.is_contiguous()
  _lib.check(lib.ms_fractional_step(kind, int(vector), lr_step.data_ptr(), indexes.data_ptr(), weight.data_ptr(),
                                    m.data_ptr(), v.data_ptr(), total_weight.data_ptr(), grad.data_ptr(),
                                    indexes.shape[0], lr_step.shape[1], float(lr), float(betas[0]), float(betas[1]),
                                    float(eps), int(bias_correction), _lib.current_stream(grad.device)),
             "fractional optimizer step")


def weighted_step(group: Group, visible_weight: torch.Tensor, visible_indexes: torch.Tensor,
                  total_weight: torch.Tensor, kind: int, basis: Optional[torch.Tensor] = None):
  """reference optim/fractional.py:108-156"""
  if group.type not in GROUP_TYPES:
    raise ValueError(f"unknown group type {group.type}")
  vector = group.type != "scalar"
  m, v = PointState(group.state).moments(group.param, per_point_second_moment=vector)

  grad = group.grad
  if group.type == "local_vector":
    assert basis is not None, "basis is required for local_vector optimizer"
    inv_basis = torch.linalg.inv(basis)
    grad[visible_indexes] = torch.einsum('bij,bj->bi', inv_basis, grad[visible_indexes])

  lr_step = group.param.new_zeros(visible_indexes.shape[0], group.param.shape[1])
  fractional_step(kind, vector, lr_step, visible_indexes.contiguous(), visible_weight.contiguous(), m, v,
                  total_weight, grad.contiguous(), group.lr, group.betas, group.eps, group.bias_correction)

  if group.clip is not None:
    max_step = group.lr * group.clip
    lr_step.clamp_(-max_step, max_step)
  if group.type == "local_vector":
    lr_step = torch.einsum('bij,bj->bi', basis, lr_step)
  if group.mask_lr is not None:
    lr_step *= group.mask_lr.view(-1).unsqueeze(0)
  if group.point_lr is not None:     # per row learning rate
    lr_step *= group.point_lr[visible_indexes].unsqueeze(1)

  lr_step[~lr_step.isfinite()] = 0.0
  return lr_step


GROUP_TYPES = {"scalar": 0, "vector": 1, "local_vector": 2}


def fused_update(group: Group, visible_weight: torch.Tensor, visible_indexes: torch.Tensor,
                 total_weight: torch.Tensor, kind: int, basis: Optional[torch.Tensor] = None,
                 grad_scale: Optional[torch.Tensor] = None):
  """The whole per-group step — ``weighted_step`` + ``param[indexes] -= lr_step * saturate(weight)``
  (reference optim/fractional.py:108-156,190-195) — as ONE kernel, ``ms_fractional_update``."""
  if group.type not in GROUP_TYPES:
    raise ValueError(f"unknown group type {group.type}")
  m, v = PointState(group.state).moments(group.param, per_point_second_moment=group.type != "scalar")
  if group.type == "local_vector":
    assert basis is not None, "basis is required for local_vector optimizer"
    d = group.param.shape[1]
    assert tuple(basis.shape) == (visible_indexes.shape[0], d, d), f"basis must be (M, {d}, {d}), got {tuple(basis.shape)}"
    basis = basis.detach().to(torch.float32).contiguous()
  else:
    basis = None

  lib = _lib.load()
  param, grad = group.param, group.grad
  _lib.require_gpu(param, grad, visible_indexes, visible_weight, total_weight)
  assert param.is_contiguous() and param.dtype == torch.float32, "fractional optimisers run in contiguous float32"
  grad = grad.contiguous()
  indexes = visible_indexes.contiguous()
  weight = visible_weight.to(torch.float32).contiguous()
  assert indexes.dtype == torch.int64
  gs = grad_scale.to(torch.float32).contiguous() if grad_scale is not None else None
  mask_lr = group.mask_lr.to(device=param.device, dtype=torch.float32).reshape(-1).contiguous() if group.mask_lr is not None else None
  if mask_lr is not None:
    assert mask_lr.shape[0] == param.shape[1], f"mask_lr must have {param.shape[1]} entries"
  point_lr = group.point_lr.to(torch.float32).contiguous() if group.point_lr is not None else None
  _lib.check(lib.ms_fractional_update(kind, GROUP_TYPES[group.type], _lib.ptr(param), _lib.ptr(grad), _lib.ptr(m), _lib.ptr(v),
                                      _lib.ptr(indexes), _lib.ptr(weight), _lib.ptr(total_weight), _lib.ptr(gs),
                                      _lib.ptr(basis), _lib.ptr(mask_lr), _lib.ptr(point_lr), indexes.shape[0],
                                      param.shape[1], float(group.lr), float(group.betas[0]), float(group.betas[1]),
                                      float(group.eps), float(group.clip) if group.clip is not None else -1.0,
                                      int(group.bias_correction), _lib.current_stream(param.device)),
             "fractional optimizer update")


def _group_args(group: Group, indexes_count: Optional[int], basis: Optional[torch.Tensor], keep: list) -> '_lib.OptimGroupC':
  """One ``ms_optim_group`` (include/mi355_splat.h); tensors made here are appended to ``keep`` so that they outlive the
  launch."""
  if group.type not in GROUP_TYPES:
    raise ValueError(f"unknown group type {group.type}")
  m, v = PointState(group.state).moments(group.param, per_point_second_moment=group.type != "scalar")
  param, grad = group.param, group.grad
  d = param.shape[1]
  _lib.require_gpu(param, grad)
  assert param.is_contiguous() and param.dtype == torch.float32, "fractional optimisers run in contiguous float32"
  grad = grad.contiguous()
  b = None
  if group.type == "local_vector":
    assert basis is not None, "basis is required for local_vector optimizer"
    assert indexes_count is None or tuple(basis.shape) == (indexes_count, d, d), \
      f"basis must be (M, {d}, {d}), got {tuple(basis.shape)}"
    b = basis.detach().to(torch.float32).contiguous()
  mask_lr = group.mask_lr.to(device=param.device, dtype=torch.float32).reshape(-1).contiguous() if group.mask_lr is not None else None
  if mask_lr is not None:
    assert mask_lr.shape[0] == d, f"mask_lr must have {d} entries"
  point_lr = group.point_lr.to(torch.float32).contiguous() if group.point_lr is not None else None
  keep.extend((grad, b, mask_lr, point_lr, m, v))
  g = _lib.OptimGroupC(group_type=GROUP_TYPES[group.type], param=_lib.ptr(param), grad=_lib.ptr(grad), m=_lib.ptr(m), v=_lib.ptr(v),
                       basis=_lib.ptr(b), mask_lr=_lib.ptr(mask_lr), point_lr=_lib.ptr(point_lr), d=d,
                       bias_correction=int(group.bias_correction), lr=float(group.lr), beta1=float(group.betas[0]),
                       beta2=float(group.betas[1]), eps=float(group.eps),
                       clip=float(group.clip) if group.clip is not None else -1.0, reserved=0.0)
  g.struct_size = ctypes.sizeof(_lib.OptimGroupC)
  return g


def fused_update_groups(groups, visible_weight: torch.Tensor, visible_indexes: Optional[torch.Tensor],
                        total_weight: torch.Tensor, kind: int, basis: Optional[torch.Tensor] = None,
                        grad_scale: Optional[torch.Tensor] = None):
  """Every parameter group of a step in ONE launch (``ms_optim_step_groups``, csrc/optim.hip): what the reference loops
  over on the host (optim/fractional.py:176-195).  ``visible_indexes`` None = dense mode: row i is point i and rows with
  a negative weight are skipped (``ms_optim_visibility_weights`` marks invisible points that way)."""
  groups = [g for g in groups if g.grad is not None]
  if not groups:
    return
  lib = _lib.load()
  _lib.require_gpu(visible_weight, total_weight, visible_indexes)
  weight = visible_weight.to(torch.float32).contiguous()
  indexes = None
  if visible_indexes is not None:
    indexes = visible_indexes.contiguous()
    assert indexes.dtype == torch.int64
  gs = grad_scale.to(torch.float32).contiguous() if grad_scale is not None else None
  keep = []
  count = weight.shape[0]
  array = (_lib.OptimGroupC * len(groups))(*[_group_args(g, count, basis, keep) for g in groups])
  _lib.check(lib.ms_optim_step_groups(kind, array, len(groups), _lib.ptr(indexes), _lib.ptr(weight), _lib.ptr(total_weight),
                                      _lib.ptr(gs), count, _lib.current_stream(weight.device)), "fractional optimizer step")


def saturate(x: torch.Tensor):
  return 1 - 1 / torch.exp(2 * x)


class FractionalOpt(torch.optim.Optimizer):
  def __init__(self, kind: int, param_groups, lr=0.001, betas=(0.9, 0.999), eps=1e-16,
               bias_correction=True, clip: Optional[float] = None):
    assert lr > 0, f"Invalid learning rate: {lr}"
    assert eps > 0, f"Invalid epsilon: {eps}"
    assert 0.0 <= betas[0] < 1.0, f"Invalid beta1: {betas[0]}"
    assert 0.0 <= betas[1] < 1.0, f"Invalid beta2: {betas[1]}"
    defaults = dict(lr=lr, betas=betas, eps=eps, mask_lr=None, point_lr=None, type="scalar",
                    bias_correction=bias_correction, clip=clip)
    self.kind = kind
    super().__init__(param_groups, defaults)

  @torch.no_grad()
  def step(self, indexes: torch.Tensor, weight: torch.Tensor, basis: Optional[torch.Tensor] = None):
    assert weight.shape == indexes.shape, f"shape mismatch {weight.shape} != {indexes.shape}"
    groups = [make_group(group, self.state) for group in self.param_groups]
    n = groups[0].param.shape[0]

    total_weight = PointState(groups[0].state).per_point('total_weight', n, weight.device)
    total_weight[indexes] += weight

    for group in groups:
      assert group.grad is None or group.num_points == n, f"param shape {group.num_points} != {n}"
    fused_update_groups(groups, weight, indexes, total_weight, self.kind, basis)


class FractionalAdam(FractionalOpt):
  def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-16, bias_correction=True):
    super().__init__(ADAM, params, lr, betas, eps, bias_correction)


class FractionalLaProp(FractionalOpt):
  def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-16, bias_correction=True):
    super().__init__(LAPROP, params, lr, betas, eps, bias_correction)


class SparseAdam(FractionalOpt):
  def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-16, bias_correction=True):
    super().__init__(ADAM, params, lr, betas, eps, bias_correction)

  def step(self, indexes: torch.Tensor, basis: Optional[torch.Tensor] = None):
    weight = torch.ones(indexes.shape[0], device=indexes.device, dtype=torch.float32)
    super().step(indexes, weight, basis)


class SparseLaProp(FractionalOpt):
  def __init__(self, params, lr=0.001, betas=(0.9, 0.999), eps=1e-16, bias_correction=True):
    super().__init__(LAPROP, params, lr, betas, eps, bias_correction)

  def step(self, indexes: torch.Tensor, basis: Optional[torch.Tensor] = None):
    weight = torch.ones(indexes.shape[0], device=indexes.device, dtype=torch.float32)
    super().step(indexes, weight, basis)
