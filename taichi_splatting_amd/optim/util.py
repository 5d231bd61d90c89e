"""Per-point optimiser state helpers under the reference's module path (``optim/util.py``): thin functions over
``fractional.PointState``, which owns the key names and shapes ('v' first moment, 'm' second moment, ...)."""
import torch

from .fractional import PointState


def get_vector_state(state: dict, param: torch.Tensor):
  """(first moment (N, D), second moment (N,)) of a vector group, created on first use"""
  return PointState(state).moments(param, per_point_second_moment=True)


def get_scalar_state(state: dict, param: torch.Tensor):
  """(first moment (N, D), second moment (N, D)) of a scalar group, created on first use"""
  return PointState(state).moments(param, per_point_second_moment=False)


def get_total_weight(state: dict, n: int, device: torch.device):
  return PointState(state).per_point('total_weight', n, device)


def get_running_vis(state: dict, shape, device: torch.device):
  """zeros(shape) on first use — the whole shape, as the reference allocates it (optim/util.py:28-32)"""
  return PointState(state).per_point('running_vis', shape, device)


def flatten_param(param: torch.Tensor):
  """(param, param.grad) as (N, D) views"""
  return param.view(param.shape[0], -1), param.grad.view(param.shape[0], -1)
