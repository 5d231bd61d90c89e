"""Spherical-harmonics colour evaluation at gathered indexes.

Same interface as reference ``indexed_spherical_harmonics.py:166-177`` (``evaluate_sh_at``);
forward/backward kernels in csrc/sh.hip (hand-derived backward instead of Taichi autodiff).
"""
from __future__ import annotations

import math

import torch

from . import _lib


def check_sh_degree(sh_features):
  assert len(sh_features.shape) == 3, f"SH features must have 3 dimensions, got {sh_features.shape}"
  n_sh = sh_features.shape[2]
  n = int(math.sqrt(n_sh))
  assert n * n == n_sh, f"SH feature count must be square, got {n_sh} ({sh_features.shape})"
  return n - 1


class _SHFunction(torch.autograd.Function):
  """reference indexed_spherical_harmonics.py:138-160"""

  @staticmethod
  def forward(ctx, params, points, indexes, camera_pos, degree, unique_indexes):
    lib = _lib.load()
    _lib.require_gpu(params, points, indexes, camera_pos)
    params_c, points_c = params.detach().contiguous(), points.detach().contiguous()
    cam_c = camera_pos.detach().contiguous()
    indexes = indexes.contiguous()
    assert indexes.dtype == torch.int64, f"indexes must be int64, got {indexes.dtype}"
    assert points_c.dtype == params_c.dtype and cam_c.dtype == params_c.dtype, "evaluate_sh_at: dtype mismatch"
    v, f = indexes.shape[0], params_c.shape[1]
    out = torch.empty((v, f), dtype=params_c.dtype, device=params_c.device)
    _lib.check(lib.ms_sh_fwd(params_c.data_ptr(), points_c.data_ptr(), indexes.data_ptr(), cam_c.data_ptr(),
                             v, f, degree, out.data_ptr(), _lib.dtype_code(params_c.dtype),
                             _lib.current_stream(params_c.device)), "evaluate_sh_at")
    ctx.save_for_backward(params_c, points_c, cam_c, out)
    ctx.indexes, ctx.degree, ctx.unique = indexes, degree, bool(unique_indexes)
    return out

  @staticmethod
  def backward(ctx, doutput):
    lib = _lib.load()
    params, points, camera_pos, out = ctx.saved_tensors
    need_params, need_points, _, need_cam, _, _ = ctx.needs_input_grad
    v, f = ctx.indexes.shape[0], params.shape[1]
    # with unique indexes covering every row (all gaussians visible) the streaming kernel writes the
    # whole gradient: skip the 4*F*D*N byte zero fill (1.15 GB at 6 M gaussians, RGB degree 3)
    all_rows_written = (ctx.unique and v == params.shape[0] and f <= 4 and need_params
                        and not need_points and not need_cam)
    g_params = (torch.empty_like(params) if all_rows_written else torch.zeros_like(params)) if need_params else None
    g_points = torch.zeros_like(points) if need_points else None
    g_cam = torch.zeros_like(camera_pos) if need_cam else None
    if v > 0 and (need_params or need_points or need_cam):
      doutput = doutput.contiguous()
      _lib.check(lib.ms_sh_bwd(params.data_ptr(), points.data_ptr(), ctx.indexes.data_ptr(),
                               camera_pos.data_ptr(), v, f, ctx.degree, out.data_ptr(), doutput.data_ptr(),
                               _lib.ptr(g_params), _lib.ptr(g_points), _lib.ptr(g_cam), int(ctx.unique),
                               _lib.dtype_code(params.dtype), _lib.current_stream(params.device)),
                 "evaluate_sh_at backward")
    return g_params, g_points, None, g_cam, None, None


def evaluate_sh_at(sh_params: torch.Tensor,   # M, K, (degree + 1)^2  (usually K=3, for RGB)
                   positions: torch.Tensor,   # M, 3
                   indexes: torch.Tensor,     # N   (indexes to gaussians) 0 to M
                   camera_pos: torch.Tensor,  # 3
                   unique_indexes: bool = False   # promise: no repeated index (faster backward)
                   ) -> torch.Tensor:         # N, K
  degree = check_sh_degree(sh_params)
  assert 0 <= degree <= 3, f"SH degree must be between 0 and 3, got {degree}"
  return _SHFunction.apply(sh_params, positions, indexes, camera_pos, degree, unique_indexes)
