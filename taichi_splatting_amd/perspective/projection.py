"""Perspective projection of 3D gaussians to packed 2D gaussians (EWA splatting).

Same interface as reference ``perspective/projection.py``: ``apply`` (:193-218) and
``project_to_image`` (:221-253).  Forward = project kernel + scan/compaction (one host read of the
visible count V); backward = hand-derived reverse mode (csrc/splat_math.h ``project_backward``)
instead of Taichi autodiff.
"""
from __future__ import annotations

import ctypes
from numbers import Integral
from typing import Tuple

import torch

from .. import _lib
from ..data_types import Gaussians3D, RasterConfig
from .params import CameraParams


def _identity_indexes(n: int, device: torch.device) -> torch.Tensor:
  """arange(n) int64 for the "every gaussian is visible" case: one shared, read-only tensor per (device, n), built
  outside inference mode so that an evaluation render under ``torch.inference_mode()`` cannot hand an inference
  tensor to later training frames (``frame.identity_indexes``)."""
  from ..frame import identity_indexes
  return identity_indexes(n, device)


def _project_forward(position, log_scaling, rotation, alpha_logit, T_camera_world, projection,
                     image_size, depth_range, blur_cov, clamp_margin, alpha_threshold, with_ndc=False):
  lib = _lib.load()
  _lib.require_gpu(position, log_scaling, rotation, alpha_logit, T_camera_world, projection)
  device, dtype = position.device, position.dtype
  code = _lib.dtype_code(dtype)
  n = position.shape[0]
  stream = _lib.current_stream(device)

  points_full = torch.empty((n, 7), dtype=dtype, device=device)
  depth_full = torch.empty((n,), dtype=dtype, device=device)
  flags = torch.empty((n,), dtype=torch.int32, device=device)
  w, h = int(image_size[0]), int(image_size[1])

  _lib.check(lib.ms_project_fwd(position.data_ptr(), log_scaling.data_ptr(), rotation.data_ptr(),
                                alpha_logit.data_ptr(), T_camera_world.data_ptr(), projection.data_ptr(),
                                w, h, float(depth_range[0]), float(depth_range[1]), float(blur_cov),
                                float(clamp_margin), float(alpha_threshold), n, points_full.data_ptr(),
                                depth_full.data_ptr(), flags.data_ptr(), code, stream), "project_to_image")
  if n == 0:
    empty_idx = torch.empty((0,), dtype=torch.int64, device=device)
    return points_full, depth_full.unsqueeze(1), empty_idx, depth_full.unsqueeze(1)

  scan = torch.empty((n + 1,), dtype=torch.int32, device=device)
  nbytes = ctypes.c_size_t(0)
  _lib.check(lib.ms_exclusive_scan_i32(None, n, None, None, None, ctypes.byref(nbytes), stream), "project_to_image")
  tmp = torch.empty((max(nbytes.value, 1),), dtype=torch.uint8, device=device)
  _lib.check(lib.ms_exclusive_scan_i32(flags.data_ptr(), n, scan.data_ptr(), None, tmp.data_ptr(),
                                       ctypes.byref(nbytes), stream), "project_to_image")
  v = int(scan[n].item())   # host sync: V sizes the outputs (the reference syncs in torch.nonzero)

  if v == n and not with_ndc:
    # every gaussian is visible: the uncompacted arrays ARE the result, no gather pass
    return points_full, depth_full.unsqueeze(1), _identity_indexes(n, device), None

  points = torch.empty((v, 7), dtype=dtype, device=device)
  depth = torch.empty((v, 1), dtype=dtype, device=device)
  ndc = torch.empty((v, 1), dtype=dtype, device=device) if with_ndc else None
  indexes = torch.empty((v,), dtype=torch.int64, device=device)
  if v > 0:
    _lib.check(lib.ms_project_gather(points_full.data_ptr(), depth_full.data_ptr(), flags.data_ptr(),
                                     scan.data_ptr(), n, float(depth_range[0]), float(depth_range[1]),
                                     points.data_ptr(), depth.data_ptr(), _lib.ptr(ndc), indexes.data_ptr(),
                                     code, stream), "project_to_image")
  return points, depth, indexes, ndc


class _ProjectFunction(torch.autograd.Function):
  """reference perspective/projection.py:123-188"""

  @staticmethod
  def forward(ctx, position, log_scaling, rotation, alpha_logit, T_camera_world, projection,
              image_size, depth_range, blur_cov, clamp_margin, alpha_threshold):
    tensors = [t.detach().contiguous() for t in
               (position, log_scaling, rotation, alpha_logit, T_camera_world, projection)]
    dtype = tensors[0].dtype
    assert all(t.dtype == dtype for t in tensors), "project_to_image: all inputs must share one dtype"

    points, depth, indexes, _ = _project_forward(*tensors, image_size, depth_range, blur_cov,
                                                 clamp_margin, alpha_threshold)
    ctx.set_materialize_grads(False)      # no zero tensors for outputs the loss does not touch (depth, indexes)
    ctx.image_size = image_size
    ctx.blur_cov, ctx.clamp_margin = blur_cov, clamp_margin
    ctx.indexes = indexes
    ctx.mark_non_differentiable(indexes)
    ctx.save_for_backward(*tensors)
    return points, depth, indexes

  @staticmethod
  def backward(ctx, dpoints, ddepth, dindexes):
    lib = _lib.load()
    position, log_scaling, rotation, alpha_logit, T_camera_world, projection = ctx.saved_tensors
    device, dtype = position.device, position.dtype
    indexes = ctx.indexes
    v = indexes.shape[0]

    # the kernel WRITES the rows listed in indexes; only culled rows need the zero fill
    alloc = torch.empty_like if v == position.shape[0] else torch.zeros_like
    grad_position = alloc(position)
    grad_log_scaling = alloc(log_scaling)
    grad_rotation = alloc(rotation)
    grad_alpha_logit = alloc(alpha_logit)
    need_camera = ctx.needs_input_grad[4] or ctx.needs_input_grad[5]
    grad_camera = torch.zeros((16,), dtype=dtype, device=device) if need_camera else None

    if dpoints is None and ddepth is None:
      return (None,) * 11
    if v > 0:
      dpoints = dpoints.contiguous() if dpoints is not None else torch.zeros((v, 7), dtype=dtype, device=device)
      ddepth = ddepth.contiguous() if ddepth is not None else None
      _lib.check(lib.ms_project_bwd(position.data_ptr(), log_scaling.data_ptr(), rotation.data_ptr(),
                                    alpha_logit.data_ptr(), T_camera_world.data_ptr(), projection.data_ptr(),
                                    int(ctx.image_size[0]), int(ctx.image_size[1]), float(ctx.blur_cov),
                                    float(ctx.clamp_margin), indexes.data_ptr(), v, dpoints.data_ptr(),
                                    _lib.ptr(ddepth), grad_position.data_ptr(), grad_log_scaling.data_ptr(),
                                    grad_rotation.data_ptr(), grad_alpha_logit.data_ptr(), _lib.ptr(grad_camera),
                                    _lib.dtype_code(dtype), _lib.current_stream(device)), "project_to_image backward")

    grad_T, grad_proj = None, None
    if need_camera:
      grad_T = torch.zeros((4, 4), dtype=dtype, device=device)
      grad_T[:3] = grad_camera[:12].view(3, 4)
      grad_proj = grad_camera[12:16].clone()
    return (grad_position, grad_log_scaling, grad_rotation, grad_alpha_logit, grad_T, grad_proj,
            None, None, None, None, None)


def apply(position: torch.Tensor, log_scaling: torch.Tensor, rotation: torch.Tensor,
          alpha_logit: torch.Tensor, T_camera_world: torch.Tensor, projection: torch.Tensor,
          image_size: Tuple[Integral, Integral], depth_range: Tuple[float, float],
          blur_cov: float = 0.0, clamp_margin: float = 0.15, alpha_threshold: float = 1. / 255.
          ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
  """Returns (points (V, 7), depth (V, 1), indexes (V,) int64) of the gaussians in view."""
  assert T_camera_world.shape[-2:] == (4, 4), f"T_camera_world must be (4, 4), got {T_camera_world.shape}"
  return _ProjectFunction.apply(position, log_scaling, rotation, alpha_logit,
                                T_camera_world.reshape(4, 4), projection.reshape(4),
                                tuple(int(x) for x in image_size), tuple(float(x) for x in depth_range),
                                float(blur_cov), float(clamp_margin), float(alpha_threshold))


def project_to_image(gaussians: Gaussians3D, camera_params: CameraParams, config: RasterConfig
                     ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
  """Project 3D gaussians to 2D gaussians in image space using perspective projection (EWA).

  Returns:
    points:  (V, 7) packed 2D gaussians [mean2, axis2, sigma2, alpha]
    depths:  (V, 1) camera-space depth
    indexes: (V,) int64 indexes of the gaussians in view
  """
  return apply(
    *gaussians.shape_tensors(),
    camera_params.T_camera_world,
    camera_params.projection,
    camera_params.image_size,
    camera_params.depth_range,
    config.blur_cov,
    config.clamp_margin,
    config.alpha_threshold)
