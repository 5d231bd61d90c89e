"""Oracle: perspective projection of 3D gaussians (torch, differentiable through autograd).

Follows ``project_kernel`` (perspective/projection.py:33-81) built from
``project_with_jacobian`` (taichi_lib/generic.py:96-121), ``gaussian_covariance_in_image``
(:126-143), ``scaled_quat_to_mat`` (:419-427), ``eig`` (:217-230), ``ellipse_bounds`` (:235-237).
"""
from __future__ import annotations

import torch


def quat_to_mat(q: torch.Tensor) -> torch.Tensor:
  """xyzw quaternion -> rotation matrix (taichi_lib/generic.py:408-416)."""
  x, y, z, w = q.unbind(-1)
  x2, y2, z2 = x * x, y * y, z * z
  rows = [1 - 2 * y2 - 2 * z2, 2 * x * y - 2 * w * z, 2 * x * z + 2 * w * y,
          2 * x * y + 2 * w * z, 1 - 2 * x2 - 2 * z2, 2 * y * z - 2 * w * x,
          2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x2 - 2 * y2]
  return torch.stack(rows, dim=-1).reshape(q.shape[:-1] + (3, 3))


def covariance_all(position, log_scaling, rotation, alpha_logit, T_camera_world, projection,
                   image_size, blur_cov=0.0, clamp_margin=0.15):
  """The projection up to the blurred 2D covariance: (uv (N, 2), a, b, c (N,), alpha (N,), z (N,)) with
  covariance [[a, b], [b, c]] — everything ``project_all`` computes before the eigen-decomposition.  Tests use it to
  push a covariance gradient through the float64 chain (the rasterizer's gradient does not depend on how the
  covariance is factored into axis and sigma)."""
  dtype, device = position.dtype, position.device
  W, H = image_size
  size = torch.tensor([W, H], dtype=dtype, device=device)
  f, c = projection[0:2], projection[2:4]
  T = T_camera_world[:3]                       # 3x4

  qh = rotation / torch.sqrt((rotation * rotation).sum(-1, keepdim=True))
  s = torch.exp(log_scaling)

  pc = position @ T[:, :3].T + T[:, 3]         # (N, 3)
  z = pc[:, 2]
  uv = (f * pc[:, :2]) / z.unsqueeze(1) + c
  t = torch.minimum(torch.maximum(uv, -size * clamp_margin), (size - 1) * (1 + clamp_margin))

  zero = torch.zeros_like(z)
  J = torch.stack([f[0] / z, zero, -(t[:, 0] - c[0]) / z,
                   zero, f[1] / z, -(t[:, 1] - c[1]) / z], dim=-1).reshape(-1, 2, 3)

  RS = quat_to_mat(qh) * s.unsqueeze(1)        # scale columns
  M = J @ T[:, :3].unsqueeze(0) @ RS           # (N, 2, 3)
  cov = M @ M.transpose(1, 2)
  a = cov[:, 0, 0] + blur_cov
  b = cov[:, 0, 1]
  cc = cov[:, 1, 1] + blur_cov
  alpha = 1.0 / (1.0 + torch.exp(-alpha_logit.reshape(-1)))
  return uv, a, b, cc, alpha, z


def project_all(position, log_scaling, rotation, alpha_logit, T_camera_world, projection,
                image_size, depth_range, blur_cov=0.0, clamp_margin=0.15, alpha_threshold=1. / 255.):
  """Projects every gaussian; returns (points (N,7), depth (N,), in_view (N,) bool)."""
  dtype, device = position.dtype, position.device
  W, H = image_size
  size = torch.tensor([W, H], dtype=dtype, device=device)
  uv, a, b, cc, alpha, z = covariance_all(position, log_scaling, rotation, alpha_logit, T_camera_world, projection,
                                          image_size, blur_cov, clamp_margin)

  # eig (generic.py:217-230)
  tr = a + cc
  det = a * cc - b * b
  gap = tr * tr - 4 * det
  sg = torch.sqrt(torch.clamp_min(gap, 0))
  l1, l2 = (tr + sg) * 0.5, (tr - sg) * 0.5
  sigma = torch.sqrt(torch.stack([l1, l2], dim=-1))
  v = torch.stack([a - l2, b], dim=-1)
  v1 = v / torch.sqrt((v * v).sum(-1, keepdim=True))
  v2 = torch.stack([-v1[:, 1], v1[:, 0]], dim=-1)

  gs = torch.sqrt(2 * torch.log(alpha / alpha_threshold))   # NaN when alpha < threshold => culled
  sc = sigma * gs.unsqueeze(1)
  e1, e2 = v1 * sc[:, 0:1], v2 * sc[:, 1:2]
  extent = torch.sqrt(e1 * e1 + e2 * e2)
  lower, upper = uv - extent, uv + extent

  in_view = ((z > depth_range[0]) & (z < depth_range[1])
             & (upper > 0).all(1) & (lower < size.unsqueeze(0)).all(1))

  points = torch.cat([uv, v1, sigma, alpha.unsqueeze(1)], dim=-1)
  return points, z, in_view


def apply(position, log_scaling, rotation, alpha_logit, T_camera_world, projection,
          image_size, depth_range, blur_cov=0.0, clamp_margin=0.15, alpha_threshold=1. / 255.):
  """Same contract as perspective/projection.py:193-218: (points (V,7), depth (V,1), indexes (V,))."""
  points, z, in_view = project_all(position, log_scaling, rotation, alpha_logit, T_camera_world,
                                   projection, image_size, depth_range, blur_cov, clamp_margin,
                                   alpha_threshold)
  idx = in_view.nonzero(as_tuple=True)[0]
  return points[idx], z[idx].unsqueeze(1), idx


def ndc_depth(depth, near, far):
  """torch_lib/projection.py:120-123"""
  return 1 - (1. / depth - 1. / far) / (1. / near - 1. / far)
