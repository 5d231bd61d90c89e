"""Oracle: tile rasterizer forward / backward (torch, CPU, any float dtype).

Forward follows ``_forward_kernel`` (rasterizer/forward.py:39-135), backward follows
``_backward_kernel`` (rasterizer/backward.py:97-224) with ``gaussian_pdf[_antialias][_with_grad]``
(taichi_lib/generic.py:311-404).  The in-group loop bound defect is NOT
reproduced: every splat of a tile's range is visited exactly once, front to back.

Per tile the pixel x splat interaction is evaluated as a matrix; the sequential recurrence
``W += alpha * (1 - W)`` is the exclusive cumulative product of (1 - alpha), so the result is the
literal loop's up to rounding.  ``rasterize_autograd`` exposes the same forward to torch autograd
(straight-through alpha clamp) as an independent check of the literal backward.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch


def perp(v):
  return torch.stack([-v[..., 1], v[..., 0]], dim=-1)


def _tile_pixels(tile_id, tiles_wide, tile_size, width, height, dtype):
  tu, tv = tile_id % tiles_wide, tile_id // tiles_wide
  ys, xs = torch.meshgrid(torch.arange(tile_size), torch.arange(tile_size), indexing='ij')
  px = (xs + tu * tile_size).reshape(-1)
  py = (ys + tv * tile_size).reshape(-1)
  in_bounds = (px < width) & (py < height)
  pix = torch.stack([px, py], dim=-1).to(dtype) + 0.5      # pixel centres (forward.py:46)
  return px, py, in_bounds, pix


def pdf(pix, g, antialias: bool):
  """pix (P, 2), g (S, 7) -> (P, S)  (generic.py:311-317 / :341-357)"""
  mean, axis, sigma = g[:, 0:2], g[:, 2:4], g[:, 4:6]
  d = pix[:, None, :] - mean[None]                          # (P, S, 2)
  a_perp = perp(axis)
  if not antialias:
    tx = (d * axis[None]).sum(-1) / sigma[None, :, 0]
    ty = (d * a_perp[None]).sum(-1) / sigma[None, :, 1]
    return torch.exp(-0.5 * (tx * tx + ty * ty))
  tx = (d * axis[None]).sum(-1)
  ty = (d * a_perp[None]).sum(-1)
  sx, sy = sigma[None, :, 0], sigma[None, :, 1]

  def S(x, s):
    z = x / s
    # == 1 / (1 + exp(-1.6 z - 0.07 z^3)) (generic.py:337-340), written overflow-free for autograd
    return torch.sigmoid(1.6 * z + 0.07 * z ** 3)
  return 2 * math.pi * sx * (S(tx + 0.5, sx) - S(tx - 0.5, sx)) * sy * (S(ty + 0.5, sy) - S(ty - 0.5, sy))


def pdf_with_grad(pix, g, antialias: bool):
  """Returns p (P,S), dp_dmean (P,S,2), dp_daxis (P,S,2), dp_dsigma (P,S,2)
  (generic.py:321-336 / :371-404)."""
  mean, axis, sigma = g[:, 0:2], g[:, 2:4], g[:, 4:6]
  d = pix[:, None, :] - mean[None]
  ax = axis[None].expand_as(d)
  a_perp = perp(axis)[None].expand_as(d)
  sx, sy = sigma[None, :, 0], sigma[None, :, 1]
  if not antialias:
    tx = (d * ax).sum(-1) / sx
    ty = (d * a_perp).sum(-1) / sy
    tx2, ty2 = tx * tx, ty * ty
    p = torch.exp(-0.5 * (tx2 + ty2))
    dp_dsigma = torch.stack([tx2 * p / sx, ty2 * p / sy], dim=-1)
    tx_s, ty_s = tx / sx, ty / sy
    dp_daxis = p[..., None] * (tx_s[..., None] * -d + ty_s[..., None] * perp(d))
    dp_dmean = p[..., None] * (tx_s[..., None] * ax + ty_s[..., None] * a_perp)
    return p, dp_dmean, dp_daxis, dp_dsigma

  tx = (d * ax).sum(-1)
  ty = (d * a_perp).sum(-1)

  def S_grad(x, s):
    z = x / s
    sg = torch.sigmoid(1.6 * z + 0.07 * z ** 3)
    ds_dx = (1.6 + 0.21 * z * z) * sg * (1 - sg)
    dS_dx = ds_dx / s
    return sg, dS_dx, dS_dx * -z
  Sx1, dSx1, dSx1s = S_grad(tx + 0.5, sx)
  Sx2, dSx2, dSx2s = S_grad(tx - 0.5, sx)
  Sy1, dSy1, dSy1s = S_grad(ty + 0.5, sy)
  Sy2, dSy2, dSy2s = S_grad(ty - 0.5, sy)
  ix, iy = sx * (Sx1 - Sx2), sy * (Sy1 - Sy2)
  tau = 2 * math.pi
  p = tau * ix * iy
  dSx = iy * sx * (dSx1 - dSx2)
  dSy = ix * sy * (dSy1 - dSy2)
  dp_dmean = tau * (dSx[..., None] * -ax + dSy[..., None] * -a_perp)
  dp_dsigma = torch.stack([tau * iy * (Sx1 - Sx2 + (dSx1s - dSx2s) * sx),
                           tau * ix * (Sy1 - Sy2 + (dSy1s - dSy2s) * sy)], dim=-1)
  dp_daxis = tau * (dSx[..., None] * d + dSy[..., None] * -perp(d))
  return p, dp_dmean, dp_daxis, dp_dsigma


class Cfg:
  """Subset of RasterConfig the oracle needs (duck-typed: any object with these fields works)."""
  def __init__(self, tile_size=16, antialias=False, clamp_max_alpha=0.99, alpha_threshold=1. / 255.,
               saturate_threshold=0.9999, use_alpha_blending=True, compute_visibility=False,
               compute_point_heuristic=False, **_):
    self.tile_size = tile_size
    self.antialias = antialias
    self.clamp_max_alpha = clamp_max_alpha
    self.alpha_threshold = alpha_threshold
    self.saturate_threshold = saturate_threshold
    self.use_alpha_blending = use_alpha_blending
    self.compute_visibility = compute_visibility
    self.compute_point_heuristic = compute_point_heuristic


def _tiles(image_size, tile_size):
  w, h = image_size
  return (w + tile_size - 1) // tile_size, (h + tile_size - 1) // tile_size


def reference_tail_order(count: int, group: int):
  """Order in which the REFERENCE's forward / backward loops visit a tile's ``count`` splats, including the
  revisits caused by its in-group loop bound ``min(group, count - group_id)`` (forward.py:86-89,
  backward.py:138-141; it should be ``count - group_id * group``).  In the last, partially filled group the
  loop runs past the valid entries and re-blends whatever the previous group left in shared memory at those
  positions.  Returns a list of indices into the tile's list.  ``group`` = threads per
  block: tile_size^2 in the forward kernel, tile_size^2 / (stride_x * stride_y) in the backward kernel."""
  order = []
  groups = (count + group - 1) // group
  for gi in range(groups):
    valid = min(group, count - gi * group)
    visited = min(group, count - gi)
    for k in range(visited):
      if k < valid:
        order.append(gi * group + k)
      elif gi > 0:
        order.append((gi - 1) * group + k)       # stale shared-memory entry of the previous group
      # gi == 0: visited == valid, nothing stale
  return order


def forward(points, feats, ranges, o2p, image_size, cfg, tile_rows: Optional[Tuple[int, int]] = None,
            return_borderline: bool = False, emulate_reference_loop_bound: bool = False,
            reference_group: Optional[int] = None):
  """Returns image (H,W,F), alpha (H,W), visibility (V,) [, borderline (H,W) bool].

  ``borderline`` marks pixels where some splat's alpha lies within 1e-6 (relative) of the
  ``alpha_threshold`` gate: a float32 implementation may legitimately flip that gate.

  ``emulate_reference_loop_bound`` reproduces the reference's loop-bound defect (see
  ``reference_tail_order``; ``reference_group`` defaults to tile_size^2, the forward kernel's block size) so
  that the deviation of this library — which visits every splat exactly once — can be quantified.
  """
  w, h = image_size
  ts = cfg.tile_size
  tiles_wide, tiles_high = _tiles(image_size, ts)
  dtype = points.dtype
  F = feats.shape[1]
  image = torch.zeros((h, w, F), dtype=dtype)
  alpha_img = torch.zeros((h, w), dtype=dtype)
  visibility = torch.zeros((points.shape[0],), dtype=dtype)
  borderline = torch.zeros((h, w), dtype=torch.bool)
  ranges = ranges.reshape(-1, 2)
  r0, r1 = (0, tiles_high) if tile_rows is None else tile_rows

  for tile_id in range(r0 * tiles_wide, r1 * tiles_wide):
    start, end = int(ranges[tile_id, 0]), int(ranges[tile_id, 1])
    px, py, inb, pix = _tile_pixels(tile_id, tiles_wide, ts, w, h, dtype)
    P = pix.shape[0]
    C = torch.zeros((P, F), dtype=dtype)
    W = torch.where(inb, torch.zeros(P, dtype=dtype), torch.ones(P, dtype=dtype))
    if end > start:
      ids = o2p[start:end].long()
      if emulate_reference_loop_bound:
        ids = ids[torch.tensor(reference_tail_order(end - start, reference_group or ts * ts), dtype=torch.long)]
      g, f = points[ids], feats[ids]
      a_raw = g[None, :, 6] * pdf(pix, g, cfg.antialias)                  # (P, S)
      a = torch.clamp_max(a_raw, cfg.clamp_max_alpha)
      gate = a > cfg.alpha_threshold
      if return_borderline:
        near = (a_raw - cfg.alpha_threshold).abs() < 1e-6 * cfg.alpha_threshold + 1e-9
        bl = near.any(dim=1)
      a = torch.where(gate, a, torch.zeros_like(a))
      T_incl = torch.cumprod(1 - a, dim=1)
      T_excl = torch.cat([torch.ones((P, 1), dtype=dtype), T_incl[:, :-1]], dim=1)
      T_excl = T_excl * (1 - W)[:, None]                                   # out-of-bounds start at W = 1
      weight = a * T_excl
      W_after = 1 - T_incl * (1 - W)[:, None]
      if cfg.use_alpha_blending:
        C = weight @ f
        W = W_after[:, -1]
        visibility.index_add_(0, ids, weight.sum(0))
      else:
        # forward.py:107-112: feature of the first splat at which W >= 1 - saturate_threshold
        sat = gate & (W_after >= 1.0 - cfg.saturate_threshold)
        any_sat = sat.any(dim=1)
        first = torch.argmax(sat.to(torch.int8), dim=1)
        C = torch.where(any_sat[:, None], f[first], torch.zeros((P, F), dtype=dtype))
        W = W_after[:, -1]
        # forward.py:114-126 keeps adding the blend weights to `visibility` in this mode too.  The reference
        # stops only when all 32 lanes of a warp are saturated (forward.py:92-94), which depends on its
        # thread -> pixel map; the oracle accumulates the untruncated weights (an upper bound of the reference's).
        visibility.index_add_(0, ids, weight.sum(0))
      if return_borderline:
        borderline[py[inb], px[inb]] = bl[inb]
    image[py[inb], px[inb]] = C[inb]
    if cfg.use_alpha_blending:
      alpha_img[py[inb], px[inb]] = W[inb]
    else:
      alpha_img[py[inb], px[inb]] = (W[inb] > 0).to(dtype)
  if return_borderline:
    return image, alpha_img, visibility, borderline
  return image, alpha_img, visibility


def gate_margin(points, ranges, o2p, image_size, cfg, tile_rows: Optional[Tuple[int, int]] = None):
  """Per splat: min over the pixels of the tiles it is listed in of |alpha_pt * g / alpha_threshold - 1|, the
  relative distance of the nearest (pixel, splat) pair to the blend gate ``alpha > alpha_threshold``
  (forward.py:99-101).  A float32 implementation evaluates alpha_pt * g to ~1e-6 relative, so pairs closer than
  that may legitimately fall on the other side of the gate; tests drop such splats from a scene ("gate-stable
  scene") and can then compare images AND gradients without a borderline allowance."""
  w, h = image_size
  ts = cfg.tile_size
  tiles_wide, tiles_high = _tiles(image_size, ts)
  dtype = points.dtype
  margin = torch.full((points.shape[0],), float('inf'), dtype=dtype)
  ranges = ranges.reshape(-1, 2)
  r0, r1 = (0, tiles_high) if tile_rows is None else tile_rows
  for tile_id in range(r0 * tiles_wide, r1 * tiles_wide):
    start, end = int(ranges[tile_id, 0]), int(ranges[tile_id, 1])
    if end <= start:
      continue
    px, py, inb, pix = _tile_pixels(tile_id, tiles_wide, ts, w, h, dtype)
    if not bool(inb.any()):
      continue
    ids = o2p[start:end].long()
    g = points[ids]
    a_raw = g[None, :, 6] * pdf(pix[inb], g, cfg.antialias)
    m = (a_raw / cfg.alpha_threshold - 1).abs().min(dim=0).values
    margin.scatter_reduce_(0, ids, m, reduce='amin')
  return margin


def near_gate(points, ranges, o2p, image_size, cfg, eps: float, return_counts: bool = False):
  """Where can a float32 implementation legitimately differ?  Returns (pixel_flag (H, W) bool: some (pixel, splat)
  pair of the pixel's tile list has alpha_pt * g within ``eps`` (relative) of the blend gate; splat_flag (V,) bool:
  the splat contributes — alpha above half the threshold — to such a pixel).  One flipped gate moves its pixel and,
  through T and the remaining colour, the gradient of EVERY splat that contributes to that pixel: tests on unfiltered
  scenes require that each deviation beyond the tolerance is explained by these flags (and count the unexplained).
  ``return_counts``: additionally (H, W) int — how many pairs of the pixel sit at the gate — which bounds how far the
  pixel can move: one flipped pair of alpha ~ threshold at transmittance T changes the blended colour by
  alpha T (-f + colour behind it / (T (1 - alpha))), at most 2 alpha_threshold max|f|."""
  w, h = image_size
  ts = cfg.tile_size
  tiles_wide, tiles_high = _tiles(image_size, ts)
  pixel_flag = torch.zeros((h, w), dtype=torch.bool)
  pixel_count = torch.zeros((h, w), dtype=torch.int32)
  splat_flag = torch.zeros((points.shape[0],), dtype=torch.bool)
  ranges = ranges.reshape(-1, 2)
  for tile_id in range(tiles_wide * tiles_high):
    start, end = int(ranges[tile_id, 0]), int(ranges[tile_id, 1])
    if end <= start:
      continue
    px, py, inb, pix = _tile_pixels(tile_id, tiles_wide, ts, w, h, points.dtype)
    if not bool(inb.any()):
      continue
    ids = o2p[start:end].long()
    g = points[ids]
    a_raw = g[None, :, 6] * pdf(pix[inb], g, cfg.antialias)                  # (pixels, splats)
    near = (a_raw / cfg.alpha_threshold - 1).abs() < eps
    flagged = near.any(dim=1)
    pixel_flag[py[inb].long(), px[inb].long()] = flagged
    pixel_count[py[inb].long(), px[inb].long()] = near.sum(dim=1).to(torch.int32)
    touched = ((a_raw > 0.5 * cfg.alpha_threshold) & flagged[:, None]).any(dim=0)
    splat_flag[ids[touched]] = True
  if return_counts:
    return pixel_flag, splat_flag, pixel_count
  return pixel_flag, splat_flag


def active_visibility(points, ranges, o2p, image_size, cfg):
  """Per splat: the sum of its blend weights over the (pixel, splat) pairs the BACKWARD visits — ``forward``'s visibility
  (forward.py:127-128) without the pairs behind a pixel's saturation point, which backward.py:154,160 drop while the
  forward keeps adding them (it has no early exit).  What ``ms_frame_grads.point_visibility`` holds
  (frame.VISIBILITY_FROM_BACKWARD); it is below ``forward``'s visibility by at most 1 - saturate_threshold per pixel."""
  w, h = image_size
  ts = cfg.tile_size
  tiles_wide, tiles_high = _tiles(image_size, ts)
  dtype = points.dtype
  visibility = torch.zeros((points.shape[0],), dtype=dtype)
  ranges = ranges.reshape(-1, 2)
  for tile_id in range(tiles_wide * tiles_high):
    start, end = int(ranges[tile_id, 0]), int(ranges[tile_id, 1])
    if end <= start:
      continue
    px, py, inb, pix = _tile_pixels(tile_id, tiles_wide, ts, w, h, dtype)
    pix = pix[inb]                                  # out-of-bounds pixels start saturated (W = 1)
    P = pix.shape[0]
    if P == 0:
      continue
    ids = o2p[start:end].long()
    g = points[ids]
    a_raw = g[None, :, 6] * pdf(pix, g, cfg.antialias)
    gate = a_raw > cfg.alpha_threshold
    a = torch.where(gate, torch.clamp_max(a_raw, cfg.clamp_max_alpha), torch.zeros_like(a_raw))
    T_incl = torch.cumprod(1 - a, dim=1)
    T_excl = torch.cat([torch.ones((P, 1), dtype=dtype), T_incl[:, :-1]], dim=1)
    active = gate & ((1 - T_excl) < cfg.saturate_threshold)       # backward.py:154,160
    weight = torch.where(active, a, torch.zeros_like(a)) * T_excl
    visibility.index_add_(0, ids, weight.sum(0))
  return visibility


def saturation_margin(points, ranges, o2p, image_size, cfg):
  """Where the BACKWARD's saturation test can legitimately flip in float32.  backward.py:154,160 drop a (pixel, splat)
  pair once the accumulated weight in front of it reaches ``saturate_threshold``, i.e. once the transmittance T in front
  falls to ``1 - saturate_threshold``; a float32 T (a product of up to thousands of factors, re-associated when a tile's
  list is cut into segments) is within ~1e-5 relative of the float64 one.  Returns, per splat,

    margin (V,): min over its gated (pixel, splat) pairs of |T_before / (1 - saturate_threshold) - 1|   (inf: no pair)
    side (V,) int8: at that nearest pair, +1 = still blending (T above the limit), -1 = dropped, 0 = no pair

  A pair that flips toggles a contribution of weight <= alpha (1 - saturate_threshold): tests use the margin to say
  WHICH rows may carry such a toggle instead of allowing a count of deviating rows."""
  w, h = image_size
  ts = cfg.tile_size
  tiles_wide, tiles_high = _tiles(image_size, ts)
  dtype = points.dtype
  V = points.shape[0]
  margin = torch.full((V,), float('inf'), dtype=dtype)
  side = torch.zeros((V,), dtype=torch.int8)
  limit = 1.0 - cfg.saturate_threshold
  ranges = ranges.reshape(-1, 2)
  for tile_id in range(tiles_wide * tiles_high):
    start, end = int(ranges[tile_id, 0]), int(ranges[tile_id, 1])
    if end <= start:
      continue
    px, py, inb, pix = _tile_pixels(tile_id, tiles_wide, ts, w, h, dtype)
    if not bool(inb.any()):
      continue
    ids = o2p[start:end].long()
    g = points[ids]
    a_raw = g[None, :, 6] * pdf(pix[inb], g, cfg.antialias)
    gate = a_raw > cfg.alpha_threshold
    a = torch.where(gate, torch.clamp_max(a_raw, cfg.clamp_max_alpha), torch.zeros_like(a_raw))
    T_incl = torch.cumprod(1 - a, dim=1)
    T_excl = torch.cat([torch.ones((a.shape[0], 1), dtype=dtype), T_incl[:, :-1]], dim=1)
    rel = T_excl / limit - 1
    dist = torch.where(gate, rel.abs(), torch.full_like(rel, float('inf')))
    m, arg = dist.min(dim=0)
    s = torch.sign(rel.gather(0, arg[None])[0]).to(torch.int8)
    s = torch.where(torch.isinf(m), torch.zeros_like(s), s)
    better = m < margin[ids]
    margin[ids[better]] = m[better]
    side[ids[better]] = s[better]
  return margin, side


def backward(points, feats, ranges, o2p, image, grad_image, image_size, cfg,
             tile_rows: Optional[Tuple[int, int]] = None):
  """Literal restatement of backward.py:97-224.  Returns grad_points (V,7), grad_feats (V,F),
  point_heuristic (V,2)."""
  w, h = image_size
  ts = cfg.tile_size
  tiles_wide, tiles_high = _tiles(image_size, ts)
  dtype = points.dtype
  V, F = feats.shape
  grad_points = torch.zeros((V, 7), dtype=dtype)
  grad_feats = torch.zeros((V, F), dtype=dtype)
  heuristic = torch.zeros((V, 2), dtype=dtype)
  ranges = ranges.reshape(-1, 2)
  r0, r1 = (0, tiles_high) if tile_rows is None else tile_rows

  for tile_id in range(r0 * tiles_wide, r1 * tiles_wide):
    start, end = int(ranges[tile_id, 0]), int(ranges[tile_id, 1])
    if end <= start:
      continue
    px, py, inb, pix = _tile_pixels(tile_id, tiles_wide, ts, w, h, dtype)
    pix, px, py = pix[inb], px[inb], py[inb]       # out-of-bounds pixels start saturated (W = 1)
    P = pix.shape[0]
    if P == 0:
      continue
    ids = o2p[start:end].long()
    g, f = points[ids], feats[ids]
    C_final = image[py, px]                         # (P, F)  "remaining_features" initial value
    G = grad_image[py, px]                          # (P, F)

    p, dp_dmean, dp_daxis, dp_dsigma = pdf_with_grad(pix, g, cfg.antialias)
    pa = g[None, :, 6]
    a_raw = pa * p
    gate = a_raw > cfg.alpha_threshold
    a = torch.where(gate, torch.clamp_max(a_raw, cfg.clamp_max_alpha), torch.zeros_like(a_raw))
    T_incl = torch.cumprod(1 - a, dim=1)
    T_excl = torch.cat([torch.ones((P, 1), dtype=dtype), T_incl[:, :-1]], dim=1)
    W_before = 1 - T_excl
    active = gate & (W_before < cfg.saturate_threshold)      # backward.py:154,160
    a = torch.where(active, a, torch.zeros_like(a))
    Ti = T_excl
    weight = a * Ti                                            # (P, S)
    prefix = torch.cumsum(weight[:, :, None] * f[None], dim=1)  # (P, S, F) inclusive
    R = C_final[:, None, :] - prefix                           # remaining features after splat k
    diff = f[None] * Ti[:, :, None] - R / (1 - a)[:, :, None]
    alpha_grad = (diff * G[:, None, :]).sum(-1)                # (P, S)
    alpha_grad = torch.where(active, alpha_grad, torch.zeros_like(alpha_grad))
    aag = pa * alpha_grad

    gp = torch.cat([(aag[..., None] * dp_dmean), (aag[..., None] * dp_daxis),
                    (aag[..., None] * dp_dsigma), (p * alpha_grad)[..., None]], dim=-1)  # (P,S,7)
    grad_points.index_add_(0, ids, gp.sum(0))
    grad_feats.index_add_(0, ids, (weight[:, :, None] * G[:, None, :]).sum(0))
    pos_grad = aag[..., None] * dp_dmean
    heur = torch.stack([(aag * aag).sum(0), pos_grad.abs().sum(-1).sum(0)], dim=-1)
    heuristic.index_add_(0, ids, heur)
  return grad_points, grad_feats, heuristic


def rasterize_autograd(points, feats, ranges, o2p, image_size, cfg):
  """Differentiable forward (torch autograd) with the straight-through alpha clamp; no backward
  saturation skip.  Used to cross-check ``backward`` where saturation does not occur."""
  w, h = image_size
  ts = cfg.tile_size
  tiles_wide, tiles_high = _tiles(image_size, ts)
  dtype = points.dtype
  F = feats.shape[1]
  ranges = ranges.reshape(-1, 2)
  rows = []
  image = torch.zeros((h, w, F), dtype=dtype)
  pieces = []
  for tile_id in range(tiles_wide * tiles_high):
    start, end = int(ranges[tile_id, 0]), int(ranges[tile_id, 1])
    if end <= start:
      continue
    px, py, inb, pix = _tile_pixels(tile_id, tiles_wide, ts, w, h, dtype)
    pix, px, py = pix[inb], px[inb], py[inb]
    if pix.shape[0] == 0:
      continue
    ids = o2p[start:end].long()
    g, f = points[ids], feats[ids]
    a_raw = g[None, :, 6] * pdf(pix, g, cfg.antialias)
    gate = a_raw > cfg.alpha_threshold
    a_st = a_raw + (torch.clamp_max(a_raw, cfg.clamp_max_alpha) - a_raw).detach()   # straight through
    a = torch.where(gate, a_st, torch.zeros_like(a_st))
    T_incl = torch.cumprod(1 - a.detach(), dim=1)
    # build the transmittance differentiably: exclusive cumprod of (1 - a)
    one_minus = 1 - a
    T_excl = torch.cat([torch.ones((pix.shape[0], 1), dtype=dtype),
                        torch.cumprod(one_minus, dim=1)[:, :-1]], dim=1)
    C = (a * T_excl) @ f
    pieces.append((py, px, C))
  for py, px, C in pieces:
    image = image.index_put((py, px), C)
  return image
