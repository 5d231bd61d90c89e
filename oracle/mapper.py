"""Oracle: tile mapper (numpy float32, brute force, stable ordering).

Follows ``obb_grid_query`` / ``tile_ranges`` / ``separates_bbox`` (taichi_lib/grid_query.py:10-91),
``tile_overlaps_kernel`` / ``generate_sort_keys_kernel`` / ``find_ranges_kernel`` and the key
packing of mapper/tile_mapper.py:36-66,76-146,171-198.  The reference sorts (tile_id << 32 |
float_bits(depth)) with a stable radix sort over keys generated in point order, i.e. the final
order is (tile, depth bits, point index) — reproduced here with np.lexsort.

All arithmetic is float32 with one rounding per operation (no FMA), matching the HIP kernels
which are compiled with FP contraction off.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np

f32 = np.float32


def pad_to_tile(image_size, tile_size):
  return tuple(int(math.ceil(x / tile_size) * tile_size) for x in image_size)


def obb_queries(points: np.ndarray, image_size_pad, tile_size: int, alpha_threshold: float):
  """Per gaussian: inverse basis rows, rel_min_bound, min_tile, tile_span (grid_query.py:73-91)."""
  p = points.astype(f32)
  mean, axis1, sigma, alpha = p[:, 0:2], p[:, 2:4], p[:, 4:6], p[:, 6]
  with np.errstate(invalid='ignore', divide='ignore'):
    gs = np.sqrt(f32(2.0) * np.log(alpha / f32(alpha_threshold)))
    scale = sigma * gs[:, None]
    axis2 = np.stack([-axis1[:, 1], axis1[:, 0]], axis=1)
    v1 = axis1 * scale[:, 0:1]
    v2 = axis2 * scale[:, 1:2]
    extent = np.sqrt(v1 * v1 + v2 * v2)
    min_bound, max_bound = mean - extent, mean + extent
    inv0 = axis1 / scale[:, 0:1]
    inv1 = axis2 / scale[:, 1:2]

    ts = f32(tile_size)
    size = np.array(image_size_pad, dtype=np.int64)
    max_tile = (size - 1) // tile_size
    nan_lo = np.isnan(min_bound)
    nan_hi = np.isnan(max_bound)
    lo = np.where(nan_lo, 0x3fffffff, np.floor(np.where(nan_lo, 0, min_bound) / ts)).astype(np.int64)
    lo = np.maximum(lo, 0)
    hi = np.where(nan_hi, 0, np.ceil(np.where(nan_hi, 0, max_bound) / ts)).astype(np.int64)
    hi = np.minimum(np.maximum(hi, lo + 1), max_tile[None, :] + 1)
  span = hi - lo
  rel_min = (lo * tile_size).astype(f32) - mean
  return inv0, inv1, rel_min, lo, span


def _separates(inv0, inv1, lower, upper):
  """separates_bbox (grid_query.py:30-43) for arrays of boxes."""
  corners = [(lower[:, 0], lower[:, 1]), (upper[:, 0], lower[:, 1]),
             (upper[:, 0], upper[:, 1]), (lower[:, 0], upper[:, 1])]
  sep = np.zeros(lower.shape[0], dtype=bool)
  for inv in (inv0, inv1):
    with np.errstate(invalid='ignore'):
      proj = [inv[:, 0] * cx + inv[:, 1] * cy for cx, cy in corners]
      mn = np.fmin(np.fmin(proj[0], proj[1]), np.fmin(proj[2], proj[3]))
      mx = np.fmax(np.fmax(proj[0], proj[1]), np.fmax(proj[2], proj[3]))
      sep |= (mn > f32(1.0)) | (mx < f32(-1.0))
  return sep


def overlaps(points: np.ndarray, image_size, tile_size: int, alpha_threshold: float,
             tile_rows: Optional[Tuple[int, int]] = None):
  """All (point, tile_x, tile_y) pairs passing the OBB-vs-tile test, in point-major order."""
  w_pad, h_pad = pad_to_tile(image_size, tile_size)
  inv0, inv1, rel_min, lo, span = obb_queries(points, (w_pad, h_pad), tile_size, alpha_threshold)
  sx = np.maximum(span[:, 0], 0)
  sy = np.maximum(span[:, 1], 0)
  cand = sx * sy
  total = int(cand.sum())
  pid = np.repeat(np.arange(points.shape[0], dtype=np.int64), cand)
  start = np.cumsum(cand) - cand
  local = np.arange(total, dtype=np.int64) - np.repeat(start, cand)
  sy_rep = np.repeat(sy, cand)
  tu = local // np.maximum(sy_rep, 1)          # x outer, y inner (ti.ndrange(span.x, span.y))
  tv = local % np.maximum(sy_rep, 1)

  lower = rel_min[pid] + np.stack([tu, tv], axis=1).astype(f32) * f32(tile_size)
  upper = lower + f32(tile_size)
  keep = ~_separates(inv0[pid], inv1[pid], lower, upper)
  tx = lo[pid, 0] + tu
  ty = lo[pid, 1] + tv
  if tile_rows is not None:
    keep &= (ty >= tile_rows[0]) & (ty < tile_rows[1])
  return pid[keep], tx[keep], ty[keep]


def depth_key_bits(depth: np.ndarray, use_depth16: bool) -> np.ndarray:
  d = depth.astype(f32).reshape(-1)
  if use_depth16:
    return (np.clip(d, f32(0), f32(1)) * f32(65535.0)).astype(np.uint32).astype(np.uint64)
  return d.view(np.uint32).astype(np.uint64)


def map_to_tiles(points: np.ndarray, depth: np.ndarray, image_size, tile_size: int = 16,
                 alpha_threshold: float = 1. / 255., use_depth16: bool = False,
                 tile_rows: Optional[Tuple[int, int]] = None):
  """Returns (overlap_to_point (K,) int32, tile_ranges (TH, TW, 2) int32, counts (N,) int32)."""
  w_pad, h_pad = pad_to_tile(image_size, tile_size)
  tiles_wide, tiles_high = w_pad // tile_size, h_pad // tile_size
  pid, tx, ty = overlaps(points, image_size, tile_size, alpha_threshold, tile_rows)
  counts = np.bincount(pid, minlength=points.shape[0]).astype(np.int32)

  tile_id = tx + ty * tiles_wide
  bits = depth_key_bits(depth, use_depth16)[pid]
  order = np.lexsort((pid, bits, tile_id))        # (tile, depth bits, point index)
  o2p = pid[order].astype(np.int32)
  tile_sorted = tile_id[order]

  ranges = np.zeros((tiles_high * tiles_wide, 2), dtype=np.int32)
  if tile_sorted.shape[0] > 0:
    tiles, first = np.unique(tile_sorted, return_index=True)
    last = np.append(first[1:], tile_sorted.shape[0])
    ranges[tiles, 0] = first
    ranges[tiles, 1] = last
  return o2p, ranges.reshape(tiles_high, tiles_wide, 2), counts


def borderline_pairs(points: np.ndarray, image_size, tile_size: int, alpha_threshold: float, eps=1e-5):
  """(point, tile) candidates whose SAT statistic lies within eps of the decision threshold when
  evaluated in float64 — the only pairs on which a float32 implementation may legitimately differ."""
  w_pad, h_pad = pad_to_tile(image_size, tile_size)
  p = points.astype(np.float64)
  mean, axis1, sigma, alpha = p[:, 0:2], p[:, 2:4], p[:, 4:6], p[:, 6]
  with np.errstate(invalid='ignore', divide='ignore'):
    gs = np.sqrt(2.0 * np.log(alpha / alpha_threshold))
    scale = sigma * gs[:, None]
    axis2 = np.stack([-axis1[:, 1], axis1[:, 0]], axis=1)
    extent = np.sqrt((axis1 * scale[:, 0:1]) ** 2 + (axis2 * scale[:, 1:2]) ** 2)
    lo = np.maximum(np.floor((mean - extent) / tile_size) - 1, 0).astype(np.int64)
    hi = (np.ceil((mean + extent) / tile_size) + 1).astype(np.int64)
    hi = np.minimum(np.maximum(hi, lo + 1), np.array([w_pad, h_pad]) // tile_size)
  out = set()
  for i in range(points.shape[0]):
    if not np.isfinite(scale[i]).all():
      continue
    for tx in range(lo[i, 0], hi[i, 0]):
      for ty in range(lo[i, 1], hi[i, 1]):
        lower = np.array([tx, ty], dtype=np.float64) * tile_size - mean[i]
        upper = lower + tile_size
        xs = np.array([lower[0], upper[0], upper[0], lower[0]])
        ys = np.array([lower[1], lower[1], upper[1], upper[1]])
        near = False
        for ax, sc in ((axis1[i], scale[i, 0]), (axis2[i], scale[i, 1])):
          pr = (ax[0] * xs + ax[1] * ys) / sc
          if abs(pr.min() - 1.0) < eps or abs(pr.max() + 1.0) < eps:
            near = True
        # tile-span rounding: bounds within eps of a tile edge
        for b in ((mean[i] - extent[i]) / tile_size, (mean[i] + extent[i]) / tile_size):
          if np.any(np.abs(b - np.round(b)) < eps):
            near = True
        if near:
          out.add((i, tx, ty))
  return out
