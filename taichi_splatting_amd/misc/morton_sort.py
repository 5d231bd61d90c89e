"""Z-order (Morton) sorting of 3D points: same surface as reference ``misc/morton_sort.py:102-143``
(``argsort``, ``sort``, ``argsort_dedup``, ``sort_dedup``), built on ``ms_morton_codes64`` +
the hand-written 64-bit radix sort (``cuda_lib.radix_argsort``).

Grid (reference :102-106): anchored at the per-axis minimum of the points, ``2**20`` cells of size
``resolution`` per axis; cell = clamp((p - lower) / inc, 0, size - 1) truncated, code = bit interleave
x | y << 1 | z << 2 of the 21-bit cell coordinates (:13-33, :56-70).

Deviation: the reference's ``argsort_dedup`` indexes the UNSORTED points with positions of the SORTED
array (and passes a float tensor as sort values); here it returns ORIGINAL indices — one point per
occupied cell (the last one in stable code order), in code order — so ``points[argsort_dedup(...)]`` is
the de-duplicated, Morton-ordered set the function name promises.
"""
from __future__ import annotations

import ctypes
from typing import Tuple

import numpy as np
import torch

from .. import _lib
from ..cuda_lib import radix_sort_pairs

GRID_SIZE = 2 ** 20


def grid_at_resolution(points: torch.Tensor, resolution: float, size: int = GRID_SIZE) -> Tuple[np.ndarray, np.ndarray, int]:
  """(lower (3,), inc (3,), size) in float32, as reference :102-106 / Grid.get_inc :42-45."""
  lower = points.min(dim=0).values.to(torch.float32).cpu().numpy()
  upper = (lower + np.float32(size) * np.float32(resolution)).astype(np.float32)
  inc = ((upper - lower) / np.float32(size)).astype(np.float32)
  return lower, inc, size


def morton_codes(points: torch.Tensor, resolution: float, size: int = GRID_SIZE) -> torch.Tensor:
  """uint64 codes (stored as int64: codes use 63 bits, so the signed order equals the unsigned one)."""
  _lib.require_gpu(points)
  assert points.ndim == 2 and points.shape[1] == 3, f"points must be (N, 3), got {points.shape}"
  assert resolution > 0, "resolution must be positive"
  pts = points.detach().to(torch.float32).contiguous()
  codes = torch.empty((pts.shape[0],), dtype=torch.int64, device=pts.device)
  if pts.shape[0] == 0:
    return codes
  lower, inc, size = grid_at_resolution(pts, resolution, size)
  lib = _lib.load()
  _lib.check(lib.ms_morton_codes64(_lib.ptr(pts), pts.shape[0], lower.ctypes.data_as(ctypes.c_void_p),
                                   inc.ctypes.data_as(ctypes.c_void_p), size, _lib.ptr(codes),
                                   _lib.current_stream(pts.device)), "morton_codes")
  return codes


def _sorted_codes(points: torch.Tensor, resolution: float):
  codes = morton_codes(points, resolution)
  idx = torch.arange(codes.shape[0], dtype=torch.int32, device=codes.device)
  return radix_sort_pairs(codes, idx, 0, 63)


def argsort(points: torch.Tensor, resolution: float) -> torch.Tensor:
  """Indices (int32) that order the points along the Z-order curve (stable)."""
  return _sorted_codes(points, resolution)[1]


def sort(points: torch.Tensor, resolution: float) -> torch.Tensor:
  return points[argsort(points, resolution).long()]


def argsort_dedup(points: torch.Tensor, resolution: float) -> torch.Tensor:
  """Original indices of one point per occupied grid cell, in Z-order."""
  codes, idx = _sorted_codes(points, resolution)
  if codes.shape[0] == 0:
    return idx
  _, counts = torch.unique_consecutive(codes, return_counts=True)
  last = torch.cumsum(counts, dim=0) - 1
  return idx[last]


def sort_dedup(points: torch.Tensor, resolution: float) -> torch.Tensor:
  return points[argsort_dedup(points, resolution).long()]


__all__ = ["argsort", "sort", "argsort_dedup", "sort_dedup", "morton_codes", "grid_at_resolution"]
