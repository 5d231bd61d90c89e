"""Tile mapper: which gaussians touch which screen tiles, depth sorted per tile.

Same interface as reference ``mapper/tile_mapper.py:204-225`` (``map_to_tiles``) and ``:20-24``
(``pad_to_tile``).  Stages (csrc/mapper.hip, csrc/scan_sort.hip):
depth pre-sort of the V gaussians (stable radix sort of the 32 bit depth keys) -> OBB-vs-tile overlap
count in depth order -> exclusive scan (one host read of the total K) -> emission of (tile id, point)
-> stable radix sort on the ceil(log2 T) tile-id bits -> per-tile ranges.  The result is the order
of the reference's single 48-bit sort of ``tile_id << 32 | float_bits(depth)`` keys generated in
point order (tile, depth bits, point index) at a third of the sorted bytes.  Unlike the reference (``tile_mapper.py:177-178``) the tile id is not limited to
16 bits, so 2048x2048 @ tile 8 and 4096x4096 @ tile 16 (65536 tiles) work.
"""
from __future__ import annotations

import ctypes
import math
from numbers import Integral
from typing import Optional, Tuple

import torch

from .. import _lib
from ..data_types import RasterConfig


# overlaps per gaussian above which the pre-sort sequence moves fewer bytes than the direct one (frame.py has the
# measurements; there the choice is made from the previous frame's total, here from this call's own)
PRESORT_ABOVE = 3.5


def pad_to_tile(image_size: Tuple[Integral, Integral], tile_size: int):
  def pad(x):
    return int(math.ceil(x / tile_size) * tile_size)
  return tuple(pad(x) for x in image_size)


def map_to_tiles_strip(gaussians: torch.Tensor, depth: torch.Tensor,
                       image_size: Tuple[Integral, Integral], config: RasterConfig,
                       use_depth16: bool = False,
                       tile_rows: Optional[Tuple[int, int]] = None,
                       ndc_range: Optional[Tuple[float, float]] = None,
                       method: Optional[str] = None) -> Tuple[torch.Tensor, torch.Tensor]:
  """``map_to_tiles`` restricted to tile rows [tile_rows[0], tile_rows[1]) (multi-GPU strips).

  ``method``: ``'direct'`` (storage-order emission, stable sort on the tile bits, per-tile depth sort) or
  ``'presort'`` (gaussians sorted by depth first, overlaps sorted by tile id) — two constructions of the SAME lists
  (tile, depth key, point index); ``None`` picks by overlaps per gaussian (``PRESORT_ABOVE``), which is known here
  before anything is emitted.  ``'direct'`` sorts a tile run beyond 5120 entries with one workgroup (~12 ns per entry):
  pass ``method='presort'`` for scenes that pile tens of thousands of splats on one tile (the frame executor notices
  such runs by itself, ``frame.LONG_RUN_LIMIT``; this operator would need a second host synchronisation to do so).

  ``tile_ranges`` is still indexed by the global tile id; tiles outside the strip are empty.
  ``ndc_range=(near, far)``: ``depth`` holds camera depths and is converted to ndc depth inside the
  key kernel (what ``render_projected`` does with torch ops in the reference, renderer.py:67).
  """
  lib = _lib.load()
  _lib.require_gpu(gaussians, depth)
  assert gaussians.ndim == 2 and gaussians.shape[1] == 7, f"gaussians must be Nx7 got {gaussians.shape}"
  assert depth.ndim == 2 and depth.shape[1] == 1, f"depths must be Nx1, got {depth.shape}"
  assert gaussians.shape[0] == depth.shape[0], f"size mismatch {gaussians.shape} {depth.shape}"

  tile_size = config.tile_size
  w_pad, h_pad = pad_to_tile(image_size, tile_size)
  tile_shape = (h_pad // tile_size, w_pad // tile_size)
  num_tiles = tile_shape[0] * tile_shape[1]
  if use_depth16:
    assert num_tiles <= 65536, \
      f"tile dimensions {tile_shape} for image size {(w_pad, h_pad)} exceed the 16 bit tile id of use_depth16 keys"
  assert num_tiles < (1 << 31), "too many tiles"

  row_begin, row_end = (0, tile_shape[0]) if tile_rows is None else tile_rows
  device = gaussians.device
  stream = _lib.current_stream(device)

  with torch.no_grad():
    # the overlap test runs in float32 like the reference (Gaussian2D from taichi_lib.f32)
    points = gaussians.detach().to(torch.float32).contiguous()
    depths = depth.detach().reshape(-1).contiguous()
    if ndc_range is None or depths.dtype not in (torch.float32, torch.float64):
      depths = depths.to(torch.float32)
    v = points.shape[0]

    # ms_find_ranges writes every entry (zero fill + ranges): only the early returns need zeros from here
    tile_ranges = torch.empty((*tile_shape, 2), dtype=torch.int32, device=device)
    if v == 0:
      return torch.empty((0,), dtype=torch.int32, device=device), tile_ranges.zero_()

    def scratch(nbytes):
      return torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=device)

    def sort_pairs(keys_in, vals_in, key_bytes, begin_bit, end_bit):
      n = keys_in.shape[0]
      keys_out, vals_out = torch.empty_like(keys_in), torch.empty_like(vals_in)
      nb = ctypes.c_size_t(0)
      _lib.check(lib.ms_radix_sort_pairs(None, None, None, None, n, key_bytes, begin_bit, end_bit, None,
                                         ctypes.byref(nb), stream), "map_to_tiles")
      tmp = scratch(nb.value)
      _lib.check(lib.ms_radix_sort_pairs(keys_in.data_ptr(), vals_in.data_ptr(), keys_out.data_ptr(),
                                         vals_out.data_ptr(), n, key_bytes, begin_bit, end_bit, tmp.data_ptr(),
                                         ctypes.byref(nb), stream), "map_to_tiles")
      return keys_out, vals_out

    near, far = (0.0, 0.0) if ndc_range is None else (float(ndc_range[0]), float(ndc_range[1]))
    tile_bits = max(1, (num_tiles - 1).bit_length())

    def exclusive_scan(counts, want_total):
      cum = torch.empty((v + 1,), dtype=torch.int32, device=device)
      nbytes = ctypes.c_size_t(0)
      _lib.check(lib.ms_exclusive_scan_i32(None, v, None, None, None, ctypes.byref(nbytes), stream), "map_to_tiles")
      tmp = scratch(nbytes.value)
      _lib.check(lib.ms_exclusive_scan_i32(counts.data_ptr(), v, cum.data_ptr(), None, tmp.data_ptr(),
                                           ctypes.byref(nbytes), stream), "map_to_tiles")
      return cum, (int(cum[v].item()) if want_total else None)

    def checked(total):
      if total < 0:
        raise OverflowError("map_to_tiles: more than 2^31 - 1 tile overlaps (the overlap index is int32 like the "
                            "reference's, tile_mapper.py:150); use a larger tile size or fewer / smaller gaussians")
      return total

    counts = torch.empty((v,), dtype=torch.int32, device=device)
    cum = total = None
    if method is None and use_depth16:
      # 16 bit keys: the pre-sort is two passes over v pairs and the tile sort moves 4-byte keys — always cheaper
      method = 'presort'
    if method != 'presort':
      # 1. overlap counts in storage order (a streaming pass), exclusive scan, total K (the one host sync of the mapper)
      _lib.check(lib.ms_tile_count(points.data_ptr(), None, v, w_pad, h_pad, tile_size, config.alpha_threshold,
                                   row_begin, row_end, counts.data_ptr(), None, stream), "map_to_tiles")
      cum, total = exclusive_scan(counts, True)
      if checked(total) == 0:
        return torch.empty((0,), dtype=torch.int32, device=device), tile_ranges.zero_()
      if method is None:
        method = 'presort' if total > PRESORT_ABOVE * v else 'direct'      # presort: counted again below, in depth order

    if method == 'direct':
      # 2. keys tile << 32 | depth key (ndc / 16 bit quantisation fused) in storage order; 3. STABLE sort on the tile
      #    bits only: every tile's run keeps ascending point indices; 4. ranges; 5. each run sorted by (depth key,
      #    point index) by one workgroup (csrc/tile_sort.hip).  Same lists as the full 64 bit sort, 2 passes over K.
      keys = torch.empty((total,), dtype=torch.int64, device=device)
      values = torch.empty((total,), dtype=torch.int32, device=device)
      _lib.check(lib.ms_tile_emit_keys64(points.data_ptr(), depths.data_ptr(), _lib.dtype_code(depths.dtype), cum.data_ptr(),
                                         v, w_pad, h_pad, tile_size, config.alpha_threshold, row_begin, row_end,
                                         int(use_depth16), near, far, keys.data_ptr(), values.data_ptr(), stream),
                 "map_to_tiles")
      keys_sorted, overlap_to_point = sort_pairs(keys, values, 8, 32, 32 + tile_bits)
      _lib.check(lib.ms_find_ranges(keys_sorted.data_ptr(), total, 8, 32, num_tiles, tile_ranges.data_ptr(), stream),
                 "map_to_tiles")
      _lib.check(lib.ms_tile_depth_sort(tile_ranges.data_ptr(), num_tiles, keys_sorted.data_ptr(),
                                        overlap_to_point.data_ptr(), keys.data_ptr(), stream), "map_to_tiles")
      return overlap_to_point, tile_ranges

    assert method == 'presort', f"map_to_tiles: unknown method {method!r}"
    # 2. depth pre-sort of the V gaussians (stable: ties keep point order): 32 bit keys, 4 radix passes over V pairs
    #    (2 for depth16; passes whose digit is the same in every key are skipped) instead of 4 of the 6 passes over
    #    the K overlaps; the first pass makes the keys from the depths itself (ms_depth_argsort)
    sorted_keys = torch.empty((v,), dtype=torch.int32, device=device)
    order = torch.empty((v,), dtype=torch.int32, device=device)
    nb = ctypes.c_size_t(0)
    _lib.check(lib.ms_depth_argsort(None, v, int(use_depth16), near, far, _lib.dtype_code(depths.dtype), None, None, None,
                                    ctypes.byref(nb), stream), "map_to_tiles")
    tmp = scratch(nb.value)
    _lib.check(lib.ms_depth_argsort(depths.data_ptr(), v, int(use_depth16), near, far, _lib.dtype_code(depths.dtype),
                                    sorted_keys.data_ptr(), order.data_ptr(), tmp.data_ptr(), ctypes.byref(nb), stream),
               "map_to_tiles")

    # 3. overlap counts again, in depth order, with the depth-ordered copy of the rows the emit re-reads linearly
    ordered = torch.empty((v, 7), dtype=torch.float32, device=device)
    _lib.check(lib.ms_tile_count(points.data_ptr(), order.data_ptr(), v, w_pad, h_pad, tile_size,
                                 config.alpha_threshold, row_begin, row_end, counts.data_ptr(),
                                 ordered.data_ptr(), stream), "map_to_tiles")
    if total is None:                            # the pre-sort was asked for: this scan's total is the host sync
      cum, total = exclusive_scan(counts, True)
      if checked(total) == 0:
        return torch.empty((0,), dtype=torch.int32, device=device), tile_ranges.zero_()
    else:
      cum, _ = exclusive_scan(counts, False)

    # 4. emit (tile id, point) in depth order; 5. STABLE sort on the tile id bits only
    keys = torch.empty((total,), dtype=torch.int32, device=device)
    values = torch.empty((total,), dtype=torch.int32, device=device)
    _lib.check(lib.ms_tile_emit(ordered.data_ptr(), None, order.data_ptr(), cum.data_ptr(), v, w_pad, h_pad,
                                tile_size, config.alpha_threshold, row_begin, row_end, 2, 1,
                                keys.data_ptr(), values.data_ptr(), stream), "map_to_tiles")
    keys_sorted, overlap_to_point = sort_pairs(keys, values, 4, 0, tile_bits)

    # 6. per-tile ranges
    _lib.check(lib.ms_find_ranges(keys_sorted.data_ptr(), total, 4, 0, num_tiles,
                                  tile_ranges.data_ptr(), stream), "map_to_tiles")
    return overlap_to_point, tile_ranges


def map_to_tiles(gaussians: torch.Tensor, depth: torch.Tensor, image_size: Tuple[Integral, Integral],
                 config: RasterConfig, use_depth16: bool = False, *, method: Optional[str] = None
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
  """Maps gaussians to tiles, sorted by depth (front to back).

  Parameters:
    gaussians: (N, 7) packed 2D gaussians
    depth: (N, 1) depths (sorted as float32, must be >= 0)
    image_size: (width, height)
    config: RasterConfig (tile_size, alpha_threshold)

  Returns:
    overlap_to_point: (K,) int32, overlap index -> point index
    tile_ranges: (TH, TW, 2) int32, tile -> [start, end) range of overlap indices
  """
  return map_to_tiles_strip(gaussians, depth, image_size, config, use_depth16=use_depth16, method=method)


def with_reference_tail(overlap_to_point: torch.Tensor, tile_ranges: torch.Tensor, tile_size: int,
                        group: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
  """DEBUG AID, forward only: tile lists extended so that the kernels visit splats in the order the REFERENCE's
  kernels do, including the revisits caused by its in-group loop bound.

  The reference walks a tile's list in groups of ``group`` = tile_size^2 splats staged in shared memory, with the
  in-group bound ``min(group, count - group_index)`` where ``count - group_index * group`` is meant
  (``rasterizer/forward.py:86-89``): after the valid entries of the LAST, partially filled group it keeps going and
  blends again whatever the PREVIOUS group left in shared memory at those slots.  This library visits every splat
  exactly once, so on tiles with more than ``group`` splats its image differs from Taichi's output (config-D
  density: 57 % of the pixels by more than 1e-4, at most 5.6e-2).  To diff against an image
  rendered by the reference, rasterize with the lists returned here::

      o2p, ranges = map_to_tiles(gaussians2d, depths, image_size, config)
      o2p_ref, ranges_ref = with_reference_tail(o2p, ranges, config.tile_size)
      image_like_taichi = rasterize_with_tiles(gaussians2d, features, o2p_ref, ranges_ref.view(-1, 2),
                                               image_size, config).image          # under torch.no_grad()

  Forward only: the reference drops the gradients of the revisited entries (``backward.py:214``) while a backward pass
  over these lists would count them.  Pure index arithmetic on the device (no kernel of its own)."""
  assert overlap_to_point.dtype == torch.int32 and tile_ranges.dtype == torch.int32
  g = int(group) if group is not None else int(tile_size) * int(tile_size)
  shape = tile_ranges.shape
  r = tile_ranges.reshape(-1, 2).long()
  start, count = r[:, 0], r[:, 1] - r[:, 0]
  last = torch.clamp(count - 1, min=0) // g                     # index of the last group
  valid = count - last * g                                      # its valid entries
  visited = torch.minimum(torch.full_like(count, g), count - last)
  tail = torch.where(last > 0, torch.clamp(visited - valid, min=0), torch.zeros_like(count))
  new_count = count + tail
  new_end = torch.cumsum(new_count, dim=0)
  new_start = new_end - new_count
  total = int(new_end[-1].item()) if new_end.numel() else 0
  out = torch.empty((total,), dtype=torch.int32, device=overlap_to_point.device)
  if total:
    tile = torch.repeat_interleave(torch.arange(r.shape[0], device=r.device), new_count)
    k = torch.arange(total, device=r.device) - new_start[tile]
    # k < count: the list itself; beyond it: slot (valid + k - count) of the previous group
    stale = (last[tile] - 1) * g + valid[tile] + (k - count[tile])
    src = start[tile] + torch.where(k < count[tile], k, stale)
    out = overlap_to_point[src]
  # empty tiles keep [0, 0) like find_ranges
  new_ranges = torch.stack([torch.where(new_count > 0, new_start, torch.zeros_like(new_start)),
                            torch.where(new_count > 0, new_end, torch.zeros_like(new_end))], dim=1).to(torch.int32)
  return out, new_ranges.reshape(shape)
