#!/usr/bin/env python
"""Times the three constructions of the overlap lists on one scene through the C-ABI, kernel by kernel:

    python tools/mapper_bins_bench.py [--scene D|C|E|small|big] [--tile 16] [--iters 20]

direct: tile_count + scan + emit_keys64 + 2 radix passes + find_ranges + tile_depth_sort;  bins: memset + histogram + scan
over the tiles + emit_bins + tile_depth_sort_pairs.  Checks that both give the same lists.  One line per stage, then RBINS json.
"""
import argparse
import ctypes
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--scene', default='D')
  ap.add_argument('--n', type=int, default=0)
  ap.add_argument('--tile', type=int, default=16)
  ap.add_argument('--iters', type=int, default=20)
  ap.add_argument('--morton', action='store_true', help='gaussians in Morton order of their image position (coherent atomics)')
  args = ap.parse_args()
  from taichi_splatting_amd import _lib, RasterConfig
  from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
  from taichi_splatting_amd.perspective.projection import project_to_image
  from taichi_splatting_amd.mapper.tile_mapper import map_to_tiles, pad_to_tile
  from taichi_splatting_amd.rendering import ndc_depth
  dev = torch.device('cuda', 0)
  lib = _lib.load()
  cfg = RasterConfig(tile_size=args.tile, pixel_stride=(1, 1) if args.tile == 8 else (2, 2))
  n, side, scale = {'D': (6_000_000, 2048, 1.0), 'C': (1_000_000, 1080, 1.0), 'E': (6_000_000, 4096, 1.0),
                    'big': (1_500_000, 2048, 2.0), 'small': (6_000_000, 2048, 0.5)}[args.scene]
  n = args.n or n
  size = (1920, 1080) if args.scene == 'C' else (side, side)
  torch.manual_seed(0)
  with torch.no_grad():
    cam = random_camera(image_size=size)
    g = random_3d_gaussians(n, cam, scale_factor=scale, alpha_range=(0.1, 0.9), margin=0.0).to(dev)
    cam = cam.to(device=dev)
    g2d, depths, idx = project_to_image(g, cam, cfg)
    depth = ndc_depth(depths, cam.near_plane, cam.far_plane).reshape(-1).contiguous()
    if args.morton:
      xy = (g2d[:, :2] / 4).clamp(0, 4095).to(torch.int64)
      def spread(v):
        v = (v | (v << 8)) & 0x00ff00ff
        v = (v | (v << 4)) & 0x0f0f0f0f
        v = (v | (v << 2)) & 0x33333333
        return (v | (v << 1)) & 0x55555555
      order = torch.argsort(spread(xy[:, 0]) | (spread(xy[:, 1]) << 1))
      g2d, depth = g2d[order].contiguous(), depth[order].contiguous()
  points = g2d.contiguous()
  v = points.shape[0]
  w_pad, h_pad = pad_to_tile(size, args.tile)
  tw, th = w_pad // args.tile, h_pad // args.tile
  tiles = tw * th
  tile_bits = max(1, (tiles - 1).bit_length())
  stream = _lib.current_stream(dev)
  thr = cfg.alpha_threshold

  want_o2p, want_ranges = map_to_tiles(points, depth.reshape(-1, 1), size, cfg, method='direct')
  k = want_o2p.shape[0]
  got_o2p, got_ranges = map_to_tiles(points, depth.reshape(-1, 1), size, cfg, method='bins')
  same = bool(torch.equal(want_o2p, got_o2p) and torch.equal(want_ranges, got_ranges))
  runs = (want_ranges[..., 1] - want_ranges[..., 0]).reshape(-1)
  print(f"scene {args.scene} tile {args.tile} morton {args.morton}: V {v} K {k} tiles {tiles} longest run {int(runs.max())} lists equal: {same}")

  def sized(fn):
    nb = ctypes.c_size_t(0)
    fn(nb)
    return torch.empty((max(nb.value, 1),), dtype=torch.uint8, device=dev)

  counts = torch.empty((v,), dtype=torch.int32, device=dev)
  cum = torch.empty((v + 1,), dtype=torch.int32, device=dev)
  tmp_v = sized(lambda nb: lib.ms_exclusive_scan_i32(None, v, None, None, None, ctypes.byref(nb), stream))
  keys = torch.empty((k,), dtype=torch.int64, device=dev)
  values = torch.empty((k,), dtype=torch.int32, device=dev)
  keys_sorted = torch.empty((k,), dtype=torch.int64, device=dev)
  o2p = torch.empty((k,), dtype=torch.int32, device=dev)
  tmp_k = sized(lambda nb: lib.ms_radix_sort_pairs(None, None, None, None, k, 8, 32, 32 + tile_bits, None, ctypes.byref(nb), stream))
  ranges = torch.empty((tiles, 2), dtype=torch.int32, device=dev)
  tile_counts = torch.empty((tiles,), dtype=torch.int32, device=dev)
  cursor = torch.empty((tiles + 1,), dtype=torch.int32, device=dev)
  starts = torch.empty((tiles + 1,), dtype=torch.int32, device=dev)
  tmp_t = sized(lambda nb: lib.ms_exclusive_scan_i32(None, tiles, None, None, None, ctypes.byref(nb), stream))
  pairs = torch.empty((k,), dtype=torch.int64, device=dev)
  pairs_keep = torch.empty((k,), dtype=torch.int64, device=dev)
  ck = lambda rc: _lib.check(rc, 'mapper_bins_bench')
  dt = _lib.dtype_code(depth.dtype)

  stages_direct = [
    ('tile_count', lambda: ck(lib.ms_tile_count(points.data_ptr(), None, v, w_pad, h_pad, args.tile, thr, 0, th, counts.data_ptr(), None, stream))),
    ('scan over V', lambda: ck(lib.ms_exclusive_scan_i32(counts.data_ptr(), v, cum.data_ptr(), None, tmp_v.data_ptr(), ctypes.byref(ctypes.c_size_t(tmp_v.numel())), stream))),
    ('emit_keys64', lambda: ck(lib.ms_tile_emit_keys64(points.data_ptr(), depth.data_ptr(), dt, cum.data_ptr(), v, w_pad, h_pad, args.tile, thr, 0, th, 0, 0.0, 0.0, keys.data_ptr(), values.data_ptr(), stream))),
    ('radix passes on the tile bits', lambda: ck(lib.ms_radix_sort_pairs(keys.data_ptr(), values.data_ptr(), keys_sorted.data_ptr(), o2p.data_ptr(), k, 8, 32, 32 + tile_bits, tmp_k.data_ptr(), ctypes.byref(ctypes.c_size_t(tmp_k.numel())), stream))),
    ('find_ranges', lambda: ck(lib.ms_find_ranges(keys_sorted.data_ptr(), k, 8, 32, tiles, ranges.data_ptr(), stream))),
    ('tile_depth_sort', lambda: ck(lib.ms_tile_depth_sort(ranges.data_ptr(), tiles, keys_sorted.data_ptr(), o2p.data_ptr(), keys.data_ptr(), stream))),
  ]
  stages_bins = [
    ('zero the tile counts', lambda: tile_counts.zero_()),
    ('tile_histogram', lambda: ck(lib.ms_tile_histogram(points.data_ptr(), v, w_pad, h_pad, args.tile, thr, 0, th, tile_counts.data_ptr(), stream))),
    ('scan over the tiles', lambda: ck(lib.ms_exclusive_scan_i32(tile_counts.data_ptr(), tiles, cursor.data_ptr(), None, tmp_t.data_ptr(), ctypes.byref(ctypes.c_size_t(tmp_t.numel())), stream))),
    ('(copy of the run starts)', lambda: starts.copy_(cursor)),
    ('emit_bins', lambda: ck(lib.ms_tile_emit_bins(points.data_ptr(), depth.data_ptr(), dt, v, w_pad, h_pad, args.tile, thr, 0, th, 0, 0.0, 0.0, k, cursor.data_ptr(), pairs.data_ptr(), stream))),
    ('(copy of the pairs)', lambda: pairs_keep.copy_(pairs)),
    ('tile_depth_sort_pairs', lambda: ck(lib.ms_tile_depth_sort_pairs(want_ranges.data_ptr(), tiles, pairs.data_ptr(), o2p.data_ptr(), keys.data_ptr(), stream))),
  ]

  def time_stage(fn, before=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(args.iters):
      if before:
        before()
      e0.record(); fn(); e1.record()
      e1.synchronize()
      ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]

  # warm the clocks
  t0 = time.time()
  while time.time() - t0 < 1.5:
    for _, fn in stages_direct[:3]:
      fn()
  torch.cuda.synchronize()
  out = {'scene': args.scene, 'tile': args.tile, 'morton': args.morton, 'V': v, 'K': k, 'equal': same}
  for name, seq in (('direct', stages_direct), ('bins', stages_bins)):
    total = 0.0
    for i, (label, fn) in enumerate(seq):
      fn(); torch.cuda.synchronize()
      before = None
      if label == 'tile_histogram':
        before = lambda: tile_counts.zero_()
      if label == 'emit_bins':
        before = lambda: cursor.copy_(starts)
      if label == 'tile_depth_sort_pairs':
        before = lambda: pairs.copy_(pairs_keep)
      if label == 'tile_depth_sort':
        ks, oo = keys_sorted.clone(), o2p.clone()
        before = lambda: (keys_sorted.copy_(ks), o2p.copy_(oo))
      ms = time_stage(fn, before)
      if not label.startswith('('):
        total += ms
      print(f"  {name:7s} {label:32s} {ms:8.4f} ms")
      out[f'{name}:{label}'] = round(ms, 4)
    print(f"  {name:7s} {'TOTAL':32s} {total:8.4f} ms")
    out[f'{name}_total'] = round(total, 4)
  ok = bool(torch.equal(o2p, want_o2p))
  out['bins_o2p_equal_after_timing'] = ok
  print('RBINS ' + json.dumps(out))


if __name__ == '__main__':
  main()
