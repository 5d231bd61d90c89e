#!/usr/bin/env python
"""Secondary 'dense' profile mirroring the reference's benchmarks/bench_rasterizer.py:21-26,50-51
(1 M random 2D gaussians, 1024x768, scale_factor 4, alpha in (0.75, 1), depth in (0.1, 100), tile 16):
forward, forward+visibility, backward (features / gaussians / all), point heuristics, tile mapper."""
import sys
from dataclasses import replace
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taichi_splatting_amd import RasterConfig, map_to_tiles, rasterize_with_tiles   # noqa: E402
from taichi_splatting_amd.misc.renderer2d import project_gaussians2d               # noqa: E402
from taichi_splatting_amd.testing import random_2d_gaussians                      # noqa: E402


def timed(name, f, iters=20, warmup=3):
  for _ in range(warmup):
    f()
  torch.cuda.synchronize()
  s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  s.record()
  for _ in range(iters):
    f()
  e.record()
  torch.cuda.synchronize()
  ms = s.elapsed_time(e) / iters
  print(f"{name:40s} {ms:8.3f} ms  ({1000 / ms:8.1f} it/s)")


def main():
  n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
  size = (1024, 768)
  scale = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
  torch.manual_seed(0)
  g = random_2d_gaussians(n, size, num_channels=3, scale_factor=scale, alpha_range=(0.75, 1.0),
                          depth_range=(0.1, 100.)).to('cuda:0')
  cfg = RasterConfig(tile_size=16)
  p = project_gaussians2d(g)
  o2p, ranges = map_to_tiles(p, g.depths, size, cfg)
  per_tile = (ranges[..., 1] - ranges[..., 0]).float()
  print(f"n={n} scale_factor={scale} K={o2p.shape[0]} point_overlap={o2p.shape[0] / n:.2f} tile_points={per_tile.mean():.1f} max={int(per_tile.max())}")
  r2 = ranges.view(-1, 2)
  timed('map_to_tiles', lambda: map_to_tiles(p, g.depths, size, cfg))
  with torch.no_grad():
    timed('forward', lambda: rasterize_with_tiles(p, g.feature, o2p, r2, size, cfg))
    timed('forward_vis', lambda: rasterize_with_tiles(p, g.feature, o2p, r2, size, replace(cfg, compute_visibility=True)))

  def backward(pg, fg, c=cfg):
    pp = p.detach().requires_grad_(pg)
    ff = g.feature.detach().requires_grad_(fg)
    rasterize_with_tiles(pp, ff, o2p, r2, size, c).image.sum().backward()
  timed('backward (features)', lambda: backward(False, True))
  timed('backward (gaussians)', lambda: backward(True, False))
  timed('backward (all)', lambda: backward(True, True))
  timed('backward (compute_point_heuristic)', lambda: backward(True, True, replace(cfg, compute_point_heuristic=True)))


if __name__ == '__main__':
  main()
