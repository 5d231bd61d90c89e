#!/usr/bin/env python
"""Time the raster forward kernel (config D scene) through the C-ABI: python tools/ab_forward.py [reps]
(select a variant library with MS_SPLAT_LIB=tools/variants/lib<name>.so)"""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench
import argparse
args = argparse.Namespace(n=6_000_000, size=2048, height=None, tile=16, sh_degree=3, seed=0)
from taichi_splatting_amd import RasterConfig, map_to_tiles, rasterize_with_tiles
from taichi_splatting_amd.perspective.projection import project_to_image
from taichi_splatting_amd.spherical_harmonics import evaluate_sh_at
from taichi_splatting_amd.rendering import ndc_depth
dev = torch.device('cuda', 0)
g, cam = bench.make_scene(args, dev)
cfg = RasterConfig()
with torch.no_grad():
  p, d, idx = project_to_image(g, cam, cfg)
  f = evaluate_sh_at(g.feature, g.position, idx, cam.camera_position)
  o2p, ranges = map_to_tiles(p, ndc_depth(d, cam.near_plane, cam.far_plane), cam.image_size, cfg)
  r2 = ranges.view(-1, 2)
  fn = lambda: rasterize_with_tiles(p, f, o2p, r2, cam.image_size, cfg)
  for _ in range(20): fn()
  out = []
  for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    out.append(round(bench.cuda_time_ms(fn, iters=50, warmup=5), 4))
print("raster forward ms:", out)
