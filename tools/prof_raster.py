#!/usr/bin/env python
"""Profiling driver: runs the raster forward/backward kernels a few times on a synthetic
scene (for rocprofv3 --pmc / --kernel-trace runs).  python tools/prof_raster.py [n] [size] [tile]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taichi_splatting_amd import RasterConfig, rasterize_with_tiles, map_to_tiles   # noqa: E402
from taichi_splatting_amd.perspective.projection import project_to_image           # noqa: E402
from taichi_splatting_amd.rendering import ndc_depth                                # noqa: E402
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians       # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 16
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = 'cuda:0'
torch.manual_seed(0)
cam = random_camera(image_size=(size, size))
g = random_3d_gaussians(n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9)).to(dev)
cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2))
with torch.no_grad():
  g2d, depths, idx = project_to_image(g, cam.to(device=dev), cfg)
  o2p, ranges = map_to_tiles(g2d, ndc_depth(depths, cam.near_plane, cam.far_plane), cam.image_size, cfg)
feats = g.feature[idx].contiguous()
print(f"V={idx.shape[0]} K={o2p.shape[0]}", flush=True)
for _ in range(iters):
  p = g2d.clone().requires_grad_(True)
  f = feats.clone().requires_grad_(True)
  out = rasterize_with_tiles(p, f, o2p, ranges.view(-1, 2), cam.image_size, cfg)
  out.image.sum().backward()
torch.cuda.synchronize()
print("done", flush=True)
