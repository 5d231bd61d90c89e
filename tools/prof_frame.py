#!/usr/bin/env python
"""Profiling driver: a few whole frames (render_gaussians forward + backward on the frame executor) of a synthetic
scene, for rocprofv3 --pmc / --kernel-trace runs over EVERY kernel of the frame (tools/pmc_frame.sh).
  python tools/prof_frame.py [n] [size] [tile] [frames]     prints V and K (algorithmic bytes need them)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taichi_splatting_amd import RasterConfig, frame, render_gaussians       # noqa: E402
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
size = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
tile = int(sys.argv[3]) if len(sys.argv) > 3 else 16
frames = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = 'cuda:0'
torch.manual_seed(0)
cam = random_camera(image_size=(size, size))
g = random_3d_gaussians(n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.0)
g = g.replace(feature=(torch.rand(n, 3, 16) - 0.5) * 0.5).to(dev).requires_grad_(True)
cam = cam.to(device=dev)
cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2))
leaves = (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)
for i in range(frames):
  for t in leaves:
    t.grad = None
  r = render_gaussians(g, cam, cfg, use_sh=True)
  r.image.sum().backward()
  if i == 0:
    st = frame.frame_status(r)
    print(f"N={n} V={int(r.points.idx.shape[0])} K={st['overlaps']} size={size} tile={tile}", flush=True)
torch.cuda.synchronize()
print("done", flush=True)
