#!/usr/bin/env python
"""Index-width check far above the bench sizes: one forward + backward of 64 M gaussians (12 GB of SH parameters: element
indexes beyond 2^31) and the gradient statistics of the first and the last million rows side by side — a truncated index
would leave the tail without gradients or with garbage.  python tools/check_large_scene.py"""
import sys, torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from taichi_splatting_amd import RasterConfig, render_gaussians
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
dev = torch.device('cuda', 0)
n, size = 64_000_000, 4096
torch.manual_seed(0)
cam = random_camera(image_size=(size, size))
g = random_3d_gaussians(n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.0)
g = g.replace(feature=(torch.rand(n, 3, 16) - 0.5) * 0.5).to(dev)
cam = cam.to(device=dev)
g.requires_grad_(True)
r = render_gaussians(g, cam, RasterConfig(), use_sh=True)
r.image.sum().backward()
torch.cuda.synchronize()
print('image finite', bool(torch.isfinite(r.image).all()), 'alpha range', float(r.image_weight.min()), float(r.image_weight.max()))
for name in ('position', 'log_scaling', 'rotation', 'alpha_logit', 'feature'):
  gr = getattr(g, name).grad
  flat = gr.reshape(n, -1)
  nz = (flat.abs().sum(1) > 0)
  print(name, 'finite', bool(torch.isfinite(gr).all()), 'rows with a gradient: first million', int(nz[:1_000_000].sum()), 'last million', int(nz[-1_000_000:].sum()),
        'mean |g| first/last million', float(flat[:1_000_000].abs().mean()), float(flat[-1_000_000:].abs().mean()))
