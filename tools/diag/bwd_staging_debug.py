"""Which list positions of a tile get wrong gradients from the scan backward? (development aid)"""
import os, sys, torch
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from taichi_splatting_amd import RasterConfig, map_to_tiles, rasterize_with_tiles
from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
from taichi_splatting_amd.testing import random_2d_gaussians
DEV = 'cuda:0'
for per_tile in [int(a) for a in sys.argv[1:]] or [530]:
  size = (64, 48); n = per_tile * 12
  torch.manual_seed(per_tile)
  g = random_2d_gaussians(n, size, scale_factor=0.6, alpha_range=(0.004, 0.02))
  tiles = torch.arange(n) % 12
  g.position[:] = torch.stack([(tiles % 4) * 16 + 3.0 + 10.0 * torch.rand(n), (tiles // 4) * 16 + 3.0 + 10.0 * torch.rand(n)], 1)
  g.log_scaling[:] = torch.log(0.5 + 0.4 * torch.rand(n, 2))
  cfg = RasterConfig(tile_size=16)
  p0, depth, f0 = project_gaussians2d(g).to(DEV), g.depths.reshape(-1, 1).to(DEV), g.feature.to(DEV)
  o2p, ranges = map_to_tiles(p0, depth, size, cfg)
  torch.manual_seed(1)
  G = torch.rand(size[1], size[0], 3, device=DEV) + 0.5
  grads = {}
  for mode in ('scan', 'patch'):
    os.environ['MS_RASTER_BWD'] = mode
    p, f = p0.clone().requires_grad_(True), f0.clone().requires_grad_(True)
    out = rasterize_with_tiles(p, f, o2p, ranges.view(-1, 2), size, cfg)
    (out.image * G).sum().backward()
    grads[mode] = torch.cat([p.grad, f.grad], 1).clone()
  err = (grads['scan'] - grads['patch']).abs().max(1).values / grads['patch'].abs().max()
  bad = (err > 1e-4).nonzero().flatten()
  r2 = ranges.view(-1, 2).cpu(); o = o2p.cpu()
  print(f"per_tile {per_tile}: {bad.numel()} bad splats of {n}; runs {(r2[:,1]-r2[:,0]).tolist()}")
  where = {}
  for t in range(r2.shape[0]):
    s, e = int(r2[t, 0]), int(r2[t, 1])
    for pos in range(s, e):
      where.setdefault(int(o[pos]), []).append((t, pos - s))
  for b in bad[:40].tolist():
    print("  splat", b, "err", float(err[b]), "in lists", where.get(b))
