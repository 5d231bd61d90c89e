"""Diagnostic: the visibility identity (dL/dimage = 1: d feature[:, 0] == visibility) at full size: how many gaussians
differ, by how much (forward and backward evaluate the blend gate with their own arithmetic)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from tests.test_gpu_fullsize import scene
from taichi_splatting_amd import RasterConfig, map_to_tiles, rasterize_with_tiles
from taichi_splatting_amd.perspective.projection import project_to_image
from taichi_splatting_amd.rendering import ndc_depth

for n, size, tile in ((1_000_000, 1024, 16), (6_000_000, 2048, 16)):
  g, cam, cfg = scene(n, size, tile)
  with torch.no_grad():
    p, depth, idx = project_to_image(g, cam, cfg)
    o2p, ranges = map_to_tiles(p, ndc_depth(depth, cam.near_plane, cam.far_plane), cam.image_size, cfg)
  feats = g.feature.contiguous().requires_grad_(True)
  cfg_v = RasterConfig(tile_size=tile, pixel_stride=cfg.pixel_stride, compute_visibility=True)
  out = rasterize_with_tiles(p, feats, o2p, ranges.view(-1, 2), cam.image_size, cfg_v)
  out.image.sum().backward()
  d = (feats.grad[:, 0] - out.visibility).abs()
  tol = 2e-3 + 2e-3 * out.visibility.abs()
  print(n, size, 'max', float(d.max()), 'nan', int(torch.isnan(d).sum()), 'beyond tol', int((d > tol).sum()), 'beyond 1e-3', int((d > 1e-3).sum()),
        'worst at', int(d.argmax()), float(feats.grad[d.argmax(), 0]), float(out.visibility[d.argmax()]))
