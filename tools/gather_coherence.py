"""Is the raster forward / backward bound by the random gather of splat rows?  Same scene, gaussians permuted so that
the splats of a tile are neighbours in memory (sorted by the tile of their centre)."""
import sys; sys.path.insert(0, '.')
import torch
from taichi_splatting_amd import RasterConfig, rasterize_with_tiles, map_to_tiles, _lib
from taichi_splatting_amd.perspective.projection import project_to_image
from taichi_splatting_amd.rendering import ndc_depth
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
from taichi_splatting_amd.benchmarks.components import time_ms
dev = 'cuda:0'
torch.manual_seed(0)
size = (2048, 2048)
cam = random_camera(image_size=size)
g = random_3d_gaussians(6_000_000, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.0).to(dev)
cfg = RasterConfig()
lib = _lib.load()
with torch.no_grad():
  p, d, idx = project_to_image(g, cam.to(device=dev), cfg)
  f = g.feature[idx].contiguous()
  for name in ('random order', 'sorted by tile of the centre'):
    if name != 'random order':
      tile = (p[:, 1] / 16).floor().clamp(0, 127).long() * 128 + (p[:, 0] / 16).floor().clamp(0, 127).long()
      perm = torch.argsort(tile)
      p, d, f = p[perm].contiguous(), d[perm].contiguous(), f[perm].contiguous()
    o2p, ranges = map_to_tiles(p, ndc_depth(d, cam.near_plane, cam.far_plane), size, cfg)
    fwd = lambda: rasterize_with_tiles(p, f, o2p, ranges.view(-1, 2), size, cfg)
    image = fwd().image
    G = torch.ones_like(image)
    mom = torch.zeros((p.shape[0], 16), device=dev)
    cfg_c = _lib.raster_config_c(cfg)
    bwd = lambda: lib.ms_raster_bwd_moments(p.data_ptr(), f.data_ptr(), ranges.data_ptr(), o2p.data_ptr(), image.data_ptr(), G.data_ptr(), 2048, 2048, cfg_c, mom.data_ptr(), 0, None, 0, 128, _lib.current_stream(p.device))
    tmap = time_ms(lambda: map_to_tiles(p, ndc_depth(d, cam.near_plane, cam.far_plane), size, cfg), iters=20)
    print(f"{name}: K={o2p.shape[0]} forward {time_ms(fwd, iters=50):.3f} ms, backward kernel {time_ms(bwd, iters=20):.3f} ms, map_to_tiles {tmap:.3f} ms")
