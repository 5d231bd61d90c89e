"""-m gpu: deterministic backward (SURVEY.md 5 "race detection", 7 "Determinism").  Float atomics add in arrival
order, so the default gradients carry run-to-run noise in the last bits; with
``rasterizer.function.DETERMINISTIC_BACKWARD`` the per-patch sums are committed as fixed-point integers and the
whole backward pass is bitwise reproducible (the forward pass always is)."""
import pytest
import torch

from taichi_splatting_amd import RasterConfig, render_gaussians
from taichi_splatting_amd.rasterizer import function as raster_function
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
LEAVES = ('position', 'log_scaling', 'rotation', 'alpha_logit', 'feature')


def frame(g, cam, cfg):
  for k in LEAVES:
    getattr(g, k).grad = None
  r = render_gaussians(g, cam, cfg, use_sh=True)
  r.points.gaussians2d.retain_grad()
  r.points.features.retain_grad()
  r.image.sum().backward()
  return ([r.image.detach().clone(), r.points.gaussians2d.grad.clone(), r.points.features.grad.clone()]
          + [getattr(g, k).grad.clone() for k in LEAVES])


@pytest.mark.parametrize('n,size,tile', [(300_000, (640, 480), 16), (300_000, (640, 480), 8), (2_000_000, (1024, 1024), 16)])
def test_deterministic_backward_is_bitwise_repeatable(n, size, tile, monkeypatch):
  torch.manual_seed(0)
  cam = random_camera(image_size=size)
  g = random_3d_gaussians(n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.0)
  g = g.replace(feature=(torch.rand(n, 3, 16) - 0.5) * 0.5).to(DEV).requires_grad_(True)
  cam = cam.to(device=DEV)
  cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2))

  monkeypatch.setattr(raster_function, 'DETERMINISTIC_BACKWARD', False)
  plain = frame(g, cam, cfg)
  monkeypatch.setattr(raster_function, 'DETERMINISTIC_BACKWARD', True)
  runs = [frame(g, cam, cfg) for _ in range(3)]
  for other in runs[1:]:
    for a, b in zip(runs[0], other):
      assert torch.equal(a, b)                                  # image, 2D-boundary and 3D gradients: bit for bit
  # and it is the same gradient: the fixed-point commit rounds each per-patch sum to 2^-32; the float-atomic
  # result itself moves by ~1e-5 of the largest gradient from run to run
  for a, b in zip(runs[0], plain):
    scale = b.abs().max().item()
    assert (a - b).abs().max().item() <= 1e-4 * scale


def test_deterministic_backward_dense_scene_with_heuristics(monkeypatch):
  """Dense 2D scene (about 1500 splats per tile: several passes per staged batch in the backward kernel), point
  heuristics on: the 2D gradients and heuristics are bitwise repeatable in deterministic mode."""
  from taichi_splatting_amd import map_to_tiles, rasterize_with_tiles
  from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
  from taichi_splatting_amd.testing import random_2d_gaussians
  torch.manual_seed(0)
  size = (512, 384)
  g = random_2d_gaussians(250_000, size, scale_factor=4.0, alpha_range=(0.75, 1.0), depth_range=(0.1, 100.)).to(DEV)
  cfg = RasterConfig(compute_point_heuristic=True)
  p, f = project_gaussians2d(g), g.feature.contiguous()
  o2p, ranges = map_to_tiles(p, g.depths, size, cfg)
  assert o2p.shape[0] / ranges[..., 0].numel() > 1000

  def run():
    pg, fg = p.clone().requires_grad_(True), f.clone().requires_grad_(True)
    out = rasterize_with_tiles(pg, fg, o2p, ranges.view(-1, 2), size, cfg)
    out.image.sum().backward()
    return pg.grad, fg.grad, out.point_heuristic.clone()
  monkeypatch.setattr(raster_function, 'DETERMINISTIC_BACKWARD', True)
  a, b = run(), run()
  for x, y in zip(a, b):
    assert torch.equal(x, y) and torch.isfinite(x).all() and float(x.abs().sum()) > 0
  monkeypatch.setattr(raster_function, 'DETERMINISTIC_BACKWARD', False)
  for x, y in zip(a, run()):
    assert (x - y).abs().max().item() <= 1e-4 * y.abs().max().item()
