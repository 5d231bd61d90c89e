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
