"""-m gpu: the tile-strip sharding used for multi-GPU rendering, emulated on ONE GPU: every "rank"
renders its strip of the same frame in turn (no process group: the all-reduce is the identity), and
the sum of the per-strip gradients / the union of the strips must equal the full-frame render."""
import pytest
import torch

from taichi_splatting_amd import RasterConfig, render_gaussians
from taichi_splatting_amd.distributed import render_strip_step, strip_rows
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('world,dtype', [(2, torch.float64), (3, torch.float64), (8, torch.float32)])
def test_strips_compose_to_full_frame(world, dtype):
  torch.manual_seed(world)
  size = (320, 208)
  cam = random_camera(image_size=size)
  n = 20000
  g = random_3d_gaussians(n, cam, scale_factor=1.5, alpha_range=(0.1, 0.9))
  g = g.replace(feature=(torch.rand(n, 3, 16) - 0.5) * 0.5).to(dtype=dtype)
  cam = cam.to(device=DEV, dtype=dtype)
  cfg = RasterConfig()
  torch.manual_seed(0)
  G = torch.randn(size[1], size[0], 3, dtype=dtype, device=DEV)

  full = g.to(DEV).requires_grad_(True)
  r = render_gaussians(full, cam, cfg, use_sh=True)
  (r.image * G).sum().backward()

  image = torch.zeros_like(r.image)
  sums = None
  for rank in range(world):
    part = g.to(DEV).requires_grad_(True)
    rows = strip_rows((size[1] + 15) // 16, world, rank)
    rendering, _ = render_strip_step(part, cam, cfg, lambda img, px: (img * G).sum(), use_sh=True,
                                     rank=rank, world_size=world)
    y0, y1 = rows[0] * 16, min(rows[1] * 16, size[1])
    assert float(rendering.image.detach()[:y0].abs().sum()) == 0 and float(rendering.image.detach()[y1:].abs().sum()) == 0
    image[y0:y1] = rendering.image[y0:y1]
    grads = [part.position.grad, part.log_scaling.grad, part.rotation.grad, part.alpha_logit.grad, part.feature.grad]
    sums = grads if sums is None else [a + b for a, b in zip(sums, grads)]

  tol = 1e-10 if dtype == torch.float64 else 1e-5
  assert torch.allclose(image, r.image, atol=tol)
  for got, want in zip(sums, [full.position.grad, full.log_scaling.grad, full.rotation.grad,
                              full.alpha_logit.grad, full.feature.grad]):
    scale = max(1.0, want.abs().max().item())
    assert torch.allclose(got, want, atol=(1e-8 if dtype == torch.float64 else 2e-3) * scale), \
      ((got - want).abs().max(), scale)


@pytest.mark.parametrize('features', [3, 6])
def test_cropped_strip_equals_rows_of_full_frame(features):
  # rasterize_with_tiles(..., tile_rows, crop_to_rows=True): only the strip's pixel rows are allocated
  from taichi_splatting_amd import rasterize_with_tiles, map_to_tiles
  from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
  from taichi_splatting_amd.testing import random_2d_gaussians
  torch.manual_seed(features)
  size = (200, 150)            # 10 tile rows, the last one partial
  cfg = RasterConfig()
  g = random_2d_gaussians(5000, size, num_channels=features, scale_factor=1.5).to(DEV)
  p = project_gaussians2d(g)
  o2p, ranges = map_to_tiles(p, g.depths, size, cfg)
  G = torch.randn(size[1], size[0], features, device=DEV)
  pf = p.clone().requires_grad_(True); ff = g.feature.clone().requires_grad_(True)
  full = rasterize_with_tiles(pf, ff, o2p, ranges.view(-1, 2), size, cfg)
  gp_sum, gf_sum = torch.zeros_like(p), torch.zeros_like(g.feature)
  for rows in ((0, 3), (3, 3), (3, 9), (9, 10)):
    ps = p.clone().requires_grad_(True); fs = g.feature.clone().requires_grad_(True)
    y0, y1 = rows[0] * 16, min(rows[1] * 16, size[1])
    out = rasterize_with_tiles(ps, fs, o2p, ranges.view(-1, 2), size, cfg, tile_rows=rows, crop_to_rows=True)
    assert out.image.shape == (y1 - y0, size[0], features) and out.image_weight.shape == (y1 - y0, size[0])
    assert torch.equal(out.image, full.image[y0:y1]) and torch.equal(out.image_weight, full.image_weight[y0:y1])
    (out.image * G[y0:y1]).sum().backward()
    gp_sum += ps.grad; gf_sum += fs.grad
  (full.image * G).sum().backward()
  assert torch.allclose(gp_sum, pf.grad, rtol=1e-4, atol=1e-4 * pf.grad.abs().max().item())
  assert torch.allclose(gf_sum, ff.grad, rtol=1e-4, atol=1e-5)
