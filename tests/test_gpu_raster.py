"""-m gpu: the raster forward/backward kernels (through the C-ABI) vs the oracle.

Tolerances: float64 1e-9 (same algorithm, different summation order); float32 1e-4 absolute on
pixels and 1e-4 of the largest gradient on gradients (BASELINE.json north_star: "pixels/grads within
1e-4"), for EVERY pixel and EVERY splat, on GATE-STABLE scenes: the blend gate ``alpha > alpha_threshold``
(forward.py:99-101) is a discontinuity, a (pixel, splat) pair within float32 rounding of it may legitimately fall
on either side, and one flipped gate moves that pixel by ~alpha_threshold * |f| and the gradients of every splat
behind it.  ``gate_stable`` removes the splats that have such a pair (closer than 1e-4 relative, measured by the
float64 oracle: oracle.raster.gate_margin) before the comparison, so nothing is masked or averaged afterwards."""
from dataclasses import replace

import numpy as np
import pytest
import torch

from oracle import mapper as omap, raster as orast
from taichi_splatting_amd import RasterConfig, rasterize, rasterize_with_tiles, map_to_tiles
from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
from taichi_splatting_amd.testing import random_2d_gaussians

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def scene(n, size, seed, scale=1.0, alpha=(0.1, 0.9), channels=3, tile_size=16, depth_range=(0., 1.)):
  torch.manual_seed(seed)
  g = random_2d_gaussians(n, size, num_channels=channels, scale_factor=scale, alpha_range=alpha, depth_range=depth_range)
  p = project_gaussians2d(g)
  o2p, ranges, _ = omap.map_to_tiles(p.numpy(), g.depths.numpy(), size, tile_size)
  return p, g.feature, g.depths, torch.from_numpy(o2p), torch.from_numpy(ranges)


def cfg_for(tile_size, **kw):
  return RasterConfig(tile_size=tile_size, pixel_stride=(1, 1) if tile_size == 8 else (2, 2), **kw)


def gate_stable(g, size, cfg, rel_margin=1e-4):
  """The 2D gaussians of ``g`` that have no (pixel, splat) pair within ``rel_margin`` of the blend gate."""
  p = project_gaussians2d(g)
  o2p, ranges = map_to_tiles(p.to(DEV), g.depths.to(DEV), size, cfg)
  margin = orast.gate_margin(p.double().cpu(), ranges.cpu(), o2p.cpu(), size, cfg)
  keep = margin > rel_margin
  assert keep.float().mean() > 0.7
  return g[keep.to(g.position.device)]


def assert_within(got, want, tol, what):
  scale = want.abs().max().item()
  err = (got.cpu().double() - want.cpu().double()).abs().max().item()
  assert err < tol * max(scale, 1e-30), (what, err, scale)


@pytest.mark.parametrize('tile_size', [8, 16, 32])
@pytest.mark.parametrize('antialias', [False, True])
def test_forward_backward_f64(tile_size, antialias):
  size = (150, 100)    # not a multiple of any tile size: exercises out-of-bounds pixels
  cfg = cfg_for(tile_size, antialias=antialias, compute_visibility=True, compute_point_heuristic=True)
  p, f, d, o2p, ranges = scene(3000, size, seed=tile_size, scale=1.5, tile_size=tile_size)
  p, f = p.double(), f.double()
  img_o, a_o, vis_o = orast.forward(p, f, ranges, o2p, size, cfg)

  pg, fg = p.to(DEV).requires_grad_(True), f.to(DEV).requires_grad_(True)
  out = rasterize_with_tiles(pg, fg, o2p.to(DEV), ranges.to(DEV).view(-1, 2), size, cfg)
  assert out.image.shape == (100, 150, 3) and out.image_weight.shape == (100, 150)
  assert torch.allclose(out.image.cpu(), img_o, atol=1e-9)
  assert torch.allclose(out.image_weight.cpu(), a_o, atol=1e-9)
  assert torch.allclose(out.visibility.cpu(), vis_o, atol=1e-8)

  torch.manual_seed(0)
  G = torch.randn_like(img_o)
  gp_o, gf_o, h_o = orast.backward(p, f, ranges, o2p, img_o, G, size, cfg)
  (out.image * G.to(DEV)).sum().backward()
  scale = max(1.0, gp_o.abs().max().item())
  assert torch.allclose(pg.grad.cpu(), gp_o, atol=1e-8 * scale, rtol=1e-7)
  assert torch.allclose(fg.grad.cpu(), gf_o, atol=1e-9, rtol=1e-7)
  assert torch.allclose(out.point_heuristic.cpu(), h_o, atol=1e-7 * max(1.0, h_o.abs().max().item()), rtol=1e-6)


@pytest.mark.parametrize('tile_size', [8, 16, 32])
def test_forward_backward_f32_config_a(tile_size):
  # BASELINE config A shape: 10k random 2D gaussians, 256x256
  size = (256, 256)
  cfg = cfg_for(tile_size)
  torch.manual_seed(0)
  g = gate_stable(random_2d_gaussians(10000, size), size, cfg)
  p, f = project_gaussians2d(g), g.feature
  o2p, ranges, _ = omap.map_to_tiles(p.numpy(), g.depths.numpy(), size, tile_size)
  o2p, ranges = torch.from_numpy(o2p), torch.from_numpy(ranges)
  img_o, a_o, _, border = orast.forward(p.double(), f.double(), ranges, o2p, size, cfg, return_borderline=True)
  assert not bool(border.any())
  pg, fg = p.to(DEV).requires_grad_(True), f.to(DEV).requires_grad_(True)
  out = rasterize_with_tiles(pg, fg, o2p.to(DEV), ranges.to(DEV).view(-1, 2), size, cfg)
  assert (out.image.cpu().double() - img_o).abs().max() < 1e-4              # every pixel
  assert (out.image_weight.cpu().double() - a_o).abs().max() < 1e-4

  G = torch.ones_like(img_o)
  gp_o, gf_o, _ = orast.backward(p.double(), f.double(), ranges, o2p, img_o, G, size, cfg)
  out.image.sum().backward()
  assert_within(pg.grad, gp_o, 1e-4, 'd gaussians2d')                        # every splat, 1e-4 of the largest gradient
  assert_within(fg.grad, gf_o, 1e-4, 'd features')


@pytest.mark.parametrize('antialias', [False, True])
def test_gradcheck_single_tile(antialias):
  # the reference's own rasterizer test: tests/test_rasterizer.py:30-90 (f64, eps 1e-6, 8x8 image,
  # one tile, 1..49 gaussians, 1..3 channels, overlap_to_point = arange, tile_ranges = [[0, n]])
  cfg = RasterConfig(tile_size=8, pixel_stride=(1, 1), antialias=antialias, use_alpha_blending=True)
  torch.manual_seed(0)
  seeds = torch.randint(0, 1000, (100,))          # the reference's count (tests/test_rasterizer.py:60)
  for seed in seeds:
    torch.random.manual_seed(int(seed))
    n = int(torch.randint(1, 50, (1,)))
    channels = int(torch.randint(1, 4, (1,)))
    g = random_2d_gaussians(n, (8, 8), num_channels=channels, scale_factor=1.0, alpha_range=(0.2, 0.8))
    g2 = project_gaussians2d(g).to(device=DEV, dtype=torch.float64)
    colors = g.feature.to(device=DEV, dtype=torch.float64)
    o2p = torch.arange(0, n, device=DEV, dtype=torch.int32)
    ranges = torch.tensor([[0, n]], device=DEV, dtype=torch.int32)

    def render(mean, axis, sigma, alpha, colors):
      packed = torch.cat([mean, axis, sigma, alpha], dim=-1)
      return rasterize_with_tiles(packed, colors, overlap_to_point=o2p, tile_overlap_ranges=ranges,
                                  image_size=(8, 8), config=cfg).image
    inputs = (g2[:, 0:2].clone().requires_grad_(True), g2[:, 2:4].clone().requires_grad_(True),
              g2[:, 4:6].clone().requires_grad_(True), g2[:, 6:7].clone().requires_grad_(True),
              colors.clone().requires_grad_(True))
    torch.autograd.gradcheck(render, inputs, eps=1e-6, check_grad_dtypes=True, check_undefined_grad=True,
                             nondet_tol=1e-10)


def test_visibility_identity():
  # tests/test_visibility.py:34-64: with dL/dimage = 1, feature.grad[:, 0] == visibility (f64,
  # 320x200, default tile 16 / stride 2x2), through map_to_tiles + forward + backward
  np.random.seed(0)
  torch.manual_seed(0)
  size = (320, 200)
  cfg = RasterConfig(compute_visibility=True, compute_point_heuristic=True)
  for i in range(100):                            # the reference's count (tests/test_visibility.py:42)
    n = np.random.randint(1, 10000)
    g = random_2d_gaussians(n, size, scale_factor=0.2, alpha_range=(0.2, 1.0)).to(DEV).to(dtype=torch.float64)
    g.feature.requires_grad_(True)
    raster = rasterize(gaussians2d=project_gaussians2d(g), depth=torch.clamp(g.depths, 0, 1).to(torch.float32),
                       features=g.feature, image_size=size, config=cfg)
    raster.image.sum().backward()
    assert torch.allclose(g.feature.grad[:, 0], raster.visibility)
    assert raster.point_heuristic.shape == (n, 2)


@pytest.mark.parametrize('channels', [1, 2, 4, 5, 9])
def test_feature_widths(channels):
  size = (96, 64)
  cfg = RasterConfig()
  p, f, d, o2p, ranges = scene(1500, size, seed=channels, channels=channels)
  p, f = p.double(), f.double()
  img_o, a_o, _ = orast.forward(p, f, ranges, o2p, size, cfg)
  G = torch.randn_like(img_o)
  gp_o, gf_o, _ = orast.backward(p, f, ranges, o2p, img_o, G, size, cfg)
  pg, fg = p.to(DEV).requires_grad_(True), f.to(DEV).requires_grad_(True)
  out = rasterize_with_tiles(pg, fg, o2p.to(DEV), ranges.to(DEV).view(-1, 2), size, cfg)
  assert torch.allclose(out.image.cpu(), img_o, atol=1e-9)
  (out.image * G.to(DEV)).sum().backward()
  assert torch.allclose(pg.grad.cpu(), gp_o, atol=1e-8 * max(1, gp_o.abs().max().item()))
  assert torch.allclose(fg.grad.cpu(), gf_o, atol=1e-9)


@pytest.mark.parametrize('channels,dtype', [(5, torch.float64), (9, torch.float64), (16, torch.float64), (7, torch.float32)])
def test_point_heuristics_with_wide_features(channels, dtype):
  # prune_cost / split_score square and take |.| of dL/dalpha summed over ALL channels (backward.py:171-194):
  # with more than 4 channels they must not be assembled from channel chunks
  size = (96, 64)
  cfg = RasterConfig(compute_point_heuristic=True)
  p, f, d, o2p, ranges = scene(1500, size, seed=channels, channels=channels)
  p, f = p.double(), f.double()
  img_o, a_o, _ = orast.forward(p, f, ranges, o2p, size, cfg)
  torch.manual_seed(3)
  G = torch.randn_like(img_o)
  gp_o, gf_o, h_o = orast.backward(p, f, ranges, o2p, img_o, G, size, cfg)
  pg, fg = p.to(DEV, dtype).requires_grad_(True), f.to(DEV, dtype).requires_grad_(True)
  out = rasterize_with_tiles(pg, fg, o2p.to(DEV), ranges.to(DEV).view(-1, 2), size, cfg)
  (out.image * G.to(DEV, dtype)).sum().backward()
  tol = 1e-8 if dtype == torch.float64 else 1e-4
  for got, want in ((pg.grad, gp_o), (fg.grad, gf_o), (out.point_heuristic, h_o)):
    scale = max(1.0, want.abs().max().item())
    assert (got.cpu().double() - want).abs().max() < tol * scale, ((got.cpu().double() - want).abs().max(), scale)
  with pytest.raises(NotImplementedError):
    wide = torch.zeros((p.shape[0], 17), device=DEV, dtype=dtype)
    rasterize_with_tiles(pg.detach(), wide, o2p.to(DEV), ranges.to(DEV).view(-1, 2), size, cfg)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32])
def test_quantile_mode_visibility_vs_oracle(dtype):
  """use_alpha_blending=False + compute_visibility (forward.py:102-126): the weight keeps accumulating past the
  quantile and `visibility` sums it for every gated splat.  Against the oracle's restatement (untruncated sums; the
  reference itself stops a warp once its 32 pixels are saturated — its values are <= these), through
  rasterize_with_tiles AND the frame executor (rasterize)."""
  from taichi_splatting_amd import rasterize
  size = (96, 64)
  cfg = RasterConfig(use_alpha_blending=False, compute_visibility=True, saturate_threshold=0.3)
  p, f, d, o2p, ranges = scene(1500, size, seed=3, scale=2.0, alpha=(0.3, 0.9), channels=1)
  p64, f64 = p.double(), f.double()
  img_o, a_o, vis_o = orast.forward(p64, f64, ranges, o2p, size, cfg)
  assert float(vis_o.max()) > 1.0                       # sums over many pixels, past the quantile
  tol = 1e-9 if dtype == torch.float64 else 2e-5
  out = rasterize_with_tiles(p.to(DEV, dtype), f.to(DEV, dtype), o2p.to(DEV), ranges.to(DEV).view(-1, 2), size, cfg)
  assert torch.equal(out.image_weight.cpu().double(), a_o)
  assert (out.visibility.cpu().double() - vis_o).abs().max().item() <= tol * float(vis_o.max())
  out2 = rasterize(p.to(DEV, dtype), d.to(DEV, dtype), f.to(DEV, dtype), size, cfg)
  assert (out2.visibility.cpu().double() - vis_o).abs().max().item() <= tol * float(vis_o.max())
  if dtype == torch.float64:
    assert ((out.image.cpu() - img_o).abs().max(-1).values > 1e-12).float().mean() < 1e-3      # (quantile ties: see below)
  # the blending weights are the same numbers: with blending on, the same splats get the same visibility
  cfg_b = RasterConfig(use_alpha_blending=True, compute_visibility=True)
  out_b = rasterize_with_tiles(p.to(DEV, dtype), f.to(DEV, dtype), o2p.to(DEV), ranges.to(DEV).view(-1, 2), size, cfg_b)
  assert (out_b.visibility.cpu().double() - vis_o).abs().max().item() <= max(tol, 2e-5) * float(vis_o.max())


def test_quantile_render_no_blending():
  # use_alpha_blending=False + saturate_threshold = median_threshold (renderer.py:77-82)
  size = (128, 96)
  cfg = RasterConfig(use_alpha_blending=False, saturate_threshold=0.25)
  p, f, d, o2p, ranges = scene(4000, size, seed=7, scale=2.0, alpha=(0.3, 0.9), channels=1)
  p, f = p.double(), f.double()
  img_o, a_o, _ = orast.forward(p, f, ranges, o2p, size, cfg)
  out = rasterize_with_tiles(p.to(DEV), f.to(DEV), o2p.to(DEV), ranges.to(DEV).view(-1, 2), size, cfg)
  assert torch.equal(out.image_weight.cpu(), a_o)
  got = out.image.cpu()
  mism = (got - img_o).abs().max(-1).values > 1e-12
  assert mism.float().mean() < 1e-3
  # ... and each of them must BE a pixel numerically on the quantile threshold: walking the pixel's list in float64,
  # the accumulated weight reaches 1 - saturate_threshold within 1e-9 at the splat one of the two results took, and
  # the kernel's value is the feature of that splat or of its neighbour in the list (nothing else may differ, and by
  # no more than the two candidates differ)
  ts = cfg.tile_size
  tiles_wide = (size[0] + ts - 1) // ts
  for y, x in torch.nonzero(mism).tolist():
    tile = (y // ts) * tiles_wide + x // ts
    start, end = [int(v) for v in ranges.view(-1, 2)[tile]]
    ids = o2p[start:end].long()
    gp = p[ids]
    pix = torch.tensor([[x + 0.5, y + 0.5]], dtype=torch.float64)
    a = torch.clamp_max(gp[:, 6] * orast.pdf(pix, gp, cfg.antialias)[0], cfg.clamp_max_alpha)
    a = torch.where(a > cfg.alpha_threshold, a, torch.zeros_like(a))
    T = torch.cumprod(torch.cat([torch.ones(1, dtype=torch.float64), 1 - a[:-1]]), dim=0)
    W = torch.cumsum(a * T, dim=0)
    k = int(torch.nonzero(W >= 1 - cfg.saturate_threshold)[0])
    near = min(abs(float(W[j]) - (1 - cfg.saturate_threshold)) for j in (k - 1, k) if j >= 0)
    assert near < 1e-9, ((x, y), near)
    candidates = f[ids[max(k - 1, 0):k + 2]]
    assert bool(((candidates - got[y, x]).abs().max(dim=-1).values < 1e-12).any()), (x, y)


def test_many_batches_per_tile_and_early_exit():
  # > 256 splats per tile (several LDS batches) with opaque splats: exercises the corrected
  # in-group loop bound and the backward saturation early-out
  size = (64, 64)
  cfg = RasterConfig()
  p, f, d, o2p, ranges = scene(6000, size, seed=11, scale=6.0, alpha=(0.6, 1.0))
  assert int((ranges[..., 1] - ranges[..., 0]).max()) > 600
  p, f = p.double(), f.double()
  img_o, a_o, _ = orast.forward(p, f, ranges, o2p, size, cfg)
  G = torch.randn_like(img_o)
  gp_o, gf_o, _ = orast.backward(p, f, ranges, o2p, img_o, G, size, cfg)
  pg, fg = p.to(DEV).requires_grad_(True), f.to(DEV).requires_grad_(True)
  out = rasterize_with_tiles(pg, fg, o2p.to(DEV), ranges.to(DEV).view(-1, 2), size, cfg)
  assert torch.allclose(out.image.cpu(), img_o, atol=1e-9)
  (out.image * G.to(DEV)).sum().backward()
  assert torch.allclose(pg.grad.cpu(), gp_o, atol=1e-7 * max(1, gp_o.abs().max().item()))
  assert torch.allclose(fg.grad.cpu(), gf_o, atol=1e-8)


def test_empty_inputs_and_requires_grad_subsets():
  cfg = RasterConfig()
  size = (40, 24)
  out = rasterize(torch.zeros((0, 7), device=DEV), torch.zeros((0, 1), device=DEV), torch.zeros((0, 3), device=DEV), size, cfg)
  assert out.image.shape == (24, 40, 3) and float(out.image.abs().sum()) == 0 and float(out.image_weight.abs().sum()) == 0
  p, f, d, o2p, ranges = scene(300, size, seed=2)
  pg = p.to(DEV).requires_grad_(True)
  out = rasterize_with_tiles(pg, f.to(DEV), o2p.to(DEV), ranges.to(DEV).view(-1, 2), size, cfg)
  out.image.sum().backward()
  assert pg.grad is not None and pg.grad.abs().sum() > 0
  fg = f.to(DEV).requires_grad_(True)
  out = rasterize_with_tiles(p.to(DEV), fg, o2p.to(DEV), ranges.to(DEV).view(-1, 2), size, cfg)
  out.image.sum().backward()
  assert fg.grad is not None and fg.grad.abs().sum() > 0


@pytest.mark.parametrize('tile_size,n,size,scale,alpha',
                         [(t, *case) for t in (8, 16, 32) for case in ((20000, (333, 200), 1.0, (0.1, 0.9)), (8000, (96, 64), 6.0, (0.6, 1.0)))]
                         + [(16, 70000, (640, 480), 1.0, (0.1, 0.9))])      # the large case at the product tile size only (suite time)
def test_f32_product_kernels_match_f64_generic_kernels(tile_size, n, size, scale, alpha):
  # the hand-tuned float/RGB kernels (raster_fast.hip forward, raster_bwd_scan.hip backward) against the generic
  # f64 instantiation (itself checked against the oracle above), incl. many LDS batches per tile, image sizes
  # that are not tile multiples, point heuristics, visibility — on a gate-stable scene, strictly
  torch.manual_seed(n + tile_size)
  cfg = cfg_for(tile_size, compute_point_heuristic=True, compute_visibility=True)
  g = gate_stable(random_2d_gaussians(n, size, scale_factor=scale, alpha_range=alpha), size, cfg).to(DEV)
  p32 = project_gaussians2d(g)
  o2p, ranges = map_to_tiles(p32, g.depths, size, cfg)
  ranges = ranges.view(-1, 2)
  torch.manual_seed(1)
  G = torch.randn(size[1], size[0], 3, device=DEV)
  res = {}
  for dtype in (torch.float64, torch.float32):
    p = p32.to(dtype).clone().requires_grad_(True)
    f = g.feature.to(dtype).clone().requires_grad_(True)
    out = rasterize_with_tiles(p, f, o2p, ranges, size, cfg)
    (out.image * G.to(dtype)).sum().backward()
    res[dtype] = (out.image.detach().double(), out.image_weight.detach().double(), p.grad.double(), f.grad.double(),
                  out.point_heuristic.double(), out.visibility.double())
  img64, a64, gp64, gf64, h64, v64 = res[torch.float64]
  img32, a32, gp32, gf32, h32, v32 = res[torch.float32]
  assert (img32 - img64).abs().max() < 1e-4 and (a32 - a64).abs().max() < 1e-4
  assert v64.sum() > 0
  for what, got, want in (('d gaussians2d', gp32, gp64), ('d features', gf32, gf64), ('heuristics', h32, h64), ('visibility', v32, v64)):
    assert_within(got, want, 1e-4, what)


@pytest.mark.parametrize('heuristics', [False, True])
@pytest.mark.parametrize('tile_size', [8, 16])
def test_sub_pixel_splats_keep_their_gradient_digits(tile_size, heuristics):
  # splats far narrower than a pixel (the 2D operators accept any sigma): the backward's grid form of the moment sums
  # would cancel catastrophically there (basis entries of tens per pixel) — the kernel switches those chunks to the
  # per-pixel form.  Mixed with ordinary splats so that both forms run inside one tile.
  torch.manual_seed(77 + tile_size)
  size, n = (96, 80), 6000
  cfg = cfg_for(tile_size, compute_point_heuristic=heuristics)
  g = random_2d_gaussians(n, size, scale_factor=1.0, alpha_range=(0.3, 0.95))
  tiny = torch.rand(n) < 0.5
  g.log_scaling[tiny] = torch.log(0.01 + 0.15 * torch.rand(int(tiny.sum()), 2))          # sigma 0.01 .. 0.16 px
  g.position[tiny] = (g.position[tiny].floor() + 0.5 + 0.02 * torch.randn(int(tiny.sum()), 2)).clamp(0.5, min(size) - 0.5)
  g = gate_stable(g, size, cfg).to(DEV)
  p32 = project_gaussians2d(g)
  assert (p32[:, 4:6].min(dim=1).values < 0.2).float().mean() > 0.3
  o2p, ranges = map_to_tiles(p32, g.depths, size, cfg)
  ranges = ranges.view(-1, 2)
  torch.manual_seed(2)
  G = torch.randn(size[1], size[0], 3, device=DEV)
  res = {}
  for dtype in (torch.float64, torch.float32):
    p = p32.to(dtype).clone().requires_grad_(True)
    f = g.feature.to(dtype).clone().requires_grad_(True)
    out = rasterize_with_tiles(p, f, o2p, ranges, size, cfg)
    (out.image * G.to(dtype)).sum().backward()
    res[dtype] = (p.grad.double(), f.grad.double(), out.point_heuristic.double())
  gp64, gf64, h64 = res[torch.float64]
  gp32, gf32, h32 = res[torch.float32]
  assert gp64[:, 4:6].abs().max() > 0                       # the tiny splats do receive sigma gradients
  # column by column: a sub-pixel splat's d sigma is orders of magnitude above its d mean
  for k, name in enumerate(('mean.x', 'mean.y', 'axis.x', 'axis.y', 'sigma.x', 'sigma.y', 'alpha')):
    assert_within(gp32[:, k], gp64[:, k], 1e-4, f"d gaussians2d[{name}]")
  assert_within(gf32, gf64, 1e-4, 'd features')
  if heuristics:
    assert_within(h32, h64, 1e-4, 'heuristics')


@pytest.mark.parametrize('tile_size,n,size,scale,alpha', [(16, 20000, (333, 200), 1.0, (0.1, 0.9)), (8, 6000, (96, 64), 5.0, (0.5, 1.0)),
                                                          (32, 100000, (640, 480), 1.5, (0.02, 0.9))])
def test_f32_antialias_matches_f64(tile_size, n, size, scale, alpha):
  # antialiased pdf: the float kernels use v_exp_f32 / v_rcp_f32 and the contribution-rectangle cull; the f64
  # instantiation (exact formulation, checked against the oracle above) is the reference; gate-stable scene
  torch.manual_seed(n)
  cfg = cfg_for(tile_size, antialias=True, compute_point_heuristic=True, compute_visibility=True)
  g = gate_stable(random_2d_gaussians(n, size, scale_factor=scale, alpha_range=alpha), size, cfg).to(DEV)
  p32 = project_gaussians2d(g)
  o2p, ranges = map_to_tiles(p32, g.depths, size, cfg)
  ranges = ranges.view(-1, 2)
  torch.manual_seed(1)
  G = torch.randn(size[1], size[0], 3, device=DEV)
  res = {}
  for dtype in (torch.float64, torch.float32):
    p = p32.to(dtype).clone().requires_grad_(True)
    f = g.feature.to(dtype).clone().requires_grad_(True)
    out = rasterize_with_tiles(p, f, o2p, ranges, size, cfg)
    (out.image * G.to(dtype)).sum().backward()
    res[dtype] = (out.image.detach().double(), p.grad.double(), f.grad.double(), out.point_heuristic.double(),
                  out.visibility.double())
  assert (res[torch.float32][0] - res[torch.float64][0]).abs().max() < 1e-4
  assert res[torch.float64][1].abs().sum() > 0
  for what, got, want in zip(('d gaussians2d', 'd features', 'heuristics', 'visibility'), res[torch.float32][1:], res[torch.float64][1:]):
    assert_within(got, want, 1e-4, what)
