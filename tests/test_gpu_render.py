"""-m gpu: render_gaussians end to end (project -> SH -> map -> rasterize, forward + backward) vs
the CPU oracle pipeline, in float64 (tight) and float32 (1e-4, BASELINE.json north_star)."""
import numpy as np
import pytest
import torch

from oracle import raster as orast, render as orender, projection as oproj, sh as osh
from taichi_splatting_amd import Gaussians3D, RasterConfig, render_gaussians
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def make_scene(n, size, seed, sh_degree=None, dtype=torch.float64):
  torch.manual_seed(seed)
  cam = random_camera(image_size=size)
  g = random_3d_gaussians(n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.1)
  if sh_degree is not None:
    g = g.replace(feature=(torch.rand(n, 3, (sh_degree + 1) ** 2) - 0.5) * 0.5)
  return g.to(dtype=dtype), cam.to(dtype=dtype)


def oracle_render_with_grads(g, cam, cfg, use_sh, G):
  leaves = [t.detach().clone().requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
  pos, ls, rot, al, feat = leaves
  points, depths, idx = oproj.apply(pos, ls, rot, al, cam.T_camera_world, cam.projection, cam.image_size,
                                    cam.depth_range, cfg.blur_cov, cfg.clamp_margin, cfg.alpha_threshold)
  if use_sh:
    feats = osh.evaluate_sh_at(feat, pos.detach(), idx, torch.inverse(cam.T_camera_world)[0:3, 3])
  else:
    feats = feat[idx]
  from oracle import mapper as omap
  ndc = oproj.ndc_depth(depths.detach(), *cam.depth_range)
  o2p, ranges, _ = omap.map_to_tiles(points.detach().numpy().astype(np.float32), ndc.numpy().astype(np.float32),
                                     cam.image_size, cfg.tile_size, cfg.alpha_threshold)
  o2p, ranges = torch.from_numpy(o2p), torch.from_numpy(ranges)
  image, alpha, _ = orast.forward(points.detach(), feats.detach(), ranges, o2p, cam.image_size, cfg)
  gp, gf, _ = orast.backward(points.detach(), feats.detach(), ranges, o2p, image, G, cam.image_size, cfg)
  torch.autograd.backward([points, feats], [gp, gf])
  return image, alpha, idx, [x.grad if x.grad is not None else torch.zeros_like(x) for x in leaves]


@pytest.mark.parametrize('use_sh,degree', [(False, None), (True, 0), (True, 3)])
def test_render_f64_matches_oracle(use_sh, degree):
  size = (200, 120)
  g, cam = make_scene(4000, size, seed=3 + (degree or 0), sh_degree=degree)
  cfg = RasterConfig()
  torch.manual_seed(1)
  G = torch.randn(size[1], size[0], 3, dtype=torch.float64)
  image_o, alpha_o, idx_o, grads_o = oracle_render_with_grads(g, cam, cfg, use_sh, G)

  gd = g.to(DEV).requires_grad_(True)
  r = render_gaussians(gd, cam.to(device=DEV), cfg, use_sh=use_sh)
  assert torch.equal(r.points.idx.cpu(), idx_o)
  assert torch.allclose(r.image.cpu(), image_o, atol=1e-9)
  assert torch.allclose(r.image_weight.cpu(), alpha_o, atol=1e-9)
  (r.image * G.to(DEV)).sum().backward()
  for name, got, want in zip(('position', 'log_scaling', 'rotation', 'alpha_logit', 'feature'),
                             (gd.position.grad, gd.log_scaling.grad, gd.rotation.grad, gd.alpha_logit.grad, gd.feature.grad),
                             grads_o):
    scale = max(1.0, want.abs().max().item())
    assert torch.allclose(got.cpu(), want, atol=1e-7 * scale, rtol=1e-6), (name, (got.cpu() - want).abs().max(), scale)


def oracle_forward(g, cam, cfg, use_sh, **kw):
  g64, cam64 = g.to(dtype=torch.float64), cam.to(dtype=torch.float64)
  return orender.render_forward(g64.position, g64.log_scaling, g64.rotation, g64.alpha_logit, g64.feature,
                                cam64.T_camera_world, cam64.projection, cam.image_size, cam.depth_range, cfg,
                                use_sh=use_sh, **kw)


def test_render_f32_within_1e4_config_b_shape():
  # BASELINE config B shape, down-scaled to what the oracle finishes in seconds: SH degree 0, forward only.
  # Gate-stable scene (see tests/test_gpu_raster.py): every pixel within 1e-4 of the float64 oracle.
  size = (256, 256)
  g, cam = make_scene(20000, size, seed=0, sh_degree=0, dtype=torch.float32)
  cfg = RasterConfig()
  o = oracle_forward(g, cam, cfg, True)
  margin = orast.gate_margin(o['points'].detach(), o['ranges'], o['o2p'], size, cfg)
  keep = torch.ones(20000, dtype=torch.bool)
  keep[o['indexes'][margin < 1e-4]] = False
  g = g[keep]
  o = oracle_forward(g, cam, cfg, True)
  r = render_gaussians(g.to(DEV), cam.to(device=DEV), cfg, use_sh=True)
  assert torch.equal(r.points.idx.cpu(), o['indexes'])
  assert (r.image.cpu().double() - o['image']).abs().max() < 1e-4
  assert (r.image_weight.cpu().double() - o['alpha']).abs().max() < 1e-4


def test_render_options_median_depth_and_visibility():
  size = (160, 96)
  g, cam = make_scene(3000, size, seed=9)
  cfg = RasterConfig(compute_visibility=True, compute_point_heuristic=True)
  gd = g.to(DEV).requires_grad_(True)
  r = render_gaussians(gd, cam.to(device=DEV), cfg, render_median_depth=True)
  assert r.median_depth_image.shape == (96, 160)
  assert r.points.visibility.shape == r.points.idx.shape
  r.image.sum().backward()
  assert r.points.prune_cost.shape == r.points.idx.shape and r.points.split_score.shape == r.points.idx.shape

  # against the oracle (float64): image, visibility, and the median depth = quantile render of the depths
  # (renderer.py:77-82: use_alpha_blending=False, saturate_threshold=median_threshold)
  o = oracle_forward(g, cam, cfg, False)
  assert torch.allclose(r.image.detach().cpu(), o['image'], atol=1e-9)
  assert torch.allclose(r.points.visibility.cpu(), o['visibility'], atol=1e-8)
  qcfg = orast.Cfg(use_alpha_blending=False, saturate_threshold=cfg.median_threshold)
  med, med_alpha, _ = orast.forward(o['points'].detach(), o['depths'].detach(), o['ranges'], o['o2p'], size, qcfg)
  mism = (r.median_depth_image.cpu() - med[..., 0]).abs() > 1e-9
  assert mism.float().mean() < 1e-3          # only pixels numerically on the quantile threshold may differ
  assert float(med[..., 0].max()) > 0
  cfg2 = RasterConfig()
  r2 = render_gaussians(g.to(DEV), cam.to(device=DEV), cfg2)
  with pytest.raises(AssertionError):
    _ = r2.points.visibility


def test_render_depth16_keys_and_strip_window():
  size = (200, 120)
  g, cam = make_scene(3000, size, seed=21)
  cfg = RasterConfig()
  full = render_gaussians(g.to(DEV), cam.to(device=DEV), cfg)
  d16 = render_gaussians(g.to(DEV), cam.to(device=DEV), cfg, use_depth16=True)
  # 16 bit keys (tile_mapper.py:49-66): the oracle pipeline with the same quantised sort keys gives the same image
  from oracle import mapper as omap, projection as oproj
  o = oracle_forward(g, cam, cfg, False)
  ndc = oproj.ndc_depth(o['depths'].detach(), *cam.depth_range)
  o2p16, ranges16, _ = omap.map_to_tiles(o['points'].detach().numpy().astype(np.float32), ndc.numpy().astype(np.float32),
                                         size, cfg.tile_size, cfg.alpha_threshold, use_depth16=True)
  img16, _, _ = orast.forward(o['points'].detach(), o['features'].detach(), torch.from_numpy(ranges16), torch.from_numpy(o2p16), size, cfg)
  assert torch.allclose(d16.image.cpu(), img16, atol=1e-9)
  assert torch.allclose(full.image.cpu(), o['image'], atol=1e-9)
  part = render_gaussians(g.to(DEV), cam.to(device=DEV), cfg, tile_rows=(2, 5))
  assert torch.equal(part.image[32:80], full.image[32:80])
  assert float(part.image[:32].abs().sum()) == 0 and float(part.image[80:].abs().sum()) == 0
