"""Pins for the rasterizer / tile-mapper oracle, for which the reference holds no golden data
(SURVEY.md 8c): hand-computable known answers, gradcheck (the reference's own rasterizer test,
tests/test_rasterizer.py:62-90), literal-backward vs autograd, the visibility identity
(tests/test_visibility.py:34-64) and brute-force mapper invariants."""
import math

import numpy as np
import pytest
import torch

from oracle import mapper as omap, raster as orast
from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
from taichi_splatting_amd.testing import random_2d_gaussians


def one_tile(n, ts=8):
  return torch.arange(n, dtype=torch.int32), torch.tensor([[0, n]], dtype=torch.int32)


def test_kat_single_isotropic_splat():
  # one isotropic splat centred on pixel centre (3.5, 4.5): image = min(alpha, .99) * f there,
  # alpha * exp(-1/(2 sigma^2)) * f at the 4-neighbours
  sigma, alpha = 1.5, 0.6
  p = torch.tensor([[3.5, 4.5, 1.0, 0.0, sigma, sigma, alpha]], dtype=torch.float64)
  f = torch.tensor([[0.2, 0.5, 1.0]], dtype=torch.float64)
  o2p, ranges = one_tile(1)
  img, a, vis = orast.forward(p, f, ranges, o2p, (8, 8), orast.Cfg(tile_size=8))
  assert torch.allclose(img[4, 3], alpha * f[0])
  nb = alpha * math.exp(-1 / (2 * sigma ** 2))
  for (y, x) in ((4, 2), (4, 4), (3, 3), (5, 3)):
    assert torch.allclose(img[y, x], nb * f[0])
    assert abs(a[y, x].item() - nb) < 1e-14
  assert abs(vis[0].item() - a.sum().item()) < 1e-12


def test_kat_two_stacked_splats_and_clamp_threshold():
  # C = f1 a1 + f2 a2 (1 - a1); a1 clamps to 0.99; contributions below 1/255 are dropped
  p = torch.tensor([[2.5, 2.5, 1.0, 0.0, 50.0, 50.0, 1.0],
                    [2.5, 2.5, 0.0, 1.0, 50.0, 50.0, 0.5],
                    [2.5, 2.5, 1.0, 0.0, 50.0, 50.0, 0.003]], dtype=torch.float64)
  f = torch.tensor([[1.0], [10.0], [1000.0]], dtype=torch.float64)
  o2p, ranges = one_tile(3)
  img, a, _ = orast.forward(p, f, ranges, o2p, (8, 8), orast.Cfg(tile_size=8))
  a1 = 0.99
  a2 = 0.5 * math.exp(0.0)
  assert abs(img[2, 2, 0].item() - (1.0 * a1 + 10.0 * a2 * (1 - a1))) < 1e-12
  assert abs(a[2, 2].item() - (a1 + a2 * (1 - a1))) < 1e-12


def _random_tile_inputs(seed, antialias=False):
  torch.manual_seed(seed)
  n = int(torch.randint(1, 50, (1,)))
  ch = int(torch.randint(1, 4, (1,)))
  g = random_2d_gaussians(n, (8, 8), num_channels=ch, scale_factor=1.0, alpha_range=(0.2, 0.8))
  return project_gaussians2d(g).double(), g.feature.double(), n


@pytest.mark.parametrize('antialias', [False, True])
def test_oracle_gradcheck_single_tile(antialias):
  # protocol of the reference's tests/test_rasterizer.py:30-90 (8x8 image, one tile, 1-49 splats)
  cfg = orast.Cfg(tile_size=8, antialias=antialias)
  for seed in range(3):
    p, f, n = _random_tile_inputs(seed, antialias)
    o2p, ranges = one_tile(n)

    def render(p_, f_):
      return orast.rasterize_autograd(p_, f_, ranges, o2p, (8, 8), cfg)
    torch.autograd.gradcheck(render, (p.clone().requires_grad_(True), f.clone().requires_grad_(True)),
                             eps=1e-6, atol=1e-5, rtol=1e-3)


@pytest.mark.parametrize('antialias', [False, True])
def test_literal_backward_matches_autograd(antialias):
  cfg = orast.Cfg(tile_size=8, antialias=antialias)
  for seed in range(10):
    p, f, n = _random_tile_inputs(seed)
    o2p, ranges = one_tile(n)
    img, _, _ = orast.forward(p, f, ranges, o2p, (8, 8), cfg)
    G = torch.randn_like(img)
    gp, gf, _ = orast.backward(p, f, ranges, o2p, img, G, (8, 8), cfg)
    pp, ff = p.clone().requires_grad_(True), f.clone().requires_grad_(True)
    img2 = orast.rasterize_autograd(pp, ff, ranges, o2p, (8, 8), cfg)
    assert torch.allclose(img2, img, atol=1e-13)
    (img2 * G).sum().backward()
    # the literal backward skips saturated pixels (W >= 0.9999): allow that much slack
    assert torch.allclose(gp, pp.grad, atol=2e-3 * max(1.0, pp.grad.abs().max().item()), rtol=1e-3)
    assert torch.allclose(gf, ff.grad, atol=2e-4, rtol=1e-3)


def test_visibility_identity_and_mapper_invariants():
  torch.manual_seed(3)
  size = (320, 200)
  g = random_2d_gaussians(3000, size, scale_factor=0.2, alpha_range=(0.2, 1.0))
  p = project_gaussians2d(g).double()
  depth = torch.clamp(g.depths, 0, 1).float()
  o2p, ranges, counts = omap.map_to_tiles(p.numpy(), depth.numpy(), size, 16)
  K = o2p.shape[0]
  assert counts.sum() == K
  flat = ranges.reshape(-1, 2)
  nonempty = flat[flat[:, 1] > flat[:, 0]]
  # ranges partition [0, K)
  assert nonempty[:, 0].min() == 0 and nonempty[:, 1].max() == K
  order = np.argsort(nonempty[:, 0])
  assert np.all(nonempty[order][1:, 0] == nonempty[order][:-1, 1])
  # depth non-decreasing inside each range, ties by ascending point index
  d = depth.numpy().reshape(-1)
  for s, e in nonempty:
    dd, ii = d[o2p[s:e]], o2p[s:e]
    assert np.all(np.diff(dd) >= 0)
    ties = np.diff(dd) == 0
    assert np.all(np.diff(ii)[ties] > 0)

  cfg = orast.Cfg(compute_visibility=True)
  o2p_t, ranges_t = torch.from_numpy(o2p), torch.from_numpy(ranges)
  f = g.feature.double()
  img, a, vis = orast.forward(p, f, ranges_t, o2p_t, size, cfg)
  gp, gf, _ = orast.backward(p, f, ranges_t, o2p_t, img, torch.ones_like(img), size, cfg)
  # with dL/dimage = 1 the feature gradient is the visibility (tests/test_visibility.py:58-64),
  # up to the pixels the backward pass skips once saturated
  assert torch.allclose(gf[:, 0], vis, rtol=1e-5, atol=2e-3)


def test_mapper_membership_is_bruteforce_sat():
  # every (point, tile) membership agrees with an independent float64 SAT evaluation, except
  # pairs that are numerically on the decision boundary
  torch.manual_seed(5)
  size = (100, 70)
  g = random_2d_gaussians(200, size, scale_factor=1.5, alpha_range=(0.05, 1.0))
  p = project_gaussians2d(g).numpy().astype(np.float32)
  thr = 1 / 255.
  pid, tx, ty = omap.overlaps(p, size, 16, thr)
  got = set(zip(pid.tolist(), tx.tolist(), ty.tolist()))
  w_pad, h_pad = omap.pad_to_tile(size, 16)
  want = set()
  p64 = p.astype(np.float64)
  for i in range(p.shape[0]):
    mean, a1, sig, alpha = p64[i, 0:2], p64[i, 2:4], p64[i, 4:6], p64[i, 6]
    if alpha <= thr:
      continue
    gs = math.sqrt(2 * math.log(alpha / thr))
    sc = sig * gs
    a2 = np.array([-a1[1], a1[0]])
    ext = np.sqrt((a1 * sc[0]) ** 2 + (a2 * sc[1]) ** 2)
    lo = np.maximum(np.floor((mean - ext) / 16), 0).astype(int)
    hi = np.ceil((mean + ext) / 16).astype(int)
    hi = np.minimum(np.maximum(hi, lo + 1), [w_pad // 16, h_pad // 16])
    for x in range(lo[0], hi[0]):
      for y in range(lo[1], hi[1]):
        corners = np.array([[x, y], [x + 1, y], [x + 1, y + 1], [x, y + 1]], dtype=np.float64) * 16 - mean
        ok = True
        for ax, s in ((a1, sc[0]), (a2, sc[1])):
          pr = corners @ ax / s
          if pr.min() > 1 or pr.max() < -1:
            ok = False
        if ok:
          want.add((i, x, y))
  border = omap.borderline_pairs(p, size, 16, thr)
  assert (got ^ want) <= border, sorted(got ^ want)[:5]


# ---- second opinion: literal scalar loops (oracle/raster_scalar.py) vs the vectorised oracle ------------------

def test_scalar_loop_oracle_agrees_on_multi_group_tiles():
  # two 16x16 tiles with > 256 splats each (several groups of the reference's staging), depth ties, out-of-image
  # columns (width 28 is not a tile multiple); forward image / alpha / visibility and all gradients
  from oracle import raster_scalar as osc
  torch.manual_seed(5)
  size = (28, 16)
  g = random_2d_gaussians(800, size, scale_factor=2.5, alpha_range=(0.05, 0.9))
  p, f = project_gaussians2d(g).double(), g.feature.double()
  o2p, ranges, _ = omap.map_to_tiles(p.numpy(), g.depths.numpy(), size, 16)
  assert int((ranges[..., 1] - ranges[..., 0]).max()) > 256
  o2p_t, ranges_t = torch.from_numpy(o2p), torch.from_numpy(ranges)
  cfg = orast.Cfg(compute_point_heuristic=True)
  img, alpha, vis = orast.forward(p, f, ranges_t, o2p_t, size, cfg)
  G = torch.randn_like(img)
  gp, gf, heur = orast.backward(p, f, ranges_t, o2p_t, img, G, size, cfg)

  rl = [tuple(int(v) for v in r) for r in ranges.reshape(-1, 2)]
  img_s, alpha_s, vis_s = osc.forward(p.tolist(), f.tolist(), rl, o2p.tolist(), size)
  assert torch.allclose(torch.tensor(img_s, dtype=torch.float64), img, atol=1e-12)
  assert torch.allclose(torch.tensor(alpha_s, dtype=torch.float64), alpha, atol=1e-12)
  assert torch.allclose(torch.tensor(vis_s, dtype=torch.float64), vis, atol=1e-11)
  gp_s, gf_s, heur_s = osc.backward(p.tolist(), f.tolist(), rl, o2p.tolist(), img.tolist(), G.tolist(), size)
  scale = float(gp.abs().max())
  assert torch.allclose(torch.tensor(gp_s, dtype=torch.float64), gp, atol=1e-11 * scale, rtol=1e-9)
  assert torch.allclose(torch.tensor(gf_s, dtype=torch.float64), gf, atol=1e-12, rtol=1e-9)
  assert torch.allclose(torch.tensor(heur_s, dtype=torch.float64), heur, atol=1e-10 * float(heur.abs().max()), rtol=1e-9)


def test_reference_tail_order_known_answers():
  # forward.py:86-89: group g visits min(group, count - g) entries; beyond the valid ones it re-reads what the
  # previous group left in shared memory
  assert orast.reference_tail_order(5, 256) == list(range(5))
  assert orast.reference_tail_order(256, 256) == list(range(256))
  order = orast.reference_tail_order(300, 256)
  assert order[:300] == list(range(300))
  assert order[300:] == list(range(44, 256))            # stale entries 44..255 of group 0, blended a second time
  assert len(order) == 256 + min(256, 300 - 1)
  order = orast.reference_tail_order(130, 64)           # backward block of 64 threads (tile 16, stride 2x2)
  assert order[:130] == list(range(130)) and order[130:] == list(range(64 + 2, 128))


def test_reference_loop_bound_emulation_changes_only_unsaturated_tails():
  torch.manual_seed(2)
  size = (16, 16)
  g = random_2d_gaussians(300, size, scale_factor=3.0, alpha_range=(0.02, 0.2))
  p, f = project_gaussians2d(g).double(), g.feature.double()
  o2p, ranges, _ = omap.map_to_tiles(p.numpy(), g.depths.numpy(), size, 16)
  o2p, ranges = torch.from_numpy(o2p), torch.from_numpy(ranges)
  assert int(ranges.reshape(-1, 2)[0, 1]) > 256
  cfg = orast.Cfg()
  img, alpha, _ = orast.forward(p, f, ranges, o2p, size, cfg)
  img_r, alpha_r, _ = orast.forward(p, f, ranges, o2p, size, cfg, emulate_reference_loop_bound=True)
  assert bool((alpha_r >= alpha - 1e-15).all()) and float((alpha_r - alpha).max()) > 1e-3   # re-blended tail adds weight
  # opaque front layer: nothing is left to re-blend, both orders give the same image
  p2 = p.clone(); p2[:, 6] = 0.995; p2[:, 4:6] = 30.0
  img2, alpha2, _ = orast.forward(p2, f, ranges, o2p, size, cfg)
  img2_r, alpha2_r, _ = orast.forward(p2, f, ranges, o2p, size, cfg, emulate_reference_loop_bound=True)
  assert torch.allclose(img2, img2_r, atol=1e-9) and torch.allclose(alpha2, alpha2_r, atol=1e-9)
  # <= 256 splats per tile: the defect cannot trigger
  few = o2p[:200]
  r_few = torch.tensor([[0, 200]], dtype=torch.int32)
  a, _, _ = orast.forward(p, f, r_few, few, size, cfg)
  b, _, _ = orast.forward(p, f, r_few, few, size, cfg, emulate_reference_loop_bound=True)
  assert torch.equal(a, b)


def test_gate_margin_matches_borderline_pixels():
  torch.manual_seed(9)
  size = (48, 32)
  g = random_2d_gaussians(600, size, scale_factor=1.5)
  p, f = project_gaussians2d(g).double(), g.feature.double()
  o2p, ranges, _ = omap.map_to_tiles(p.numpy(), g.depths.numpy(), size, 16)
  o2p, ranges = torch.from_numpy(o2p), torch.from_numpy(ranges)
  cfg = orast.Cfg()
  margin = orast.gate_margin(p, ranges, o2p, size, cfg)
  listed = torch.zeros(600, dtype=torch.bool); listed[o2p.long()] = True
  assert bool(torch.isfinite(margin[listed]).all()) and bool(torch.isinf(margin[~listed]).all())
  # dropping every splat closer than 1e-3 to the gate leaves a scene without borderline pixels at 1e-6
  keep = margin > 1e-3
  assert 0.5 < keep.float().mean() < 1.0
  p2, f2, d2 = p[keep], f[keep], g.depths[keep]
  o2p2, ranges2, _ = omap.map_to_tiles(p2.numpy(), d2.numpy(), size, 16)
  _, _, _, border = orast.forward(p2, f2, torch.from_numpy(ranges2), torch.from_numpy(o2p2), size, cfg, return_borderline=True)
  assert not bool(border.any())


def test_non_blending_mode_accumulates_visibility():
  # forward.py:114-126: the blend weights keep flowing into `visibility` with use_alpha_blending=False
  torch.manual_seed(4)
  size = (16, 16)
  g = random_2d_gaussians(60, size, scale_factor=2.0, alpha_range=(0.1, 0.5))
  p, f = project_gaussians2d(g).double(), g.feature.double()
  o2p, ranges, _ = omap.map_to_tiles(p.numpy(), g.depths.numpy(), size, 16)
  o2p, ranges = torch.from_numpy(o2p), torch.from_numpy(ranges)
  _, _, vis_blend = orast.forward(p, f, ranges, o2p, size, orast.Cfg())
  _, _, vis_median = orast.forward(p, f, ranges, o2p, size, orast.Cfg(use_alpha_blending=False, saturate_threshold=0.25))
  assert float(vis_blend.sum()) > 0 and torch.allclose(vis_blend, vis_median)


def test_saturation_margin_known_answers():
  """oracle.raster.saturation_margin (round 6): per splat, how close its nearest gated (pixel, splat) pair sits to the
  backward's saturation test T_before <= 1 - saturate_threshold (backward.py:154,160), and on which side.  Four wide
  splats over one tile, alpha 0.9 each (flat over the 8 x 8 tile to ~1e-4): T in front of them = 1, 0.1, 0.01, 0.001 ->
  with saturate_threshold 0.99 (limit 0.01) the margins are 99, 9, ~0 (the gaussian's fall-off decides the side), 0.9."""
  n = 4
  p = torch.tensor([[4.0, 4.0, 1.0, 0.0, 400.0, 400.0, 0.9]] * n, dtype=torch.float64)
  o2p, ranges = one_tile(n)
  cfg = orast.Cfg(tile_size=8, saturate_threshold=0.99)
  margin, side = orast.saturation_margin(p, ranges, o2p, (8, 8), cfg)
  assert abs(margin[0].item() - 99.0) < 1e-2 and side[0].item() == 1
  assert abs(margin[1].item() - 9.0) < 1e-2 and side[1].item() == 1
  assert margin[2].item() < 1e-3                                      # T ~ 0.0100x: on the limit
  assert abs(margin[3].item() - 0.9) < 1e-2 and side[3].item() == -1   # T ~ 0.001: dropped by the backward
  # the backward really drops the fourth splat and keeps the second
  f = torch.rand(n, 3, dtype=torch.float64)
  img, _, _ = orast.forward(p, f, ranges, o2p, (8, 8), cfg)
  gp, gf, _ = orast.backward(p, f, ranges, o2p, img, torch.ones_like(img), (8, 8), cfg)
  assert float(gf[3].abs().max()) == 0.0 and float(gf[1].abs().max()) > 0
  # a splat below the blend gate everywhere has no gated pair: margin inf, side 0
  faint = torch.tensor([[4.0, 4.0, 1.0, 0.0, 2.0, 2.0, 0.001]], dtype=torch.float64)
  m, s = orast.saturation_margin(faint, torch.tensor([[0, 1]], dtype=torch.int32), torch.arange(1, dtype=torch.int32), (8, 8), cfg)
  assert math.isinf(m[0].item()) and s[0].item() == 0


def test_active_visibility_known_answers():
  """oracle.raster.active_visibility (round 6): forward visibility without the pairs the backward drops behind a pixel's
  saturation point.  The four flat splats of the test above (alpha 0.9, T in front 1, 0.1, 0.01, 0.001; 64 pixels; limit
  0.01): the forward sums 0.9 T over the 64 pixels for each of them, the backward visits the first two everywhere, the
  third only where the fall-off leaves T above the limit, the fourth nowhere.  Equal to the forward's where nothing
  saturates."""
  n = 4
  p = torch.tensor([[4.0, 4.0, 1.0, 0.0, 400.0, 400.0, 0.9]] * n, dtype=torch.float64)
  o2p, ranges = one_tile(n)
  cfg = orast.Cfg(tile_size=8, saturate_threshold=0.99)
  f = torch.rand(n, 3, dtype=torch.float64)
  _, _, vis = orast.forward(p, f, ranges, o2p, (8, 8), cfg)
  act = orast.active_visibility(p, ranges, o2p, (8, 8), cfg)
  for k, t in enumerate((1.0, 0.1, 0.01, 0.001)):
    assert abs(vis[k].item() - 64 * 0.9 * t) < 1e-2 * 64 * 0.9 * t
  assert torch.allclose(act[:2], vis[:2], rtol=1e-12)
  assert 0.0 <= act[2].item() <= vis[2].item() and act[3].item() == 0.0
  # the gradients of the colours are the same pairs' weights times dL/dC: with dL/dC = 1 per channel, d f = active visibility
  img, _, _ = orast.forward(p, f, ranges, o2p, (8, 8), cfg)
  _, gf, _ = orast.backward(p, f, ranges, o2p, img, torch.ones_like(img), (8, 8), cfg)
  assert torch.allclose(gf[:, 0], act, rtol=1e-12, atol=1e-15)
  # nothing saturates: the two visibilities are one
  cfg2 = orast.Cfg(tile_size=8)
  q = torch.tensor([[2.0, 3.0, 0.8, 0.6, 1.5, 0.9, 0.5], [5.0, 4.0, 0.6, -0.8, 2.0, 1.0, 0.4]], dtype=torch.float64)
  o2, r2 = one_tile(2)
  _, _, v2 = orast.forward(q, torch.rand(2, 3, dtype=torch.float64), r2, o2, (8, 8), cfg2)
  assert torch.allclose(orast.active_visibility(q, r2, o2, (8, 8), cfg2), v2, rtol=1e-12)


@pytest.mark.parametrize('tile,seed', [(8, 0), (16, 1)])
def test_active_visibility_bounds_on_a_saturating_scene(tile, seed):
  """On a crowded random scene (opaque splats: pixels saturate) the backward's visibility lies below the forward's by no
  more than (1 - saturate_threshold) per pixel of the tiles the splat is listed in, never above it, and equals the colour
  gradient of a unit loss (backward.py:197) — the three properties the GPU tests of the opt-in rely on."""
  torch.manual_seed(seed)
  size = (48, 40)
  g = random_2d_gaussians(300, size, scale_factor=3.0, alpha_range=(0.6, 0.99))
  p, f = project_gaussians2d(g).double(), g.feature.double()
  o2p, ranges, _ = omap.map_to_tiles(p.numpy().astype(np.float32), g.depths.numpy().astype(np.float32), size, tile, 1.0 / 255.0)
  o2p, ranges = torch.from_numpy(o2p), torch.from_numpy(ranges)
  cfg = orast.Cfg(tile_size=tile)
  img, alpha, vis = orast.forward(p, f, ranges, o2p, size, cfg)
  act = orast.active_visibility(p, ranges, o2p, size, cfg)
  assert float(alpha.max()) > cfg.saturate_threshold                      # something does saturate
  gap = vis - act
  assert float(gap.min()) >= 0.0 and float(gap.max()) > 0.0
  # every pixel hands at most 1 - saturate_threshold to the splats behind its saturation point
  assert float(gap.sum()) <= (1.0 - cfg.saturate_threshold) * size[0] * size[1] * 1.0000001
  _, gf, _ = orast.backward(p, f, ranges, o2p, img, torch.ones_like(img), size, cfg)
  assert torch.allclose(gf[:, 0], act, rtol=1e-12, atol=1e-15)
