"""-m gpu: tile mapper (count -> scan -> emit -> radix sort -> ranges) vs the numpy oracle.
Integer outputs must match exactly; (point, tile) pairs numerically on the SAT decision boundary
are the only tolerated differences (float32 logf may differ by an ulp between host and device)."""
import numpy as np
import pytest
import torch

from oracle import mapper as omap
from taichi_splatting_amd import RasterConfig, map_to_tiles, pad_to_tile
from taichi_splatting_amd.mapper.tile_mapper import map_to_tiles_strip
from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
from taichi_splatting_amd.testing import random_2d_gaussians

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _check(p, depth, size, cfg, use_depth16=False, tile_rows=None):
  # both constructions of the lists (tile_mapper.py: 'direct' and 'presort') against the oracle, and each other
  a = _check_method(p, depth, size, cfg, use_depth16, tile_rows, 'direct')
  b = _check_method(p, depth, size, cfg, use_depth16, tile_rows, 'presort')
  assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
  return a


def _check_method(p, depth, size, cfg, use_depth16, tile_rows, method):
  o2p, ranges = map_to_tiles_strip(p.to(DEV), depth.to(DEV), size, cfg, use_depth16=use_depth16, tile_rows=tile_rows,
                                   method=method)
  w_o2p, w_ranges, _ = omap.map_to_tiles(p.numpy(), depth.numpy(), size, cfg.tile_size, cfg.alpha_threshold,
                                         use_depth16=use_depth16, tile_rows=tile_rows)
  assert o2p.dtype == torch.int32 and ranges.dtype == torch.int32
  assert tuple(ranges.shape) == w_ranges.shape
  if o2p.shape[0] == w_o2p.shape[0] and np.array_equal(ranges.cpu().numpy(), w_ranges):
    if use_depth16:
      # quantised depths tie often; order is (tile, q(depth), point) in both
      pass
    assert np.array_equal(o2p.cpu().numpy(), w_o2p)
    return o2p, ranges
  # tolerate only numerically borderline (point, tile) pairs
  def pairs(o, r):
    r = r.reshape(-1, 2)
    tiles_wide = r.shape[0] // ranges.shape[0]
    out = set()
    for t in range(r.shape[0]):
      for i in o[r[t, 0]:r[t, 1]]:
        out.add((int(i), t % ranges.shape[1], t // ranges.shape[1]))
    return out
  got, want = pairs(o2p.cpu().numpy(), ranges.cpu().numpy()), pairs(w_o2p, w_ranges)
  border = omap.borderline_pairs(p.numpy(), size, cfg.tile_size, cfg.alpha_threshold)
  assert (got ^ want) <= border, sorted(got ^ want)[:10]
  return o2p, ranges


@pytest.mark.parametrize('tile_size', [8, 16, 32])
@pytest.mark.parametrize('n,size,scale', [(1, (64, 64), 1.0), (1000, (320, 200), 0.5), (20000, (333, 210), 2.0),
                                          (50000, (1024, 768), 4.0)])
def test_map_to_tiles_matches_oracle(tile_size, n, size, scale):
  torch.manual_seed(n + tile_size)
  g = random_2d_gaussians(n, size, scale_factor=scale, alpha_range=(0.05, 1.0), depth_range=(0.1, 100.0))
  p = project_gaussians2d(g)
  cfg = RasterConfig(tile_size=tile_size, pixel_stride=(1, 1) if tile_size == 8 else (2, 2))
  _check(p, g.depths, size, cfg)


def test_depth_ties_keep_point_order_and_depth16():
  torch.manual_seed(0)
  size = (256, 256)
  g = random_2d_gaussians(5000, size, scale_factor=1.0)
  p = project_gaussians2d(g)
  depth = (torch.rand(5000, 1) * 4).floor() / 4      # only 4 distinct depths => ties everywhere
  _check(p, depth, size, RasterConfig())
  _check(p, torch.rand(5000, 1), size, RasterConfig(), use_depth16=True)


def test_edge_cases():
  cfg = RasterConfig()
  empty = torch.zeros((0, 7))
  o2p, ranges = map_to_tiles(empty.to(DEV), torch.zeros((0, 1), device=DEV), (100, 60), cfg)
  assert o2p.shape == (0,) and tuple(ranges.shape) == (4, 7, 2) and int(ranges.abs().sum()) == 0
  # everything off screen / alpha below the threshold => K == 0
  p = torch.tensor([[-500., -500., 1, 0, 2, 2, 0.5], [50., 30., 1, 0, 2, 2, 0.001]])
  o2p, ranges = map_to_tiles(p.to(DEV), torch.zeros((2, 1), device=DEV), (100, 60), cfg)
  assert o2p.shape == (0,) and int(ranges.abs().sum()) == 0
  assert pad_to_tile((100, 60), 16) == (112, 64)
  with pytest.raises(AssertionError):
    map_to_tiles(torch.zeros((3, 6), device=DEV), torch.zeros((3, 1), device=DEV), (64, 64), cfg)


def test_more_than_65535_tiles():
  # 2048x2048 @ tile 8 = 65536 tiles: asserts in the reference (tile_mapper.py:177-178)
  torch.manual_seed(1)
  size = (2048, 2048)
  g = random_2d_gaussians(30000, size, scale_factor=1.0)
  p = project_gaussians2d(g)
  _check(p, g.depths, size, RasterConfig(tile_size=8, pixel_stride=(1, 1)))


def test_strips_partition_the_overlaps():
  torch.manual_seed(2)
  size = (640, 400)
  g = random_2d_gaussians(20000, size, scale_factor=2.0)
  p = project_gaussians2d(g)
  cfg = RasterConfig()
  full, franges = _check(p, g.depths, size, cfg)
  total = 0
  for rows in ((0, 7), (7, 19), (19, 25)):
    o2p, ranges = _check(p, g.depths, size, cfg, tile_rows=rows)
    total += o2p.shape[0]
    counts = (ranges[..., 1] - ranges[..., 0])
    assert int(counts[:rows[0]].sum()) == 0 and int(counts[rows[1]:].sum()) == 0
    assert torch.equal(counts[rows[0]:rows[1]], (franges[..., 1] - franges[..., 0])[rows[0]:rows[1]])
  assert total == full.shape[0]
