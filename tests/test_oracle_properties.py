"""Randomised (hypothesis) invariants of the CPU oracle's tile mapper and Morton ordering: the checker
itself must satisfy the properties the GPU path is later held to."""
import numpy as np
import torch
from hypothesis import given, settings, strategies as st, HealthCheck

from oracle import mapper as omap, morton as omorton
from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
from taichi_splatting_amd.testing import random_2d_gaussians

COMMON = dict(deadline=None, max_examples=20, suppress_health_check=[HealthCheck.too_slow])


@settings(**COMMON)
@given(n=st.integers(1, 400), w=st.integers(9, 200), h=st.integers(9, 150), tile=st.sampled_from([8, 16, 32]),
       scale=st.floats(0.3, 8.0), seed=st.integers(0, 1 << 30), cut=st.floats(0.0, 1.0))
def test_mapper_order_and_strip_decomposition(n, w, h, tile, scale, seed, cut):
  torch.manual_seed(seed)
  size = (w, h)
  g = random_2d_gaussians(n, size, scale_factor=scale, alpha_range=(0.0, 0.9))
  g.depths[::5] = g.depths[0].clone()               # ties
  p = project_gaussians2d(g).numpy()
  d = g.depths.numpy()
  o2p, ranges, counts = omap.map_to_tiles(p, d, size, tile)
  ranges = ranges.reshape(-1, 2)
  assert counts.sum() == o2p.shape[0] == (ranges[:, 1] - ranges[:, 0]).sum()
  bits = omap.depth_key_bits(d, False)
  for s, e in ranges:
    ids = o2p[s:e]
    key = bits[ids].astype(np.int64) * (n + 1) + ids          # (depth bits, index) strictly increasing
    assert np.all(np.diff(key) > 0)
  # strips partition the overlaps
  tiles_high, tiles_wide = (h + tile - 1) // tile, (w + tile - 1) // tile
  mid = int(round(cut * tiles_high))
  total = 0
  for rows in ((0, mid), (mid, tiles_high)):
    o2p_s, ranges_s, _ = omap.map_to_tiles(p, d, size, tile, tile_rows=rows)
    ranges_s = ranges_s.reshape(-1, 2)
    total += o2p_s.shape[0]
    for t in range(rows[0] * tiles_wide, rows[1] * tiles_wide):
      assert np.array_equal(o2p_s[ranges_s[t, 0]:ranges_s[t, 1]], o2p[ranges[t, 0]:ranges[t, 1]])
  assert total == o2p.shape[0]


@settings(**COMMON)
@given(n=st.integers(1, 2000), res=st.floats(1e-3, 2.0), seed=st.integers(0, 1 << 30))
def test_morton_order_properties(n, res, seed):
  rng = np.random.default_rng(seed)
  pts = rng.uniform(-5, 5, size=(n, 3)).astype(np.float32)
  codes = omorton.morton_codes(pts, res)
  order = omorton.argsort(pts, res)
  assert sorted(order.tolist()) == list(range(n)) and np.all(np.diff(codes[order].astype(np.float64)) >= 0)
  keep = omorton.argsort_dedup(pts, res)
  assert len(set(codes[keep].tolist())) == len(keep) == len(set(codes.tolist()))
  # locality: points in the same cell have the same code
  lower, inc, _ = omorton.grid_at_resolution(pts, res)
  cell = np.floor((pts - lower) / inc).astype(np.int64)
  _, inv = np.unique(cell, axis=0, return_inverse=True)
  assert len(set(zip(inv.reshape(-1).tolist(), codes.tolist()))) == len(set(codes.tolist()))
