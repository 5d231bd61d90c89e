"""-m gpu: randomised (hypothesis) property tests of the integer primitives and of the render path's
size-independent invariants: sortedness + stability, scan == cumsum, routing == brute force, and
"what a strip renders is what the full frame renders" for random scenes / strip boundaries."""
import numpy as np
import pytest
import torch
from hypothesis import example, given, settings, strategies as st, HealthCheck

from taichi_splatting_amd import cuda_lib

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
COMMON = dict(deadline=None, max_examples=60, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])


@settings(**COMMON)
@given(n=st.integers(0, 300_000), hi=st.integers(1, 1 << 20), seed=st.integers(0, 1 << 30))
def test_scan_is_cumsum(n, hi, seed):
  g = torch.Generator(device='cpu').manual_seed(seed)
  x = torch.randint(0, min(hi, (1 << 31) // max(n, 1)), (n,), dtype=torch.int32, generator=g).to(DEV)
  out, total = cuda_lib.full_cumsum(x)
  want = torch.cumsum(x.long(), 0)
  assert out.shape[0] == n + 1 and int(out[0]) == 0
  assert torch.equal(out[1:].long(), want) and total == (int(want[-1]) if n else 0)


@settings(**COMMON)
@given(n=st.integers(1, 200_000), start_bit=st.integers(0, 24), width=st.integers(1, 32), distinct=st.integers(1, 1 << 16),
       wide=st.booleans(), seed=st.integers(0, 1 << 30))
def test_radix_sort_is_a_stable_sort_on_the_bit_range(n, start_bit, width, distinct, wide, seed):
  g = torch.Generator(device='cpu').manual_seed(seed)
  total_bits = 62 if wide else 31
  end_bit = min(start_bit + width, total_bits)
  keys = (torch.randint(0, distinct, (n,), dtype=torch.int64, generator=g) * 2654435761) % (1 << total_bits)
  keys = keys.to(torch.int64 if wide else torch.int32).to(DEV)
  vals = torch.arange(n, dtype=torch.int32, device=DEV)
  ko, vo = cuda_lib.radix_sort_pairs(keys, vals, start_bit=start_bit, end_bit=end_bit)
  field = (keys.long() >> start_bit) & ((1 << (end_bit - start_bit)) - 1)
  _, order = torch.sort(field, stable=True)
  assert torch.equal(vo.long(), order)
  assert torch.equal(ko.long(), keys.long()[order])


@settings(**COMMON)
@given(n=st.integers(0, 60_000), world=st.integers(1, 16), tiles_high=st.integers(1, 70), seed=st.integers(0, 1 << 30))
@example(n=1, world=1, tiles_high=1, seed=114742218)      # the only splat is routed nowhere: nothing to pack
def test_strip_routing_matches_brute_force(n, world, tiles_high, seed):
  from taichi_splatting_amd import RasterConfig, distributed as D
  from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
  from taichi_splatting_amd.testing import random_2d_gaussians
  torch.manual_seed(seed)
  size = (160, tiles_high * 16 - (seed % 7))
  cuts = sorted(torch.randint(0, tiles_high + 1, (world - 1,)).tolist())
  bounds = [0] + cuts + [tiles_high]
  if n == 0:
    return
  g = random_2d_gaussians(n, size, scale_factor=1.0 + (seed % 5), alpha_range=(0.0, 0.9)).to(DEV)
  p = project_gaussians2d(g)
  cfg = RasterConfig()

  def loopback(send, send_counts, recv_counts, group):
    return send.clone()

  ids = torch.arange(n, device=DEV)
  g2, f2, d, gid, plan = D.exchange_to_strips(p, g.feature, g.depths, size, cfg, bounds, global_index=ids,
                                              exchange=loopback, return_plan=True)
  lo, hi = D.splat_row_span(p, size, cfg)
  b = torch.tensor(bounds, device=DEV)
  want = []
  for r in range(world):
    m = (hi > lo) & (lo < b[r + 1]) & (hi > b[r]) if bounds[r + 1] > bounds[r] else torch.zeros(n, dtype=torch.bool, device=DEV)
    want.append(m)
  # every non-empty strip receives exactly the splats whose row span meets it (empty strips may receive
  # splats that span across them), in index order
  counts = plan.send_counts
  offs = np.concatenate([[0], np.cumsum(counts)])
  for r in range(world):
    got = gid[offs[r]:offs[r + 1]]
    assert bool((got[1:] > got[:-1]).all()) if got.numel() > 1 else True
    if bounds[r + 1] > bounds[r]:
      assert torch.equal(got, want[r].nonzero().squeeze(1)), r
  assert torch.equal(g2, p[gid]) and torch.equal(d.view(-1), g.depths.view(-1)[gid])


@settings(deadline=None, max_examples=10, suppress_health_check=[HealthCheck.too_slow])
@given(n=st.integers(100, 30_000), w=st.integers(17, 400), h=st.integers(17, 300), tile=st.sampled_from([8, 16, 32]),
       scale=st.floats(0.3, 6.0), seed=st.integers(0, 1 << 30), cut=st.floats(0.0, 1.0))
def test_strips_render_the_rows_of_the_full_frame(n, w, h, tile, scale, seed, cut):
  from taichi_splatting_amd import RasterConfig, rasterize_with_tiles, map_to_tiles
  from taichi_splatting_amd.mapper.tile_mapper import map_to_tiles_strip
  from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
  from taichi_splatting_amd.testing import random_2d_gaussians
  torch.manual_seed(seed)
  size = (w, h)
  cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2))
  g = random_2d_gaussians(n, size, scale_factor=scale).to(DEV)
  p = project_gaussians2d(g)
  o2p, ranges = map_to_tiles(p, g.depths, size, cfg)
  full = rasterize_with_tiles(p, g.feature, o2p, ranges.view(-1, 2), size, cfg)
  tiles_high = (h + tile - 1) // tile
  mid = int(round(cut * tiles_high))
  for rows in ((0, mid), (mid, tiles_high)):
    o2p_s, ranges_s = map_to_tiles_strip(p, g.depths, size, cfg, tile_rows=rows)
    out = rasterize_with_tiles(p, g.feature, o2p_s, ranges_s.view(-1, 2), size, cfg, tile_rows=rows, crop_to_rows=True)
    y0, y1 = min(rows[0] * tile, h), min(rows[1] * tile, h)
    assert torch.equal(out.image, full.image[y0:y1]) and torch.equal(out.image_weight, full.image_weight[y0:y1])
    r_full = ranges.view(-1, 2)[rows[0] * ((w + tile - 1) // tile):rows[1] * ((w + tile - 1) // tile)]
    r_strip = ranges_s.view(-1, 2)[rows[0] * ((w + tile - 1) // tile):rows[1] * ((w + tile - 1) // tile)]
    assert torch.equal(r_full[:, 1] - r_full[:, 0], r_strip[:, 1] - r_strip[:, 0])
