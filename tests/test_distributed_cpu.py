"""N > 1 path on CPU (gloo, world_size 2): strip assignment and the single all-reduce of the
2D-boundary gradients.  Each rank computes the backward of ITS strip with the oracle standing in for
the GPU kernels (the product path itself needs a GPU); the reduced gradient must equal the
full-frame gradient."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from taichi_splatting_amd.distributed import strip_rows, allreduce_boundary_grads


def test_strip_rows_partition():
  for tiles_high in (1, 7, 8, 128, 255):
    for world in (1, 2, 3, 8):
      rows = [strip_rows(tiles_high, world, r) for r in range(world)]
      assert rows[0][0] == 0 and rows[-1][1] == tiles_high
      for a, b in zip(rows[:-1], rows[1:]):
        assert a[1] == b[0] and a[0] <= a[1]
  # weighted: boundaries follow the cumulative weight
  w = [0, 0, 10, 10, 0, 0, 10, 10]
  rows = [strip_rows(8, 2, r, w) for r in range(2)]
  assert rows == [(0, 4), (4, 8)] or rows[0][1] in (4, 5, 6)
  assert rows[0][1] == rows[1][0]
  heavy = [100, 1, 1, 1, 1, 1, 1, 1]
  assert strip_rows(8, 2, 0, heavy)[1] <= 2


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank, world, port, ret):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from oracle import mapper as omap, raster as orast
    from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
    from taichi_splatting_amd.testing import random_2d_gaussians
    torch.manual_seed(0)     # replicated scene
    size = (128, 96)
    g = random_2d_gaussians(700, size, scale_factor=1.5)
    p, f = project_gaussians2d(g).double(), g.feature.double()
    cfg = orast.Cfg()
    tiles_high = (size[1] + 15) // 16
    rows = strip_rows(tiles_high, world, rank)
    o2p, ranges, _ = omap.map_to_tiles(p.numpy(), g.depths.numpy(), size, 16, tile_rows=rows)
    o2p, ranges = torch.from_numpy(o2p), torch.from_numpy(ranges)
    img, _, _ = orast.forward(p, f, ranges, o2p, size, cfg, tile_rows=rows)
    G = torch.ones_like(img)
    gp, gf, _ = orast.backward(p, f, ranges, o2p, img, G, size, cfg, tile_rows=rows)
    gp, gf = allreduce_boundary_grads(gp, gf)

    # full frame on every rank for comparison
    o2p_f, ranges_f, _ = omap.map_to_tiles(p.numpy(), g.depths.numpy(), size, 16)
    o2p_f, ranges_f = torch.from_numpy(o2p_f), torch.from_numpy(ranges_f)
    img_f, _, _ = orast.forward(p, f, ranges_f, o2p_f, size, cfg)
    gp_f, gf_f, _ = orast.backward(p, f, ranges_f, o2p_f, img_f, torch.ones_like(img_f), size, cfg)
    ok = (torch.allclose(gp, gp_f, atol=1e-10) and torch.allclose(gf, gf_f, atol=1e-10)
          and torch.allclose(img[rows[0] * 16:rows[1] * 16], img_f[rows[0] * 16:rows[1] * 16], atol=1e-12)
          and float(img[:rows[0] * 16].abs().sum()) == 0.0)
    ret[rank] = bool(ok)
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])      # 700 rows: divisible by 2, padded for 3 (reduce-scatter + all-gather)
def test_strip_backward_reduce_scatter_all_gather(world):
  port = _free_port()
  mgr = mp.Manager()
  ret = mgr.dict()
  mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
  assert dict(ret) == {r: True for r in range(world)}


# ---- gaussian-sharded + strip-sharded path: routing and the all-to-all exchange (gloo) -----------

def test_routing_plan_matches_brute_force():
  from taichi_splatting_amd.distributed import route_to_strips, expand_routes
  torch.manual_seed(3)
  tiles_high, n = 13, 500
  lo = torch.randint(0, tiles_high, (n,))
  hi = torch.minimum(lo + torch.randint(0, 6, (n,)), torch.tensor(tiles_high))     # some empty (hi == lo)
  for bounds in ([0, 13], [0, 6, 13], [0, 4, 4, 9, 13], [0, 1, 2, 3, 13]):
    world = len(bounds) - 1
    first, copies, counts = route_to_strips(lo, hi, bounds)
    send_index, dest = expand_routes(first, copies, int(counts.sum()))
    want = [(d, i) for d in range(world) for i in range(n)
            if hi[i] > lo[i] and lo[i] < bounds[d + 1] and hi[i] > bounds[d]
            # empty strips in the middle of a span still receive the splat (harmless): count them too
            or (hi[i] > lo[i] and bounds[d] == bounds[d + 1] and lo[i] < bounds[d] < hi[i])]
    got = list(zip(dest.tolist(), send_index.tolist()))
    # every (strip, splat) intersection is routed, in (dest, index) order, and nothing is routed twice
    assert set(w for w in want if bounds[w[0]] != bounds[w[0] + 1]) <= set(got)
    assert got == sorted(set(got))
    assert counts.tolist() == [sum(1 for d, _ in got if d == r) for r in range(world)]


def _sharded_worker(rank, world, port, ret):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from oracle import mapper as omap, raster as orast
    from taichi_splatting_amd import RasterConfig
    from taichi_splatting_amd.distributed import shard_range, strip_bounds, exchange_to_strips
    from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
    from taichi_splatting_amd.testing import random_2d_gaussians
    torch.manual_seed(0)     # every rank builds the same scene and keeps its shard
    size = (96, 80)
    g = random_2d_gaussians(400, size, scale_factor=1.5)
    g.depths[::7] = g.depths[0]                   # depth ties: order must fall back to the gaussian index
    p_full, f_full, d_full = project_gaussians2d(g), g.feature.clone(), g.depths.squeeze(-1) if g.depths.dim() > 1 else g.depths
    cfg, rcfg = orast.Cfg(), RasterConfig()
    tiles_high = (size[1] + 15) // 16
    bounds = strip_bounds(tiles_high, world)
    rows = (bounds[rank], bounds[rank + 1])

    b, e = shard_range(p_full.shape[0], world, rank)
    p = p_full[b:e].clone().requires_grad_(True)
    f = f_full[b:e].clone().requires_grad_(True)
    gid = torch.arange(b, e)
    g2, f2, d2, gid2 = exchange_to_strips(p, f, d_full[b:e], size, rcfg, bounds, global_index=gid)
    # received splats: by source rank, in source order = ascending global index
    assert torch.equal(gid2, torch.sort(gid2).values)
    assert torch.equal(g2.detach(), p_full[gid2]) and torch.equal(d2, d_full[gid2])

    o2p, ranges, _ = omap.map_to_tiles(g2.detach().numpy(), d2.numpy(), size, 16, tile_rows=rows)
    o2p, ranges = torch.from_numpy(o2p), torch.from_numpy(ranges)
    img, _, _ = orast.forward(g2.detach().double(), f2.detach().double(), ranges, o2p, size, cfg, tile_rows=rows)
    G = torch.ones_like(img)
    gp, gf, _ = orast.backward(g2.detach().double(), f2.detach().double(), ranges, o2p, img, G, size, cfg, tile_rows=rows)
    torch.autograd.backward([g2, f2], [gp.float(), gf.float()])     # reverse all-to-all + scatter-add

    o2p_f, ranges_f, _ = omap.map_to_tiles(p_full.numpy(), d_full.numpy(), size, 16)
    o2p_f, ranges_f = torch.from_numpy(o2p_f), torch.from_numpy(ranges_f)
    img_f, _, _ = orast.forward(p_full.double(), f_full.double(), ranges_f, o2p_f, size, cfg)
    gp_f, gf_f, _ = orast.backward(p_full.double(), f_full.double(), ranges_f, o2p_f, img_f, torch.ones_like(img_f), size, cfg)
    r0, r1 = rows[0] * 16, min(rows[1] * 16, size[1])
    ok = (torch.allclose(img[r0:r1], img_f[r0:r1], atol=1e-12)
          and torch.allclose(p.grad.double(), gp_f[b:e], rtol=1e-5, atol=1e-5 * float(gp_f.abs().max()))
          and torch.allclose(f.grad.double(), gf_f[b:e], rtol=1e-5, atol=1e-6)
          and float(gp_f[b:e].abs().sum()) > 0)
    ret[rank] = bool(ok)
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_exchange_matches_full_frame(world):
  port = _free_port()
  mgr = mp.Manager()
  ret = mgr.dict()
  mp.spawn(_sharded_worker, args=(world, port, ret), nprocs=world, join=True)
  assert dict(ret) == {r: True for r in range(world)}


# ---- a rank whose loss does not depend on its strip still joins the gradient collective -----------

def _constant_loss_worker(rank, world, port, ret):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from taichi_splatting_amd import RasterConfig
    from taichi_splatting_amd.distributed import shard_range, exchange_to_strips, backward_through_exchange
    from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
    from taichi_splatting_amd.testing import random_2d_gaussians
    torch.manual_seed(0)
    size = (96, 80)
    g = random_2d_gaussians(200, size, scale_factor=1.5)
    p_full, f_full, d_full = project_gaussians2d(g), g.feature.clone(), g.depths.reshape(-1)
    tiles_high = (size[1] + 15) // 16
    bounds = [0] + [tiles_high] * world            # rank 0 renders everything, the other strips are empty
    b, e = shard_range(p_full.shape[0], world, rank)
    p = p_full[b:e].clone().requires_grad_(True)
    f = f_full[b:e].clone().requires_grad_(True)
    g2, f2, d2, gid2, plan = exchange_to_strips(p, f, d_full[b:e], size, RasterConfig(), bounds, global_index=torch.arange(b, e),
                                                return_plan=True)
    # what a loss_fn does on an empty strip: a constant.  Without the zero-gradient pass this rank would skip the
    # reverse all-to-all inside backward() and rank 0 would hang in it.
    loss = (g2.sum() + 2 * f2.sum()) if rank == 0 else torch.zeros(())
    backward_through_exchange(loss, g2, f2, plan.ran)
    assert plan.ran.get('backward')
    visible = p.grad.abs().sum(dim=1) > 0
    ok = (torch.all(p.grad[visible] == 1.0) and torch.all(f.grad[visible] == 2.0) and int(visible.sum()) > 0
          and (rank == 0 or g2.shape[0] == 0))
    ret[rank] = bool(ok)
  finally:
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_constant_loss_rank_still_joins_reverse_exchange():
  world = 3
  port = _free_port()
  mgr = mp.Manager()
  ret = mgr.dict()
  mp.spawn(_constant_loss_worker, args=(world, port, ret), nprocs=world, join=True)
  assert dict(ret) == {r: True for r in range(world)}


def test_overlap_balanced_bounds_follow_the_work():
  from oracle import mapper as omap
  from taichi_splatting_amd import RasterConfig
  from taichi_splatting_amd.distributed import overlap_balanced_bounds
  from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
  from taichi_splatting_amd.testing import random_2d_gaussians
  torch.manual_seed(0)
  size = (256, 512)                                     # 32 tile rows
  g = random_2d_gaussians(6000, size, scale_factor=1.5)
  g.position[:, 1] = g.position[:, 1] ** 2 / size[1]    # crowd the splats towards the top of the image
  p = project_gaussians2d(g)
  cfg = RasterConfig()
  for world in (1, 2, 4, 8):
    bounds = overlap_balanced_bounds(p, size, cfg, world)
    assert bounds[0] == 0 and bounds[-1] == 32 and len(bounds) == world + 1
    assert all(a <= b for a, b in zip(bounds[:-1], bounds[1:]))
  bounds = overlap_balanced_bounds(p, size, cfg, 4)
  # true overlaps per strip (oracle mapper) are far better balanced than with even strips
  o2p, ranges, _ = omap.map_to_tiles(p.numpy(), g.depths.numpy(), size, 16)
  per_row = (ranges[..., 1] - ranges[..., 0]).sum(axis=1)
  work = lambda b: [int(per_row[b[i]:b[i + 1]].sum()) for i in range(len(b) - 1)]
  balanced, even = work(bounds), work([0, 8, 16, 24, 32])
  assert max(balanced) < 0.6 * max(even), (balanced, even)
  assert max(balanced) < 1.35 * (sum(balanced) / 4), balanced
