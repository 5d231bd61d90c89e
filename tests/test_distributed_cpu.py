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
    size = (160, 112)
    g = random_2d_gaussians(1500, size, scale_factor=1.5)
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


def test_two_rank_strip_backward_allreduce():
  world = 2
  port = _free_port()
  mgr = mp.Manager()
  ret = mgr.dict()
  mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
  assert dict(ret) == {0: True, 1: True}
