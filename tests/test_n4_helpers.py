"""N4: Morton-code oracle (known answers) and the 2D densification helpers of
misc/renderer2d.py (pure torch, CPU)."""
import math

import numpy as np
import pytest
import torch

from oracle import morton as omorton
from taichi_splatting_amd.misc import renderer2d as r2d
from taichi_splatting_amd.testing import random_2d_gaussians


def test_morton_known_answers():
  cell = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [3, 3, 3], [2 ** 21 - 1] * 3, [2 ** 21, 0, 0], [5, 0, 2]], dtype=np.uint32)
  codes = omorton.cell_code64(cell)
  assert codes[:4].tolist() == [1, 2, 4, 63]
  assert int(codes[4]) == (1 << 63) - 1                      # all 63 bits
  assert int(codes[5]) == 0                                  # bit 21 is masked off (21 bits per axis)
  assert int(codes[6]) == 0b1000001 + (1 << 5) - 0           # x = 101b -> bits 0 and 6, z = 10b -> bit 5
  rng = np.random.default_rng(0)
  cells = rng.integers(0, 2 ** 21, size=(2000, 3)).astype(np.uint32)
  assert np.array_equal(omorton.cell_code64(cells), omorton.interleave_bitwise(cells))


def test_morton_grid_and_order():
  rng = np.random.default_rng(1)
  pts = rng.uniform(-3, 5, size=(500, 3)).astype(np.float32)
  res = 0.01
  codes = omorton.morton_codes(pts, res)
  lower, inc, _ = omorton.grid_at_resolution(pts, res)
  cell = np.floor((pts - lower) / inc).astype(np.int64)
  # decode: every third bit back to a coordinate
  for a in range(3):
    dec = np.zeros(500, dtype=np.int64)
    for b in range(21):
      dec |= ((codes >> np.uint64(3 * b + a)) & np.uint64(1)).astype(np.int64) << b
    assert np.array_equal(dec, cell[:, a])
  order = omorton.argsort(pts, res)
  assert np.all(np.diff(codes[order].astype(np.float64)) >= 0) and sorted(order.tolist()) == list(range(500))
  # dedup: coarse grid -> one representative per cell, all cells kept
  keep = omorton.argsort_dedup(pts, 1.0)
  coarse = omorton.morton_codes(pts, 1.0)
  assert len(set(coarse[keep].tolist())) == len(keep) == len(set(coarse.tolist()))


def test_point_basis_and_covariance():
  torch.manual_seed(0)
  g = random_2d_gaussians(50, (64, 64))
  basis = r2d.point_basis(g)
  v1 = g.rotation / g.rotation.norm(dim=1, keepdim=True)
  assert torch.allclose(basis[:, :, 0], v1 * g.scaling[:, 0:1]) and torch.allclose((basis[:, :, 0] * basis[:, :, 1]).sum(1), torch.zeros(50), atol=1e-5)
  cov = r2d.point_covariance(g)
  assert torch.allclose(cov, cov.transpose(1, 2)) and torch.allclose(torch.linalg.det(cov), (g.scaling[:, 0] * g.scaling[:, 1]) ** 2, rtol=1e-3)
  rot = r2d.point_rotation(g)
  assert torch.allclose(rot @ rot.transpose(1, 2), torch.eye(2).expand(50, 2, 2), atol=1e-5)
  # sampling statistics: offsets are distributed with the gaussian's covariance
  big = g[0:1].apply(lambda t: t.expand(20000, *t.shape[1:]).clone(), batch_size=[20000])
  s = r2d.sample_gaussians(big)
  emp = (s.T @ s) / s.shape[0]
  assert torch.allclose(emp, cov[0], rtol=0.1, atol=0.05 * cov[0].abs().max().item())


@pytest.mark.parametrize('n', [2, 3])
def test_split_helpers(n):
  torch.manual_seed(n)
  g = random_2d_gaussians(40, (64, 64))
  out = r2d.split_gaussians2d(g, n=n)
  assert out.position.shape == (40 * n, 2) and out.batch_size[0] == 40 * n
  assert torch.allclose(out.scaling, torch.repeat_interleave(g.scaling, n, 0) / math.sqrt(n), rtol=1e-5)
  assert torch.equal(out.feature, torch.repeat_interleave(g.feature, n, 0)) and bool((out.depths > 0).all())

  uni = r2d.uniform_split_gaussians2d(g, n=n, sep=0.7)
  axis = torch.argmax(g.log_scaling, dim=1)
  v1 = g.rotation / g.rotation.norm(dim=1, keepdim=True)
  v2 = torch.stack([-v1[:, 1], v1[:, 0]], -1)
  direction = torch.where(axis[:, None] == 0, v1, v2)
  sigma = g.scaling.gather(1, axis[:, None])
  steps = torch.linspace(-0.7, 0.7, n)
  want = g.position[:, None, :] + steps[None, :, None] * (direction * sigma)[:, None, :]
  assert torch.allclose(uni.position.view(40, n, 2), want, atol=1e-4)
  # the split axis shrinks by sqrt(n)/n, the other one is unchanged
  shrink = uni.scaling.view(40, n, 2)[:, 0] / g.scaling
  assert torch.allclose(shrink.gather(1, axis[:, None]), torch.full((40, 1), math.sqrt(n) / n), rtol=1e-4)
  assert torch.allclose(shrink.gather(1, 1 - axis[:, None]), torch.ones(40, 1), rtol=1e-4)
  rnd = r2d.uniform_split_gaussians2d(g, n=n, random_axis=True)
  assert rnd.position.shape == (40 * n, 2)
