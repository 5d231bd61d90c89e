"""CPU tests of round-4 host logic (no GPU): the reference-order debug lists, the optimiser-state helpers'
reference semantics, the frame's pinned K ring."""
import numpy as np
import pytest
import torch

from oracle import raster as orast


def test_with_reference_tail_reproduces_the_reference_visiting_order():
  from taichi_splatting_amd.mapper.tile_mapper import with_reference_tail
  rng = np.random.default_rng(0)
  group = 16                                   # tile 4 -> groups of 16: small enough to hit every case
  counts = np.array([0, 1, 15, 16, 17, 31, 32, 33, 40, 47, 48, 49, 100, 0, 5], dtype=np.int64)
  ends = np.cumsum(counts)
  starts = ends - counts
  ranges = np.stack([np.where(counts > 0, starts, 0), np.where(counts > 0, ends, 0)], axis=1).astype(np.int32)
  o2p = rng.integers(0, 1000, size=int(ends[-1])).astype(np.int32)
  out, new_ranges = with_reference_tail(torch.from_numpy(o2p), torch.from_numpy(ranges).view(3, 5, 2), tile_size=4)
  assert new_ranges.shape == (3, 5, 2) and new_ranges.dtype == torch.int32 and out.dtype == torch.int32
  new_ranges = new_ranges.view(-1, 2).numpy()
  revisits = 0
  for t, c in enumerate(counts):
    want = o2p[starts[t]:ends[t]][np.array(orast.reference_tail_order(int(c), group), dtype=np.int64)] if c else o2p[:0]
    got = out.numpy()[new_ranges[t, 0]:new_ranges[t, 1]]
    assert np.array_equal(got, want), (t, c)
    revisits += len(want) - c
  assert revisits > 0, "the case list must contain tiles with a partially filled last group"
  # lists that fit one group are untouched
  one = torch.from_numpy(np.array([[0, 10]], dtype=np.int32))
  out1, r1 = with_reference_tail(torch.arange(10, dtype=torch.int32), one, tile_size=4)
  assert torch.equal(out1, torch.arange(10, dtype=torch.int32)) and torch.equal(r1, one)


def test_optimizer_state_helpers_follow_the_reference_semantics():
  from taichi_splatting_amd.optim import util
  p = torch.zeros(7, 3)
  st = {}
  v, m = util.get_vector_state(st, p)
  assert v.shape == (7, 3) and m.shape == (7,)
  st = {}
  v, m = util.get_scalar_state(st, p)
  assert v.shape == (7, 3) and m.shape == (7, 3)
  # the pair is created together, keyed on 'v' alone (reference optim/util.py:5-18)
  st = {'m': torch.ones(7)}
  v, m = util.get_vector_state(st, p)
  assert float(m.sum()) == 0.0, "a state with only 'm' gets both moments reset, as in the reference"
  st = {'v': torch.ones(7, 3)}
  with pytest.raises(KeyError):
    util.get_vector_state(st, p)
  # running_vis takes the whole shape
  assert util.get_running_vis({}, (5, 2), torch.device('cpu')).shape == (5, 2)
  assert util.get_running_vis({}, 5, torch.device('cpu')).shape == (5,)
  assert util.get_total_weight({}, 6, torch.device('cpu')).shape == (6,)


def test_visibility_weight_is_the_order4_power_mean():
  from taichi_splatting_amd.optim.visibility_aware import _track_visibility
  running = torch.tensor([0.5, 0.2, 0.9, 0.0])
  seen = torch.tensor([0.1, 0.8])
  idx = torch.tensor([2, 0])
  beta = 0.7
  prev = running[idx].clone()
  w = _track_visibility(running, seen, idx, beta)
  want = ((1 - beta) * seen ** 4 + beta * prev ** 4) ** 0.25
  assert torch.allclose(running[idx], want, rtol=1e-6)
  assert torch.allclose(w, seen / want, rtol=1e-6)
  assert float(running[1]) == pytest.approx(0.2) and float(running[3]) == 0.0


def test_mapper_choice_follows_overlaps_per_gaussian_with_hysteresis():
  # frame.py picks the mapper's launch sequence per scene shape from the last overlap total (same lists either way):
  # depth pre-sort above ~3.5 overlaps per gaussian, storage-order emission + per-tile depth sort below
  from taichi_splatting_amd import _lib, frame
  key = ('test-shape',)
  try:
    n = 1000
    frame._choose_mapper(key, 2100, n)
    assert frame._mapper_mode[key] == _lib.MAPPER_DIRECT
    frame._choose_mapper(key, int(frame.PRESORT_ABOVE * n) + 10, n)
    assert frame._mapper_mode[key] == _lib.MAPPER_PRESORT
    frame._choose_mapper(key, int(0.5 * (frame.PRESORT_ABOVE + frame.DIRECT_BELOW) * n), n)                   # inside the band: stays
    assert frame._mapper_mode[key] == _lib.MAPPER_PRESORT
    frame._choose_mapper(key, int(frame.DIRECT_BELOW * n) - 10, n)
    assert frame._mapper_mode[key] == _lib.MAPPER_DIRECT
    frame._choose_mapper(key, int(0.5 * (frame.PRESORT_ABOVE + frame.DIRECT_BELOW) * n), n)
    assert frame._mapper_mode[key] == _lib.MAPPER_DIRECT
  finally:
    frame._mapper_mode.pop(key, None)
