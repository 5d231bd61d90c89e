"""Round 5, CPU: the reference's module-level names of optim/visibility_aware.py, frame bookkeeping under failure."""
import math

import pytest
import torch


def test_visibility_aware_public_helpers_match_the_reference_signatures():
  # reference optim/visibility_aware.py:11-52: get_running_vis, lerp, max_decaying, power_lerp, update_visibility,
  # set_indexes are importable module-level names with these argument orders
  from taichi_splatting_amd.optim.visibility_aware import (get_running_vis, lerp, max_decaying, power_lerp, set_indexes,
                                                            update_visibility)
  torch.manual_seed(0)
  running, seen, idx = torch.rand(100), torch.rand(30), torch.randperm(100)[:30]
  before = running.clone()
  weight = update_visibility(running, seen, idx, None, beta=0.9)
  want = (seen ** 4 + (before[idx] ** 4 - seen ** 4) * 0.9) ** 0.25        # power_lerp(beta, visibility, running, k=4)
  assert torch.allclose(running[idx], want) and torch.allclose(weight, seen / want.clamp_min(1e-12))
  untouched = torch.ones(100, dtype=torch.bool); untouched[idx] = False
  assert torch.equal(running[untouched], before[untouched])
  assert torch.allclose(power_lerp(0.9, seen, before[idx], k=4), want)
  assert torch.allclose(lerp(0.25, seen, before[idx]), seen + (before[idx] - seen) * 0.25)
  assert torch.allclose(max_decaying(0.25, seen, before[idx]), torch.maximum(seen, lerp(0.25, seen, before[idx])))
  scattered = set_indexes(before, seen, idx)
  assert torch.equal(scattered[idx], seen) and int((scattered != 0).sum()) == 30 and scattered.shape == before.shape
  state = {}
  assert get_running_vis(state, 5, torch.device('cpu')).shape == (5,) and 'running_vis' in state
  # exp_lerp (reference :19-22): log(lerp(t, e^a, e^b)), finite where the naive form overflows; total_weight is a
  # required positional argument of update_visibility, as in the reference
  from taichi_splatting_amd.optim.visibility_aware import exp_lerp
  a, b = torch.randn(50), torch.randn(50)
  assert torch.allclose(exp_lerp(0.3, a, b), torch.log(torch.lerp(torch.exp(a), torch.exp(b), 0.3)), atol=1e-6)
  big = exp_lerp(0.5, torch.tensor([1000.0]), torch.tensor([998.0]))
  assert torch.isfinite(big).all() and abs(float(big) - (1000.0 + math.log(0.5 + 0.5 * math.exp(-2.0)))) < 1e-3
  with pytest.raises(TypeError):
    update_visibility(running, seen, idx)
