"""Fractional / visibility-aware optimisers.  CPU part: the oracle restatement of the
moment kernels is pinned by the identity with torch.optim.Adam; GPU part: the HIP kernel vs the oracle
and the optimizer classes end to end."""
import math

import pytest
import torch

from oracle import optim as oopt


def test_oracle_adam_with_unit_weights_is_torch_adam():
  torch.manual_seed(0)
  n, d = 50, 3
  p0 = torch.randn(n, d)
  ref = p0.clone().requires_grad_(True)
  opt = torch.optim.Adam([ref], lr=0.01, betas=(0.9, 0.999), eps=1e-16)
  p = p0.clone()
  m, v, tw = torch.zeros(n, d), torch.zeros(n, d), torch.zeros(n)
  idx = torch.arange(n)
  sat = 1 - math.exp(-2.0)
  for it in range(5):
    g = torch.randn(n, d)
    ref.grad = g.clone()
    before = ref.detach().clone()
    opt.step()
    adam_step = before - ref.detach()
    tw += 1.0
    step = oopt.fractional_step(0, False, idx, torch.ones(n), m, v, tw, g, 0.01, (0.9, 0.999), 1e-16, True)
    assert torch.allclose(step, adam_step, rtol=1e-4, atol=5e-7), (step - adam_step).abs().max()
    p -= step * sat
  assert torch.isfinite(p).all()


def _random_case(seed, d, vector):
  torch.manual_seed(seed)
  n, mcount = 1000, 400
  idx = torch.randperm(n)[:mcount].sort().values
  weight = torch.rand(mcount) * 1.5 + 0.01
  m = torch.randn(n, d) * 0.1
  v = (torch.rand(n) if vector else torch.rand(n, d)) * 0.1
  tw = torch.rand(n) * 5 + weight.max()
  grad = torch.randn(n, d)
  return idx, weight, m, v, tw, grad


@pytest.mark.gpu
@pytest.mark.parametrize('kind', [0, 1])
@pytest.mark.parametrize('vector,d', [(False, 1), (False, 3), (False, 48), (True, 3), (True, 4)])
@pytest.mark.parametrize('bias_correction', [True, False])
def test_kernel_matches_oracle(kind, vector, d, bias_correction):
  from taichi_splatting_amd.optim.fractional import fractional_step
  idx, weight, m, v, tw, grad = _random_case(kind * 10 + d, d, vector)
  m_o, v_o = m.clone(), v.clone()
  want = oopt.fractional_step(kind, vector, idx, weight, m_o, v_o, tw, grad, 0.02, (0.9, 0.99), 1e-16, bias_correction)
  dev = 'cuda:0'
  m_g, v_g = m.to(dev), v.to(dev)
  step = torch.zeros(idx.shape[0], d, device=dev)
  fractional_step(kind, vector, step, idx.to(dev), weight.to(dev), m_g, v_g, tw.to(dev), grad.to(dev), 0.02,
                  (0.9, 0.99), 1e-16, bias_correction)
  # float32 pow(beta, w) differs by an ulp between host and device; 1 - beta^w amplifies it ~100x
  assert torch.allclose(step.cpu(), want, rtol=2e-4, atol=1e-6), (step.cpu() - want).abs().max()
  assert torch.allclose(m_g.cpu(), m_o, rtol=2e-4, atol=1e-6)
  assert torch.allclose(v_g.cpu(), v_o, rtol=2e-4, atol=1e-6)


@pytest.mark.gpu
def test_optimizer_classes_end_to_end():
  from taichi_splatting_amd.optim import (FractionalAdam, SparseLaProp, VisibilityAwareAdam, ParameterClass)
  dev = 'cuda:0'
  torch.manual_seed(0)
  n = 500
  tensors = dict(position=torch.randn(n, 3, device=dev), feature=torch.randn(n, 3, 4, device=dev),
                 label=torch.arange(n, device=dev))
  groups = dict(position=dict(lr=0.01, type='vector'), feature=dict(lr=0.02, type='scalar'))
  params = ParameterClass(tensors, groups, optimizer=VisibilityAwareAdam, vis_beta=0.5)
  target = torch.zeros(n, 3, device=dev)
  first = None
  for it in range(30):
    params.zero_grad()
    loss = ((params.position - target) ** 2).sum() + (params.feature ** 2).sum()
    loss.backward()
    idx = torch.arange(0, n, 2, device=dev)
    vis = torch.rand(idx.shape[0], device=dev) + 0.1
    params.step(indexes=idx, visibility=vis)
    first = first if first is not None else float(loss)
  assert float(loss) < first
  # only the visible (even) rows moved
  assert torch.equal(params.position.detach()[1::2], tensors['position'][1::2])
  # filtering / appending keeps the optimizer state aligned
  sub = params[torch.arange(0, 100, device=dev)]
  assert sub.batch_size == (100,) and sub.tensor_state['position']['total_weight'].shape == (100,)
  more = sub.append_tensors({k: v.detach()[:10] for k, v in sub.tensors.items()})
  assert more.batch_size == (110,) and more.tensor_state['feature']['m'].shape[0] == 110
  assert set(params.learning_rates) == {'position', 'feature'}
  params.set_learning_rate(position=0.5)
  assert params.learning_rates['position'] == 0.5

  # the plain fractional optimisers
  for cls, kw in ((FractionalAdam, dict(weight=True)), (SparseLaProp, dict(weight=False))):
    p = torch.nn.Parameter(torch.randn(200, 3, device=dev))
    opt = cls([dict(params=[p], name='p', type='scalar')], lr=0.05)
    before = float((p ** 2).sum())
    for _ in range(20):
      opt.zero_grad()
      (p ** 2).sum().backward()
      idx = torch.arange(200, device=dev)
      if kw['weight']:
        opt.step(idx, torch.full((200,), 0.7, device=dev))
      else:
        opt.step(idx)
    assert float((p ** 2).sum()) < before
