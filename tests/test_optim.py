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
@pytest.mark.parametrize('kind', [0, 1])
@pytest.mark.parametrize('group_type,d', [('scalar', 1), ('scalar', 3), ('scalar', 16), ('scalar', 48), ('scalar', 100),
                                          ('vector', 1), ('vector', 3), ('vector', 4), ('vector', 48), ('vector', 200),
                                          ('local_vector', 2), ('local_vector', 3)])
@pytest.mark.parametrize('extras', [False, True])
def test_fused_update_matches_oracle(kind, group_type, d, extras):
  # ms_fractional_update (one kernel per group) against the restated host logic of the reference
  from taichi_splatting_amd.optim.fractional import Group, fused_update
  torch.manual_seed(kind * 1000 + d + 7 * extras)
  vector = group_type != 'scalar'
  idx, weight, m, v, tw, grad = _random_case(kind * 10 + d, d, vector)
  n, mc = grad.shape[0], idx.shape[0]
  param = torch.randn(n, d)
  basis = None
  if group_type == 'local_vector':
    q, _ = torch.linalg.qr(torch.randn(mc, d, d))
    basis = q * (torch.rand(mc, 1, d) + 0.3)               # orthogonal axes scaled per column, like point_basis
  grad_scale = torch.rand(mc) + 0.5 if extras else None
  mask_lr = torch.rand(d) if extras else None
  point_lr = torch.rand(n) + 0.5 if extras else None
  clip = 0.7 if extras else None
  if extras:
    grad[idx[0], 0] = float('inf')                           # non-finite steps are dropped
  p_o, m_o, v_o = param.clone(), m.clone(), v.clone()
  oopt.group_update(kind, group_type, p_o, grad, m_o, v_o, idx, weight, tw, 0.02, (0.9, 0.99), 1e-16, True,
                    grad_scale=grad_scale, basis=basis, clip=clip, mask_lr=mask_lr, point_lr=point_lr)
  dev = 'cuda:0'
  g = lambda t: t.to(dev) if t is not None else None
  p_g = param.to(dev)
  # state naming quirk of the reference: for vector groups state['v'] is the (N, D) first moment
  state = {'v': m.to(dev), 'm': v.to(dev)}
  group = Group(name='p', type=group_type, param=p_g, grad=g(grad), state=state, lr=0.02, betas=(0.9, 0.99), eps=1e-16,
                bias_correction=True, clip=clip, mask_lr=g(mask_lr), point_lr=g(point_lr))
  fused_update(group, g(weight), g(idx), g(tw), kind, g(basis), grad_scale=g(grad_scale))
  keep = torch.ones(n, dtype=torch.bool)
  if extras:
    keep[idx[0]] = False                                      # the row fed with inf: only "finite" is required
    assert torch.isfinite(p_g.cpu()[idx[0]]).all()
  assert torch.allclose(p_g.cpu()[keep], p_o[keep], rtol=3e-4, atol=2e-6), (p_g.cpu() - p_o)[keep].abs().max()
  assert torch.allclose(state['v'].cpu()[keep], m_o[keep], rtol=3e-4, atol=2e-6)
  assert torch.allclose(state['m'].cpu()[keep], v_o[keep], rtol=3e-4, atol=2e-6)
  untouched = torch.ones(n, dtype=torch.bool); untouched[idx] = False
  assert torch.equal(p_g.cpu()[untouched], param[untouched])


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
