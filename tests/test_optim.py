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


FIVE_GROUPS = [('position', 'scalar', 3), ('log_scaling', 'vector', 3), ('rotation', 'scalar', 4), ('alpha_logit', 'scalar', 1),
               ('feature', 'vector', 48), ('sh', 'scalar', 48), ('offset', 'local_vector', 3), ('wide', 'scalar', 12)]


def _five_group_case(seed, n=3000, mcount=1700, unaligned=False):
  torch.manual_seed(seed)
  idx = torch.randperm(n)[:mcount].sort().values
  weight = torch.rand(mcount) * 1.5 + 0.01
  tw = torch.rand(n) * 5 + weight.max()
  groups = {}
  for name, kind, d in FIVE_GROUPS:
    groups[name] = dict(type=kind, d=d, param=torch.randn(n, d), grad=torch.randn(n, d), m=torch.randn(n, d) * 0.1,
                        v=(torch.rand(n, d) if kind == 'scalar' else torch.rand(n)) * 0.1)
  q, _ = torch.linalg.qr(torch.randn(mcount, 3, 3))
  basis = q * (torch.rand(mcount, 1, 3) + 0.3)
  return idx, weight, tw, groups, basis


@pytest.mark.gpu
@pytest.mark.parametrize('kind', [0, 1])
@pytest.mark.parametrize('dense', [False, True])
def test_all_groups_in_one_launch_match_the_oracle(kind, dense):
  """ms_optim_step_groups: eight groups (rows of 3 / 3 / 4 / 1 / 48 / 48 / 3 / 12 floats; scalar, vector and local_vector;
  16-byte pieces where the row length allows) in one launch against the restated host logic, group by group — with a
  sparse index list and in the dense mode (row i = point i, negative weights skip)."""
  from taichi_splatting_amd.optim.fractional import Group, fused_update_groups
  idx, weight, tw, groups, basis = _five_group_case(31 + kind)
  n, mc = tw.shape[0], idx.shape[0]
  grad_scale = torch.rand(mc) + 0.5
  want = {}
  for name, g in groups.items():
    p_o, m_o, v_o = g['param'].clone(), g['m'].clone(), g['v'].clone()
    oopt.group_update(kind, g['type'], p_o, g['grad'], m_o, v_o, idx, weight, tw, 0.02, (0.9, 0.99), 1e-16, True,
                      grad_scale=grad_scale, basis=basis if g['type'] == 'local_vector' else None, clip=0.9)
    want[name] = (p_o, m_o, v_o)
  dev = 'cuda:0'
  if dense:
    w_full, gs_full, basis_full = torch.full((n,), -1.0), torch.zeros(n), torch.zeros(n, 3, 3)
    w_full[idx], gs_full[idx], basis_full[idx] = weight, grad_scale, basis
    args = (w_full.to(dev), None, tw.to(dev))
    extra = dict(basis=basis_full.to(dev), grad_scale=gs_full.to(dev))
  else:
    args = (weight.to(dev), idx.to(dev), tw.to(dev))
    extra = dict(basis=basis.to(dev), grad_scale=grad_scale.to(dev))
  states, objs = {}, []
  for name, g in groups.items():
    states[name] = {'v': g['m'].to(dev), 'm': g['v'].to(dev)}      # (the reference's naming quirk: 'v' is the first moment)
    objs.append(Group(name=name, type=g['type'], param=g['param'].to(dev), grad=g['grad'].to(dev), state=states[name], lr=0.02,
                      betas=(0.9, 0.99), eps=1e-16, bias_correction=True, clip=0.9, mask_lr=None, point_lr=None))
  fused_update_groups(objs, args[0], args[1], args[2], kind, **extra)
  untouched = torch.ones(n, dtype=torch.bool); untouched[idx] = False
  for obj in objs:
    p_o, m_o, v_o = want[obj.name]
    assert torch.allclose(obj.param.cpu(), p_o, rtol=3e-4, atol=2e-6), (obj.name, (obj.param.cpu() - p_o).abs().max())
    assert torch.allclose(states[obj.name]['v'].cpu(), m_o, rtol=3e-4, atol=2e-6), obj.name
    assert torch.allclose(states[obj.name]['m'].cpu(), v_o, rtol=3e-4, atol=2e-6), obj.name
    assert torch.equal(obj.param.cpu()[untouched], groups[obj.name]['param'][untouched]), obj.name


@pytest.mark.gpu
def test_visibility_weights_kernel_matches_the_oracle_and_dense_mode_skips():
  from taichi_splatting_amd import _lib
  lib = _lib.load()
  dev = 'cuda:0'
  torch.manual_seed(3)
  n = 5000
  vis_full = torch.rand(n) * (torch.rand(n) > 0.4)                 # 40 % invisible
  idx = (vis_full > 1e-8).nonzero().squeeze(1)
  running, tw = torch.rand(n), torch.rand(n) * 3
  r_o, tw_o = running.clone(), tw.clone()
  w_o, gs_o = oopt.visibility_weights(r_o, vis_full[idx], idx, tw_o, 0.8, 0.1)
  stream = _lib.current_stream(torch.device(dev))
  # sparse
  r_g, tw_g = running.to(dev), tw.to(dev)
  w_g, gs_g = torch.empty(idx.shape[0], device=dev), torch.empty(idx.shape[0], device=dev)
  idx_g, vis_g, vis_full_g = idx.to(dev), vis_full[idx].to(dev), vis_full.to(dev)      # (alive across the launches)
  _lib.check(lib.ms_optim_visibility_weights(idx_g.data_ptr(), vis_g.data_ptr(), idx.shape[0], 0.8, 0.1,
                                             1e-12, 1e-8, r_g.data_ptr(), tw_g.data_ptr(), w_g.data_ptr(), gs_g.data_ptr(), stream), "w")
  for got, want in ((r_g, r_o), (tw_g, tw_o), (w_g, w_o), (gs_g, gs_o)):
    assert torch.allclose(got.cpu(), want, rtol=2e-6, atol=1e-7), (got.cpu() - want).abs().max()
  # dense: same state, weights at the visible rows, -1 elsewhere, invisible rows untouched
  r_d, tw_d = running.to(dev), tw.to(dev)
  w_d, gs_d = torch.empty(n, device=dev), torch.empty(n, device=dev)
  _lib.check(lib.ms_optim_visibility_weights(None, vis_full_g.data_ptr(), n, 0.8, 0.1, 1e-12, 1e-8, r_d.data_ptr(),
                                             tw_d.data_ptr(), w_d.data_ptr(), gs_d.data_ptr(), stream), "w dense")
  assert torch.equal(r_d, r_g) and torch.equal(tw_d, tw_g)
  assert torch.equal(w_d.cpu()[idx], w_g.cpu())
  hidden = torch.ones(n, dtype=torch.bool); hidden[idx] = False
  assert bool((w_d.cpu()[hidden] == -1).all()) and torch.equal(r_d.cpu()[hidden], running[hidden])


@pytest.mark.gpu
def test_visibility_aware_dense_step_equals_the_indexed_step():
  """``step(None, visibility_of_every_point)`` (no torch.nonzero, no host synchronisation) leaves parameters and
  optimiser state bit for bit where ``step(visible, visibility[visible])`` of the reference's loop leaves them."""
  from taichi_splatting_amd.optim import ParameterClass, VisibilityAwareAdam
  dev = 'cuda:0'
  torch.manual_seed(0)
  n = 4000
  base = dict(position=torch.randn(n, 3), log_scaling=torch.randn(n, 3), rotation=torch.randn(n, 4), alpha_logit=torch.randn(n, 1),
              feature=torch.randn(n, 3, 16))
  groups = dict(position=dict(lr=0.01), log_scaling=dict(lr=0.02, type='vector'), rotation=dict(lr=0.01), alpha_logit=dict(lr=0.05),
                feature=dict(lr=0.02, type='vector'))
  results = []
  for dense in (False, True):
    params = ParameterClass({k: v.clone().to(dev) for k, v in base.items()}, groups, optimizer=VisibilityAwareAdam, vis_beta=0.8,
                            vis_smooth=0.1, betas=(0.9, 0.95))
    torch.manual_seed(1)
    for it in range(4):
      params.zero_grad()
      loss = sum((t ** 2).sum() * (i + 1) for i, t in enumerate(params.tensors.values()))
      loss.backward()
      vis = (torch.rand(n, device=dev) * (torch.rand(n, device=dev) > 0.3)).contiguous()
      if dense:
        params.step(indexes=None, visibility=vis)
      else:
        visible = (vis > 1e-8).nonzero().squeeze(1)
        params.step(indexes=visible, visibility=vis[visible])
    results.append(({k: v.detach().clone() for k, v in params.tensors.items()}, params.tensor_state))
  (pa, sa), (pb, sb) = results
  for k in pa:
    assert torch.equal(pa[k], pb[k]), k
    for key in sa[k]:
      assert torch.equal(sa[k][key], sb[k][key]), (k, key)
