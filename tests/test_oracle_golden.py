"""The oracle restatements of projection / SH / ndc depth and the scene generators are pinned
against fixtures produced by the REFERENCE's own torch_lib (tests/golden/make_fixtures.py).
Tolerances are those of the reference tests: default torch.allclose (rtol 1e-5, atol 1e-8) in f64
(tests/test_projection.py:38-74), atol 1e-5 for SH (tests/util.py:62-63)."""
import pytest
import torch

from oracle import projection as oproj, sh as osh
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians, random_2d_gaussians
from taichi_splatting_amd.rendering import ndc_depth, inverse_ndc_depth
from .conftest import load_golden


def _eval_with_grad(f, *args):
  args = [a.detach().clone().requires_grad_(True) for a in args]
  out = f(*args)
  outs = out if isinstance(out, tuple) else (out,)
  loss = sum(o.mean() for o in outs if o.is_floating_point())
  loss.backward()
  return outs, [a.grad if a.grad is not None else torch.zeros_like(a) for a in args]


@pytest.mark.parametrize('seed', range(5))
@pytest.mark.parametrize('tag', ['f64', 'f32'])
def test_projection_matches_reference(seed, tag):
  fix = load_golden(f'projection_seed{seed}.pt')
  ref = fix[tag]

  def f(*inputs):
    return oproj.apply(*inputs, fix['image_size'], fix['depth_range'], blur_cov=fix['blur_cov'])
  outs, grads = _eval_with_grad(f, *ref['inputs'])
  assert torch.equal(outs[2], ref['indexes'])
  # f32: the eigenvector of a near-isotropic covariance is ill-conditioned (the reference's own f32
  # output differs from its f64 output by ~1e-4 there)
  tol = dict() if tag == 'f64' else dict(rtol=1e-3, atol=1e-3)
  assert torch.allclose(outs[0], ref['points'], **tol)
  assert torch.allclose(outs[1], ref['depth'], **tol)
  if tag == 'f64':
    for g, gr in zip(grads, ref['grads']):
      assert torch.allclose(g, gr, rtol=1e-5, atol=1e-9), (g - gr).abs().max()


@pytest.mark.parametrize('degree', range(4))
def test_sh_matches_reference(degree):
  fix = load_golden(f'sh_deg{degree}.pt')
  outs, grads = _eval_with_grad(lambda p, x, c: osh.evaluate_sh_at(p, x, fix['indexes'], c),
                                fix['params'], fix['points'], fix['camera_pos'])
  assert torch.allclose(outs[0], fix['out'], atol=1e-12)
  for g, gr in zip(grads, fix['grads']):
    assert torch.allclose(g, gr, atol=1e-10), (g - gr).abs().max()


def test_ndc_depth_matches_reference():
  fix = load_golden('ndc.pt')
  assert torch.allclose(oproj.ndc_depth(fix['depth'], fix['near'], fix['far']), fix['ndc'], atol=1e-14)
  assert torch.allclose(ndc_depth(fix['depth'], fix['near'], fix['far']), fix['ndc'], atol=1e-14)
  assert torch.allclose(inverse_ndc_depth(fix['ndc'], fix['near'], fix['far']), fix['inverse'], atol=1e-12)


@pytest.mark.parametrize('seed', range(2))
def test_generators_reproduce_reference_streams(seed):
  fix = load_golden(f'random_data_seed{seed}.pt')
  torch.manual_seed(seed)
  cam = random_camera(image_size=(640, 480)) if seed == 0 else random_camera()
  g3 = random_3d_gaussians(64, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.1 * seed)
  g2 = random_2d_gaussians(64, (320, 200), num_channels=3, scale_factor=0.7, alpha_range=(0.2, 0.8),
                           depth_range=(0.1, 50.0))
  c = fix['camera']
  assert tuple(cam.image_size) == tuple(c['image_size'])
  assert torch.equal(cam.projection, c['projection'])
  assert torch.equal(cam.T_camera_world, c['T_camera_world'])
  assert cam.near_plane == c['near_plane'] and cam.far_plane == c['far_plane']
  for k, v in fix['g3'].items():
    assert torch.allclose(getattr(g3, k), v, rtol=1e-6, atol=1e-7), k
  for k, v in fix['g2'].items():
    assert torch.allclose(getattr(g2, k), v, rtol=1e-6, atol=1e-7), k
