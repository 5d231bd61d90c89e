"""csrc/splat_math.h (the per-element math every gfx950 kernel instantiates) compiled for the host
and checked against the oracle / the reference fixtures: forward projection, the hand-derived
projection and SH backward chains, the pdf gradients and the OBB tile test.  This validates the
derivations on a machine without a GPU; the -m gpu tests then validate the kernels themselves."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import mapper as omap, raster as orast, sh as osh
from .conftest import load_golden

c_dp = ctypes.POINTER(ctypes.c_double)


def dp(a):
  return a.ctypes.data_as(ctypes.c_void_p)


def npd(t):
  return np.ascontiguousarray(t.detach().double().numpy())


@pytest.mark.parametrize('seed', range(5))
def test_projection_forward_backward_vs_reference_fixture(hostmath, seed):
  fix = load_golden(f'projection_seed{seed}.pt')
  ref = fix['f64']
  pos, ls, rot, al, T, P = [npd(t) for t in ref['inputs']]
  n = pos.shape[0]
  W, H = fix['image_size']
  near, far = fix['depth_range']
  points = np.zeros((n, 7)); depth = np.zeros(n); flag = np.zeros(n, dtype=np.int32)
  hostmath.hm_project_fwd(dp(pos), dp(ls), dp(rot), dp(al), dp(T), dp(P), W, H, ctypes.c_double(near),
                          ctypes.c_double(far), ctypes.c_double(fix['blur_cov']), ctypes.c_double(0.15),
                          ctypes.c_double(1 / 255.), ctypes.c_int64(n), dp(points), dp(depth), dp(flag))
  idx = np.nonzero(flag)[0]
  assert np.array_equal(idx, ref['indexes'].numpy())
  assert np.allclose(points[idx], ref['points'].numpy(), rtol=1e-5, atol=1e-8)
  assert np.allclose(depth[idx], ref['depth'].numpy()[:, 0], rtol=1e-5, atol=1e-8)

  # backward under the fixture's loss (sum of means of points and depth over the V visible rows)
  v = idx.shape[0]
  g_points = np.zeros((n, 7)); g_depth = np.zeros(n)
  g_points[idx] = 1.0 / (v * 7)
  g_depth[idx] = 1.0 / v
  d_pos = np.zeros((n, 3)); d_ls = np.zeros((n, 3)); d_rot = np.zeros((n, 4)); d_al = np.zeros(n); d_cam = np.zeros(16)
  # only visible rows contribute: run the backward on those rows
  sel = lambda a: np.ascontiguousarray(a[idx])
  d_pos_v = np.zeros((v, 3)); d_ls_v = np.zeros((v, 3)); d_rot_v = np.zeros((v, 4)); d_al_v = np.zeros(v)
  hostmath.hm_project_bwd(dp(sel(pos)), dp(sel(ls)), dp(sel(rot)), dp(sel(al)), dp(T), dp(P), W, H,
                          ctypes.c_double(fix['blur_cov']), ctypes.c_double(0.15), ctypes.c_int64(v),
                          dp(sel(g_points)), dp(sel(g_depth)), dp(d_pos_v), dp(d_ls_v), dp(d_rot_v), dp(d_al_v), dp(d_cam))
  d_pos[idx], d_ls[idx], d_rot[idx], d_al[idx] = d_pos_v, d_ls_v, d_rot_v, d_al_v
  r_pos, r_ls, r_rot, r_al, r_T, r_P = [g.numpy() for g in ref['grads']]
  for name, got, want in (('position', d_pos, r_pos), ('log_scaling', d_ls, r_ls), ('rotation', d_rot, r_rot),
                          ('alpha_logit', d_al, r_al[:, 0]), ('projection', d_cam[12:], r_P),
                          ('T_camera_world', d_cam[:12].reshape(3, 4), r_T[:3])):
    assert np.allclose(got, want, rtol=1e-5, atol=1e-9), (name, np.abs(got - want).max())
  assert np.all(r_T[3] == 0)


@pytest.mark.parametrize('degree', range(4))
def test_sh_forward_backward_vs_reference_fixture(hostmath, degree):
  fix = load_golden(f'sh_deg{degree}.pt')
  params, points, cam = npd(fix['params']), npd(fix['points']), npd(fix['camera_pos'])
  indexes = np.ascontiguousarray(fix['indexes'].numpy().astype(np.int64))
  v, f = indexes.shape[0], params.shape[1]
  out = np.zeros((v, f))
  g_out = np.full((v, f), 1.0 / (v * f))
  g_params = np.zeros_like(params); g_pos = np.zeros_like(points); g_cam = np.zeros(3)
  hostmath.hm_sh(dp(params), dp(points), dp(indexes), dp(cam), ctypes.c_int64(v), f, degree, dp(g_out), dp(out),
                 dp(g_params), dp(g_pos), dp(g_cam))
  assert np.allclose(out, fix['out'].numpy(), atol=1e-12)
  for got, want in zip((g_params, g_pos, g_cam), fix['grads']):
    assert np.allclose(got, want.numpy(), atol=1e-10), np.abs(got - want.numpy()).max()


@pytest.mark.parametrize('antialias', [0, 1])
def test_pdf_gradients_vs_autograd(hostmath, antialias):
  torch.manual_seed(antialias)
  n = 500
  pix = (torch.rand(n, 2, dtype=torch.float64) * 8).requires_grad_(False)
  mean = torch.rand(n, 2, dtype=torch.float64) * 8
  axis = torch.nn.functional.normalize(torch.randn(n, 2, dtype=torch.float64), dim=1)
  sigma = torch.rand(n, 2, dtype=torch.float64) * 3 + 0.3
  g = torch.cat([mean, axis, sigma, torch.ones(n, 1, dtype=torch.float64)], dim=1).requires_grad_(True)
  # oracle pdf on matched (pixel i, gaussian i) pairs = diagonal of the (P, S) matrix
  p = torch.stack([orast.pdf(pix[i:i + 1], g[i:i + 1], bool(antialias))[0, 0] for i in range(n)])
  p.sum().backward()
  g6 = np.ascontiguousarray(g.detach().numpy()[:, :6])
  out_p = np.zeros(n); dmean = np.zeros((n, 2)); daxis = np.zeros((n, 2)); dsigma = np.zeros((n, 2)); plain = np.zeros(n)
  hostmath.hm_pdf(dp(npd(pix)), dp(g6), ctypes.c_int64(n), antialias, dp(out_p), dp(dmean), dp(daxis), dp(dsigma), dp(plain))
  assert np.allclose(out_p, p.detach().numpy(), atol=1e-13)
  assert np.allclose(plain, out_p, atol=1e-13)
  grad = g.grad.numpy()
  assert np.allclose(dmean, grad[:, 0:2], atol=1e-11)
  assert np.allclose(daxis, grad[:, 2:4], atol=1e-11)
  assert np.allclose(dsigma, grad[:, 4:6], atol=1e-11)


@pytest.mark.parametrize('tile', [8, 16, 32])
def test_obb_query_matches_numpy_oracle_exactly(hostmath, tile):
  from taichi_splatting_amd.testing import random_2d_gaussians
  from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
  torch.manual_seed(tile)
  size = (333, 210)
  g = random_2d_gaussians(5000, size, scale_factor=2.0, alpha_range=(0.001, 1.0))
  p = np.ascontiguousarray(project_gaussians2d(g).numpy().astype(np.float32))
  w_pad, h_pad = omap.pad_to_tile(size, tile)
  counts = np.zeros(p.shape[0], dtype=np.int32); spans = np.zeros((p.shape[0], 4), dtype=np.int32)
  hostmath.hm_tile_count(dp(p), ctypes.c_int64(p.shape[0]), w_pad, h_pad, tile, ctypes.c_float(1 / 255.), dp(counts), dp(spans))
  _, _, want = omap.map_to_tiles(p, g.depths.numpy(), size, tile)
  assert np.array_equal(counts, want)
