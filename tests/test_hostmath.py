"""csrc/splat_math.h (the per-element math every gfx950 kernel instantiates) compiled for the host
and checked against the oracle / the reference fixtures: forward projection, the hand-derived
projection and SH backward chains, the pdf gradients and the OBB tile test.  This validates the
derivations on a machine without a GPU; the -m gpu tests then validate the kernels themselves."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import mapper as omap, raster as orast, sh as osh
from .conftest import covariance_of, load_golden

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


@pytest.mark.parametrize('seed', [1, 2, 3])
def test_projection_backward_f32_conditioning(hostmath, seed):
  """The float32 instantiation of ``project_backward`` against the float64 oracle with random upstream gradients
  (protocol of tests/test_gpu_projection_sh.py::test_projection_f32_backward_vs_oracle, no GPU needed): the
  closed-form eigen-pair derivative keeps EVERY gaussian within 1e-4 of the largest gradient, where the reference's
  chain evaluated in float32 (the torch oracle in float32) has rows off by 1e-2 ... 1e+1."""
  from oracle import projection as oproj
  from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
  torch.manual_seed(seed)
  camera = random_camera()
  n = 4000
  g = random_3d_gaussians(n=n, camera_params=camera, margin=0.5, scale_factor=0.1 if seed % 2 else 1.0)
  in32 = [t.float() for t in g.shape_tensors()] + [camera.T_camera_world.float(), camera.projection.float()]

  def run(args, gp=None, gd=None):
    args = [a.detach().clone().requires_grad_(True) for a in args]
    points, depth, idx = oproj.apply(*args, camera.image_size, camera.depth_range, blur_cov=0.3)
    if gp is None:
      torch.manual_seed(100 + seed)
      gp, gd = torch.randn(points.shape, dtype=torch.float64), torch.randn(depth.shape, dtype=torch.float64)
    torch.autograd.backward([points, depth], [gp.to(points), gd.to(depth)])
    return idx, [a.grad for a in args], gp, gd
  idx64, g64, gp, gd = run([t.double() for t in in32])
  idx32, g32, _, _ = run(in32, gp, gd)
  assert torch.equal(idx64, idx32)
  idx = idx64.numpy()
  v = len(idx)
  pos, ls, rot, al, T, P = [npd(t) for t in in32]
  sel = lambda a: np.ascontiguousarray(a[idx])
  W, H = camera.image_size
  d_pos = np.zeros((v, 3)); d_ls = np.zeros((v, 3)); d_rot = np.zeros((v, 4)); d_al = np.zeros(v); d_cam = np.zeros(16)
  hostmath.hm_project_bwd_f32(dp(sel(pos)), dp(sel(ls)), dp(sel(rot)), dp(sel(al[:, 0])), dp(T), dp(P), int(W), int(H),
                              ctypes.c_double(0.3), ctypes.c_double(0.15), ctypes.c_int64(v), dp(npd(gp)),
                              dp(np.ascontiguousarray(npd(gd)[:, 0])), dp(d_pos), dp(d_ls), dp(d_rot), dp(d_al), dp(d_cam))
  broke = False
  for k, (name, got) in enumerate((('position', d_pos), ('log_scaling', d_ls), ('rotation', d_rot))):
    want = g64[k].numpy()[idx]
    scale = np.abs(want).max()
    err = np.abs(got - want).max(axis=1) / scale
    ref = np.abs(g32[k].double().numpy()[idx] - want).max(axis=1) / scale
    assert err.max() <= 1e-4, (name, err.max())
    assert np.median(err) < 1e-6, (name, np.median(err))
    broke = broke or ref.max() > 1e-3
  assert broke, "the scene should contain rows on which the reference chain loses its digits in float32"


@pytest.mark.parametrize('seed', range(12))
def test_projection_forward_f32_covariance(hostmath, seed):
  """The float32 instantiation of ``project_forward`` against the float64 oracle (protocol of
  /root/reference tests/test_projection.py:50-74, 1..10000 gaussians): mean, sigma, alpha, depth within 1e-4 and the
  covariance REBUILT from (axis, sigma) within 1e-4 of its largest entry on every common row.  torch_lib's own
  formula chain evaluated in float32 (the torch oracle in float32) misses that on the rows whose axis is nearly
  vertical (a < c, b ~ 0: it normalises (a - l2, b) with a - l2 cancelled to a few bits) by up to 2.6e-3."""
  from oracle import projection as oproj
  from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
  torch.manual_seed(seed)
  camera = random_camera()
  n = int(torch.randint(1, 10000, (1,)))
  g = random_3d_gaussians(n=n, camera_params=camera, margin=0.5, scale_factor=0.1 if seed % 2 else 1.0)
  in32 = [t.float() for t in g.shape_tensors()] + [camera.T_camera_world.float(), camera.projection.float()]
  with torch.no_grad():
    p64, d64, i64 = oproj.apply(*[t.double() for t in in32], camera.image_size, camera.depth_range, blur_cov=0.3)
    p32, d32, i32 = oproj.apply(*in32, camera.image_size, camera.depth_range, blur_cov=0.3)
  pos, ls, rot, al, T, P = [npd(t) for t in in32]
  W, H = camera.image_size
  near, far = camera.depth_range
  points = np.zeros((n, 7)); depth = np.zeros(n); flag = np.zeros(n, dtype=np.int32)
  hostmath.hm_project_fwd_f32(dp(pos), dp(ls), dp(rot), dp(np.ascontiguousarray(al[:, 0])), dp(T), dp(P), int(W), int(H),
                              ctypes.c_double(near), ctypes.c_double(far), ctypes.c_double(0.3), ctypes.c_double(0.15),
                              ctypes.c_double(1 / 255.), ctypes.c_int64(n), dp(points), dp(depth), dp(flag))
  seen = np.nonzero(flag)[0]
  assert len(set(seen.tolist()) ^ set(i64.tolist())) <= max(1, n // 2000)
  row64 = {int(k): j for j, k in enumerate(i64.tolist())}
  common = np.array([k for k in seen.tolist() if k in row64], dtype=np.int64)
  if len(common) == 0:
    return
  want = p64[[row64[int(k)] for k in common]]
  got = torch.from_numpy(points[common])
  assert torch.allclose(got[:, [0, 1, 4, 5, 6]], want[:, [0, 1, 4, 5, 6]], rtol=1e-4, atol=1e-3)
  c_got, c_want = covariance_of(got), covariance_of(want)
  rel = (c_got - c_want).abs().max(dim=1).values / c_want.abs().max(dim=1).values
  assert rel.max().item() <= 1e-4, (rel.max().item(), int(rel.argmax()))
  assert rel.median().item() < 1e-6


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
