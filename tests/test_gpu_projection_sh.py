"""-m gpu: projection and SH kernels (through the C-ABI) vs the oracle and the reference fixtures.
Tolerances: f64 = the reference's own (torch.allclose defaults, tests/test_projection.py:38-74;
atol 1e-5 for SH, tests/util.py:62-63); f32 = 1e-4 relative except the ill-conditioned axis."""
import pytest
import torch

from oracle import projection as oproj, sh as osh
from taichi_splatting_amd import evaluate_sh_at, RasterConfig, Gaussians3D
from taichi_splatting_amd.perspective import projection as hip_proj, project_to_image
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
from .conftest import covariance_of, load_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def assert_f32_gradient_as_accurate_as_reference(got, w64, w32, what):
  """got: float32 kernel result; w64 / w32: the oracle evaluated in float64 / in float32 (torch CPU) on the same
  float32 inputs.  Errors are per row (per gaussian), relative to the largest float64 gradient.

  torch_lib's own formulas evaluated in float32 miss the float64 value on a fraction of a per cent of the gaussians by
  anything from 1e-3 to 1e+3 times the largest gradient: the derivative of the eigen-decomposition divides by
  l1 - l2 and by |(a - l2, b)|, which cancel for a nearly isotropic blurred covariance
  (tools/diag/proj_f32_grad_error.py).  Rounds 1-3 matched that conditioning; since round 4 ``project_backward``
  evaluates the eigen-pair derivative in closed form (csrc/splat_math.h), so the float32 kernels are held to the
  north_star's 1e-4 itself and must BEAT the reference arithmetic where that one breaks down:
    * at least 99.9 % of the rows within 1e-4 (measured: all of them, largest 4e-5), none beyond 1e-3, and never
      fewer rows within 1e-4 than the float32 oracle;
    * median error below 1e-6; the 99th percentile no more than the float32 oracle's (+ 1e-6);
    * sums over all gaussians (camera gradients): no more than 5x the float32 oracle's error (+ 1e-5)."""
  got, w32 = got.double(), w32.double()
  scale = w64.abs().max().item()
  if got.dim() < 2 or got.shape[0] <= 16:          # camera pose / intrinsics: one sum over all gaussians
    err, ref = (got - w64).abs().max().item() / scale, (w32 - w64).abs().max().item() / scale
    assert err <= 5 * ref + 1e-5, (what, 'sum over gaussians', err, ref)
    return
  rows = lambda t: t.reshape(t.shape[0], -1).abs().max(dim=1).values / scale
  err, ref = rows(got - w64), rows(w32 - w64)
  touched = rows(w64) > 0                            # culled gaussians have zero gradient everywhere
  err, ref = err[touched], ref[touched]
  frac, frac_ref = (err <= 1e-4).double().mean().item(), (ref <= 1e-4).double().mean().item()
  assert frac >= 0.999 and frac >= frac_ref, (what, 'rows within 1e-4', frac, frac_ref)
  assert err.max().item() <= 1e-3, (what, 'worst row', err.max().item(), ref.max().item())
  assert err.median().item() < 1e-6, (what, 'median', err.median().item())
  q99, q99_ref = err.quantile(0.99).item(), ref.quantile(0.99).item()
  assert q99 <= q99_ref + 1e-6, (what, '99th percentile', q99, q99_ref)


def assert_f32_forward_rows(h_out, o64_out):
  """Float32 kernel outputs (points (V, 7), depth (V, 1), indexes) against the FLOAT64 oracle's on the rows both
  keep: mean / sigma / alpha / depth to 1e-4 relative, and — the axis columns, which every comparison of rounds 1-4
  dropped — the covariance rebuilt from (axis, sigma) to 1e-4 of its largest entry on every row (the kernels evaluate
  the eigen-pair without the reference chain's cancellations, csrc/splat_math.h; measured on the host build: 4e-6,
  torch_lib's float32 arithmetic: 2.6e-3) and the axis itself to 1e-3 (measured 3e-5)."""
  idx_h, idx_o = h_out[2].cpu(), o64_out[2]
  row_o = {int(k): j for j, k in enumerate(idx_o.tolist())}
  keep = [j for j, k in enumerate(idx_h.tolist()) if int(k) in row_o]
  if not keep:
    return
  got_p, got_d = h_out[0].cpu()[keep].double(), h_out[1].cpu()[keep].double()
  sel = [row_o[int(idx_h[j])] for j in keep]
  want_p, want_d = o64_out[0][sel].double(), o64_out[1][sel].double()
  assert torch.allclose(got_p[:, [0, 1, 4, 5, 6]], want_p[:, [0, 1, 4, 5, 6]], rtol=1e-4, atol=1e-3)
  assert torch.allclose(got_d, want_d, rtol=1e-4, atol=1e-5)
  c_got, c_want = covariance_of(got_p), covariance_of(want_p)
  rel = (c_got - c_want).abs().max(dim=1).values / c_want.abs().max(dim=1).values
  assert rel.max().item() <= 1e-4, ('covariance from (axis, sigma)', rel.max().item())
  assert (got_p[:, 2:4] - want_p[:, 2:4]).abs().max().item() <= 1e-3, 'axis'


def _eval_with_grad(f, *args):
  args = [a.detach().clone().requires_grad_(True) for a in args]
  out = f(*args)
  outs = out if isinstance(out, tuple) else (out,)
  loss = sum(o.mean() for o in outs if o.is_floating_point())
  loss.backward()
  return [o.detach() for o in outs], [a.grad if a.grad is not None else torch.zeros_like(a) for a in args]


@pytest.mark.parametrize('seed', range(5))
def test_projection_fixture_f64(seed):
  fix = load_golden(f'projection_seed{seed}.pt')
  ref = fix['f64']
  inputs = [t.to(DEV) for t in ref['inputs']]
  outs, grads = _eval_with_grad(
    lambda *a: hip_proj.apply(*a, fix['image_size'], fix['depth_range'], blur_cov=fix['blur_cov']), *inputs)
  assert torch.equal(outs[2].cpu(), ref['indexes'])
  assert outs[2].dtype == torch.int64
  assert torch.allclose(outs[0].cpu(), ref['points'])
  assert torch.allclose(outs[1].cpu(), ref['depth'])
  for name, g, gr in zip(('position', 'log_scaling', 'rotation', 'alpha_logit', 'T_camera_world', 'projection'),
                         grads, ref['grads']):
    assert g.shape == gr.shape, name
    assert torch.allclose(g.cpu(), gr, rtol=1e-5, atol=1e-9), (name, (g.cpu() - gr).abs().max())


@pytest.mark.parametrize('seed', range(100))         # the reference's count (tests/test_projection.py:76,100)
def test_projection_random_vs_oracle(seed):
  # protocol of tests/test_projection.py:22-96 (random camera, 1..10000 points, margin .5)
  torch.manual_seed(seed)
  camera = random_camera()
  n = int(torch.randint(1, 10000, (1,)))
  g = random_3d_gaussians(n=n, camera_params=camera, margin=0.5, scale_factor=0.1)
  for dtype in (torch.float64, torch.float32):
    inputs = [t.to(dtype) for t in g.shape_tensors()] + [camera.T_camera_world.to(dtype), camera.projection.to(dtype)]
    f_o = lambda *a: oproj.apply(*a, camera.image_size, camera.depth_range, blur_cov=0.3)
    f_h = lambda *a: hip_proj.apply(*a, camera.image_size, camera.depth_range, blur_cov=0.3)
    o_out, o_grad = _eval_with_grad(f_o, *inputs)
    h_out, h_grad = _eval_with_grad(f_h, *[t.to(DEV) for t in inputs])
    if dtype == torch.float64:
      o64_out = o_out
      assert torch.equal(h_out[2].cpu(), o_out[2])
      assert torch.allclose(h_out[0].cpu(), o_out[0])
      assert torch.allclose(h_out[1].cpu(), o_out[1])
      for g_h, g_o in zip(h_grad, o_grad):
        assert torch.allclose(g_h.cpu(), g_o, rtol=1e-5, atol=1e-9), (g_h.cpu() - g_o).abs().max()
    else:
      # f32 culling decisions may flip for gaussians numerically on the frustum boundary: the rows BOTH evaluations keep
      # are compared whatever the flips (rounds 1-4 compared nothing on a seed with a flip)
      a, b = set(h_out[2].cpu().tolist()), set(o_out[2].tolist())
      assert len(a ^ b) <= max(1, n // 2000), (len(a ^ b), n)
      assert_f32_forward_rows(h_out, o64_out)


@pytest.mark.parametrize('seed', range(8))
def test_projection_f32_backward_vs_oracle(seed):
  """The float32 kernels (the product path) against the oracle, gradients included.  The reference's formulas are
  ill-conditioned in float32 for some gaussians (eigen-decomposition of a near-isotropic blurred covariance,
  quaternion normalisation): torch_lib's own arithmetic evaluated in float32 misses the float64 gradients by up
  to ~10 % of the largest gradient on such rows, so the float32 kernels are held to
    the criterion of assert_f32_gradient_as_accurate_as_reference: 1e-4 on >= 99.9 % of the gaussians, none beyond
    1e-3, never fewer rows within 1e-4 than torch_lib's own arithmetic reaches in float32."""
  torch.manual_seed(seed)
  camera = random_camera()
  n = 4000
  g = random_3d_gaussians(n=n, camera_params=camera, margin=0.5, scale_factor=0.1 if seed % 2 else 1.0)
  inputs = [t.float() for t in g.shape_tensors()] + [camera.T_camera_world.float(), camera.projection.float()]

  f_o = lambda *a: oproj.apply(*a, camera.image_size, camera.depth_range, blur_cov=0.3)
  f_h = lambda *a: hip_proj.apply(*a, camera.image_size, camera.depth_range, blur_cov=0.3)
  in64, in32, in_h = [t.double() for t in inputs], inputs, [t.to(DEV) for t in inputs]
  # a culling decision may flip between float32 and float64 for a gaussian on the edge of the view: the upstream
  # gradients are drawn per GAUSSIAN and zeroed outside the set all three evaluations agree on, so that every
  # gaussian (and every camera sum) is compared on common ground instead of skipping the seed
  with torch.no_grad():
    sets = [set(f(*a)[2].cpu().tolist()) for f, a in ((f_o, in64), (f_o, in32), (f_h, in_h))]
  common = torch.zeros(n, dtype=torch.bool)
  common[sorted(sets[0] & sets[1] & sets[2])] = True
  assert len(sets[0] ^ sets[2]) <= 4 and int(common.sum()) >= len(sets[0]) - 4      # flips are edge cases, not the rule
  torch.manual_seed(100 + seed)
  gp_all = torch.randn((n, 7), dtype=torch.float64) * common[:, None]
  gd_all = torch.randn((n, 1), dtype=torch.float64) * common[:, None]

  def run(f, args):
    args = [a.detach().clone().requires_grad_(True) for a in args]
    points, depth, idx = f(*args)
    torch.autograd.backward([points, depth], [gp_all.to(points)[idx], gd_all.to(depth)[idx]])
    keep = common.to(idx.device)[idx]
    return (points.detach()[keep], depth.detach()[keep], idx[keep]), [a.grad for a in args]
  o64, g64 = run(f_o, in64)
  o32, g32 = run(f_o, in32)
  oh, gh = run(f_h, in_h)
  assert torch.equal(oh[2].cpu(), o64[2])
  # forward: mean, sigma, alpha, depth to 1e-4 relative, the covariance rebuilt from (axis, sigma) to 1e-4
  assert_f32_forward_rows(oh, o64)
  for name, got, w64, w32 in zip(('position', 'log_scaling', 'rotation', 'alpha_logit', 'T_camera_world', 'projection'), gh, g64, g32):
    assert_f32_gradient_as_accurate_as_reference(got.cpu(), w64, w32, name)


def test_projection_empty_and_all_culled():
  cam = random_camera(image_size=(64, 48))
  cfg = RasterConfig()
  g = Gaussians3D(position=torch.zeros(0, 3), log_scaling=torch.zeros(0, 3), rotation=torch.zeros(0, 4),
                  alpha_logit=torch.zeros(0, 1), feature=torch.zeros(0, 3), batch_size=(0,)).to(DEV)
  p, d, idx = project_to_image(g, cam.to(device=DEV), cfg)
  assert p.shape == (0, 7) and d.shape == (0, 1) and idx.shape == (0,)
  # behind the camera: nothing visible
  T = cam.T_camera_world
  behind = (torch.inverse(T) @ torch.tensor([0., 0., -5., 1.]))[:3]
  g = Gaussians3D(position=behind.repeat(10, 1), log_scaling=torch.zeros(10, 3),
                  rotation=torch.tensor([[0., 0, 0, 1]]).repeat(10, 1), alpha_logit=torch.zeros(10, 1),
                  feature=torch.zeros(10, 3), batch_size=(10,)).to(DEV)
  p, d, idx = project_to_image(g, cam.to(device=DEV), cfg)
  assert p.shape == (0, 7) and idx.shape == (0,)


@pytest.mark.parametrize('degree', range(4))
def test_sh_fixture_f64(degree):
  fix = load_golden(f'sh_deg{degree}.pt')
  idx = fix['indexes'].to(DEV)
  outs, grads = _eval_with_grad(lambda p, x, c: evaluate_sh_at(p, x, idx, c),
                                fix['params'].to(DEV), fix['points'].to(DEV), fix['camera_pos'].to(DEV))
  assert torch.allclose(outs[0].cpu(), fix['out'], atol=1e-12)
  for g, gr in zip(grads, fix['grads']):
    assert torch.allclose(g.cpu(), gr, atol=1e-10), (g.cpu() - gr).abs().max()


@pytest.mark.parametrize('seed', range(100))         # the reference's count (tests/test_spherical_harmonics.py:33,52)
def test_sh_random_f32_vs_oracle(seed):
  # protocol of tests/test_spherical_harmonics.py:16-45 (f32, atol 1e-5, repeated indexes)
  torch.random.manual_seed(seed)
  dimension = int(torch.randint(1, 4, (1,)))
  degree = int(torch.randint(0, 4, (1,)))
  n = int(torch.randint(1, 102, (1,)))
  params = torch.rand(n, dimension, (degree + 1) ** 2)
  points = torch.randn(n, 3)
  camera_pos = torch.randn(3)
  indexes = torch.randint(0, n, (max(n // 2, 1),))
  o_out, o_grad = _eval_with_grad(lambda p, x, c: osh.evaluate_sh_at(p, x, indexes, c), params, points, camera_pos)
  h_out, h_grad = _eval_with_grad(lambda p, x, c: evaluate_sh_at(p, x, indexes.to(DEV), c),
                                  params.to(DEV), points.to(DEV), camera_pos.to(DEV))
  assert torch.allclose(h_out[0].cpu(), o_out[0], atol=1e-5)
  for g_h, g_o in zip(h_grad, o_grad):
    assert torch.allclose(g_h.cpu(), g_o, atol=1e-5), (g_h.cpu() - g_o).abs().max()


@pytest.mark.parametrize('seed,n', [(0, 1), (1, 77), (2, 4096 + 5), (3, 30000)])
def test_frame_sh_degree3_colours_vs_oracle(seed, n):
  """The frame executor's float32 RGB degree-3 SH kernel (sh_fwd_rows_deg3_kernel: coalesced block loads, 4 x 4
  summation) against the oracle in float64 on the visible set, at the reference's tolerance (atol 1e-5,
  tests/util.py:62-63) — not only against the modular kernel"""
  from taichi_splatting_amd import RasterConfig, render_gaussians
  from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
  torch.manual_seed(seed)
  cam = random_camera(image_size=(192, 128))
  g = random_3d_gaussians(n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.5)
  g = g.replace(feature=(torch.rand(n, 3, 16) - 0.5) * 0.8)
  r = render_gaussians(g.to(DEV), cam.to(device=DEV), RasterConfig(), use_sh=True)
  idx = r.points.idx.cpu()
  if idx.numel() == 0:
    return
  cam_pos = torch.inverse(cam.T_camera_world.double())[:3, 3]
  want = osh.evaluate_sh_at(g.feature.double(), g.position.double(), idx, cam_pos)
  got = r.points.features.detach().cpu().double()
  assert got.shape == want.shape
  assert float((got - want).abs().max()) < 1e-5


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
def test_camera_position_kernel_matches_torch_inverse(dtype):
  # perspective/params.py:62-65: inverse(T_camera_world)[0:3, 3]
  from taichi_splatting_amd.testing import random_camera
  for seed in range(5):
    torch.manual_seed(seed)
    cam = random_camera(image_size=(64, 48)).to(device='cuda:0', dtype=dtype)
    want = torch.inverse(cam.T_camera_world.cpu().double())[0:3, 3]
    got = cam.camera_position
    assert got.dtype == dtype and got.is_cuda
    assert torch.allclose(got.cpu().double(), want, rtol=1e-5 if dtype == torch.float32 else 1e-12, atol=1e-6 if dtype == torch.float32 else 1e-12)
  # general (non-rigid) matrix and the autograd fallback
  T = torch.randn(4, 4, dtype=dtype, device='cuda:0') + 3 * torch.eye(4, dtype=dtype, device='cuda:0')
  cam2 = cam.__class__(projection=cam.projection, T_camera_world=T, near_plane=cam.near_plane, far_plane=cam.far_plane,
                       image_size=cam.image_size)
  assert torch.allclose(cam2.camera_position.cpu().double(), torch.inverse(T.cpu().double())[0:3, 3], rtol=1e-4, atol=1e-5)
  Tg = T.clone().requires_grad_(True)
  cam3 = cam.__class__(projection=cam.projection, T_camera_world=Tg, near_plane=cam.near_plane, far_plane=cam.far_plane,
                       image_size=cam.image_size)
  cam3.camera_position.sum().backward()
  assert Tg.grad is not None
