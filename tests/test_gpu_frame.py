"""-m gpu: the frame executor (csrc/frame.hip, taichi_splatting_amd/frame.py) against the modular composition of the
same kernels (project_to_image -> evaluate_sh_at -> map_to_tiles -> rasterize_with_tiles, the reference's own
structure, renderer.py:23-108).  Same kernels, same tile order => images are bit-identical; gradients agree to
float-atomic noise.  Covers what the executor changes: no compaction (culled gaussians in place), device-side overlap
count with a capacity (growth, overflow), the fused per-gaussian backward, lazily built ``points`` and handed-out
2D-boundary gradients, HIP-graph capture."""
from dataclasses import replace

import pytest
import torch

from taichi_splatting_amd import RasterConfig, frame, render_gaussians
from taichi_splatting_amd.renderer import viewspace_gradient
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def make_scene(n, size, seed, sh_degree=None, dtype=torch.float32, margin=0.1, channels=3):
  torch.manual_seed(seed)
  cam = random_camera(image_size=size)
  g = random_3d_gaussians(n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=margin)
  if sh_degree is not None:
    g = g.replace(feature=(torch.rand(n, channels, (sh_degree + 1) ** 2) - 0.5) * 0.5)
  elif channels != 3:
    g = g.replace(feature=torch.rand(n, channels))
  return g.to(dtype=dtype).to(DEV), cam.to(dtype=dtype).to(device=DEV)


def render_both(g, cam, cfg, use_sh, loss=None, **kw):
  """(frame rendering, leaf grads), (legacy rendering, leaf grads) for the same inputs"""
  out = []
  for use_frame in (True, False):
    frame.USE_FRAME = use_frame
    try:
      gd = g.clone().requires_grad_(True)
      r = render_gaussians(gd, cam, cfg, use_sh=use_sh, **kw)
      grads = None
      if loss is not None:
        loss(r).backward()
        grads = [t.grad for t in (gd.position, gd.log_scaling, gd.rotation, gd.alpha_logit, gd.feature)]
      out.append((r, grads))
    finally:
      frame.USE_FRAME = True
  return out


def assert_grads_close(a, b, tol=2e-5, modular_conditioning=False):
  """Both paths sum the raster backward with float atomics (arrival order differs run to run, ~1e-5 of the largest
  2D gradient): 99.9 % of the entries within `tol` of the largest one, every entry within 50 x tol.

  ``modular_conditioning``: `b` comes from the MODULAR float32 composition — rasterizer -> (d axis, d sigma) ->
  projection backward, the reference's structure and API.  Its axis gradient is a small component perpendicular to
  the axis next to a large parallel one that the normalisation removes afterwards, so the perpendicular part carries
  eps * |parallel| / |perpendicular| of relative error into the 3D gradients of a few gaussians (nearly isotropic
  covariances, or Sxy << Sxx).  The frame executor hands the projection backward a covariance gradient and has no such
  rows (tests/test_gpu_configs.py holds it to the float64 truth on EVERY row); here the leaves behind the projection
  backward are therefore compared by quantiles."""
  for name, x, y in zip(('position', 'log_scaling', 'rotation', 'alpha_logit', 'feature'), a, b):
    scale = max(float(y.abs().max()), 1e-12)
    err = ((x - y).abs() / scale).flatten()
    worst = float(err.max())
    if err.numel() <= 1000:
      assert worst < tol, f"{name}: frame vs modular gradient differs by {worst:.3e} of the largest entry"
      continue
    quantile = lambda f: float(err.float().kthvalue(int(err.numel() * f))[0])
    if modular_conditioning and name in ('position', 'log_scaling', 'rotation'):
      # no cap on the worst row: on an (almost) exactly isotropic splat the modular float32 chain returns anything,
      # like the reference's own float32 kernels
      assert quantile(0.99) < tol, f"{name}: frame vs modular gradient: 99 % quantile {quantile(0.99):.3e} of the largest entry"
      assert quantile(0.999) < 50 * tol, f"{name}: frame vs modular gradient: 99.9 % quantile {quantile(0.999):.3e}"
    else:
      assert quantile(0.999) < tol, f"{name}: frame vs modular gradient: 99.9 % quantile {quantile(0.999):.3e} of the largest entry"
      assert worst < 50 * tol, f"{name}: frame vs modular gradient differs by {worst:.3e} of the largest entry"


@pytest.mark.parametrize('use_sh,degree', [(False, None), (True, 0), (True, 3)])
@pytest.mark.parametrize('tile', [8, 16, 32])
def test_frame_equals_modular_f32(use_sh, degree, tile):
  g, cam = make_scene(20000, (320, 200), seed=tile + (degree or 0), sh_degree=degree)
  cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2))
  torch.manual_seed(0)
  G = torch.randn(200, 320, 3, device=DEV)
  (rf, gf), (rl, gl) = render_both(g, cam, cfg, use_sh, loss=lambda r: (r.image * G).sum())
  # float32 RGB degree 3: the frame's SH kernel (sh_fwd_rows_deg3_kernel) sums a colour's 16 terms as 4 x 4, the
  # modular one in a row — colours agree to rounding, everything that does not depend on them bit for bit
  same = torch.equal if degree != 3 else (lambda a, b: bool(torch.allclose(a, b, rtol=0, atol=2e-6)))
  assert same(rf.image, rl.image)
  assert torch.equal(rf.image_weight, rl.image_weight)
  assert_grads_close(gf, gl, modular_conditioning=True)
  # the lazily compacted points are the modular path's
  assert torch.equal(rf.points.idx, rl.points.idx)
  assert torch.equal(rf.points.gaussians2d, rl.points.gaussians2d)
  assert torch.equal(rf.points.depths, rl.points.depths)
  assert same(rf.points.features, rl.points.features)


def test_frame_with_culled_gaussians_and_camera_grads():
  # a third of the gaussians behind / beside the camera: they stay in place, draw nothing, get zero gradients
  g, cam = make_scene(30000, (256, 256), seed=5, sh_degree=2, margin=0.6)
  cfg = RasterConfig()
  cam_f = replace(cam, T_camera_world=cam.T_camera_world.clone().requires_grad_(True), projection=cam.projection.clone().requires_grad_(True))
  cam_l = replace(cam, T_camera_world=cam.T_camera_world.clone().requires_grad_(True), projection=cam.projection.clone().requires_grad_(True))
  res = []
  for use_frame, c in ((True, cam_f), (False, cam_l)):
    frame.USE_FRAME = use_frame
    try:
      gd = g.clone().requires_grad_(True)
      r = render_gaussians(gd, c, cfg, use_sh=True)
      (r.image.sum() + 0.01 * r.points.depths.sum() + 0.1 * r.points.gaussians2d[:, 4:6].sum()).backward()
      res.append((r, [gd.position.grad, gd.log_scaling.grad, gd.rotation.grad, gd.alpha_logit.grad, gd.feature.grad],
                  c.T_camera_world.grad, c.projection.grad))
    finally:
      frame.USE_FRAME = True
  (rf, gf, tf, pf), (rl, gl, tl, pl) = res
  v = rf.points.idx.shape[0]
  assert 0 < v < 30000, v
  # (the modular path inverts a camera matrix that requires grad with torch.inverse, the frame with its own kernel:
  # the camera position, hence the SH colours, may differ in the last bit)
  assert torch.allclose(rf.image, rl.image, atol=1e-6)
  assert torch.equal(rf.points.idx, rl.points.idx)
  assert_grads_close(gf, gl, modular_conditioning=True)
  culled = torch.ones(30000, dtype=torch.bool, device=DEV)
  culled[rf.points.idx] = False
  for t in gf:
    assert float(t[culled].abs().max()) == 0.0
  # camera gradients: sums over all gaussians, the modular path's ill-conditioned rows included
  assert torch.allclose(tf, tl, rtol=1e-2, atol=5e-3 * float(tl.abs().max()))
  assert torch.allclose(pf, pl, rtol=1e-2, atol=5e-3 * float(pl.abs().max()))


@pytest.mark.parametrize('n', [1, 63, 64 * 37 + 13, 20000])
def test_frame_sh_degree3_rows_kernel(n):
  """sh_fwd_rows_deg3_kernel (coalesced block loads, quad sums) against the modular evaluate_sh_at on the visible set:
  partial last block, culled gaussians in the middle of a block (their rows are skipped, their colours unused)"""
  g, cam = make_scene(n, (256, 256), seed=n % 97, sh_degree=3, margin=0.6)
  (rf, _), (rl, _) = render_both(g, cam, RasterConfig(), True)
  assert torch.equal(rf.points.idx, rl.points.idx)
  assert rf.points.features.shape == rl.points.features.shape
  if rl.points.features.numel():
    assert float((rf.points.features.detach() - rl.points.features.detach()).abs().max()) < 1e-6
  assert torch.allclose(rf.image, rl.image, rtol=0, atol=2e-6)


@pytest.mark.parametrize('antialias', [False, True])
def test_frame_equals_modular_f64(antialias):
  g, cam = make_scene(3000, (160, 96), seed=11, sh_degree=1, dtype=torch.float64, margin=0.3)
  cfg = RasterConfig(antialias=antialias, blur_cov=0.0 if antialias else 0.3)
  torch.manual_seed(1)
  G = torch.randn(96, 160, 3, device=DEV, dtype=torch.float64)
  (rf, gf), (rl, gl) = render_both(g, cam, cfg, True, loss=lambda r: (r.image * G).sum())
  assert torch.equal(rf.image, rl.image)
  assert_grads_close(gf, gl, tol=1e-9)


@pytest.mark.parametrize('channels', [1, 4])
def test_frame_other_channel_counts(channels):
  g, cam = make_scene(5000, (128, 128), seed=2, channels=channels)
  cfg = RasterConfig()
  (rf, gf), (rl, gl) = render_both(g, cam, cfg, False, loss=lambda r: (r.image ** 2).sum())
  assert torch.equal(rf.image, rl.image)
  assert_grads_close(gf, gl)


def test_frame_visibility_heuristics_median_depth16():
  g, cam = make_scene(15000, (256, 192), seed=9, sh_degree=3)
  cfg = RasterConfig(compute_visibility=True, compute_point_heuristic=True)
  (rf, gf), (rl, gl) = render_both(g, cam, cfg, True, loss=lambda r: r.image.sum(), render_median_depth=True, use_depth16=True)
  assert torch.allclose(rf.image, rl.image, rtol=0, atol=2e-6)        # degree-3 colours: equal to rounding (see above)
  assert torch.equal(rf.median_depth_image, rl.median_depth_image)
  assert torch.allclose(rf.points.visibility, rl.points.visibility, rtol=1e-4, atol=1e-5)
  for a, b in ((rf.points.prune_cost, rl.points.prune_cost), (rf.points.split_score, rl.points.split_score)):
    assert float((a - b).abs().max()) < 1e-4 * float(b.abs().max())
  assert_grads_close(gf, gl, modular_conditioning=True)


@pytest.mark.parametrize('margin', [0.0, 0.6])
def test_frame_hands_out_retained_boundary_gradients(margin):
  # reference trainers: rendering.points.gaussians2d.retain_grad(); ...; viewspace_gradient(gaussians2d)
  g, cam = make_scene(12000, (200, 160), seed=4, sh_degree=3, margin=margin)
  cfg = RasterConfig()
  got = []
  for use_frame in (True, False):
    frame.USE_FRAME = use_frame
    try:
      gd = g.clone().requires_grad_(True)
      r = render_gaussians(gd, cam, cfg, use_sh=True)
      r.points.gaussians2d.retain_grad()
      r.points.features.retain_grad()
      (r.image.sum() + r.points.gaussians2d[:, 0].sum()).backward()
      got.append((viewspace_gradient(r.points.gaussians2d), r.points.gaussians2d.grad, r.points.features.grad))
    finally:
      frame.USE_FRAME = True
  for a, b in zip(*got):
    assert a.shape == b.shape
    assert float((a - b).abs().max()) < 2e-5 * float(b.abs().max())


def test_capacity_growth_and_overflow_flag():
  g, cam = make_scene(20000, (256, 256), seed=3, sh_degree=0)
  cfg = RasterConfig()
  frame.USE_FRAME = False
  ref = render_gaussians(g, cam, cfg, use_sh=True).image
  frame.USE_FRAME = True
  # a capacity far too small: the eager path notices after the forward is enqueued and re-runs the emission
  frame._k_capacity[frame._shape_key(torch.device(DEV), 20000, (256, 256), cfg, None, False)] = 1000
  r = render_gaussians(g, cam, cfg, use_sh=True)
  assert torch.equal(r.image, ref)
  st = frame.frame_status(r)
  assert not st['overflow'] and st['capacity'] >= st['overlaps'] > 1000
  # and the frame after it runs at the grown capacity straight away
  r2 = render_gaussians(g, cam, cfg, use_sh=True)
  assert torch.equal(r2.image, ref) and frame.frame_status(r2)['capacity'] == st['capacity']


def test_two_frames_in_flight_share_nothing():
  # loss over two views before one backward: each frame keeps its own saved state
  g, cam = make_scene(8000, (160, 128), seed=6, sh_degree=1)
  torch.manual_seed(8)
  cam2 = random_camera(image_size=(160, 128)).to(device=DEV)
  cfg = RasterConfig()
  grads = []
  for use_frame in (True, False):
    frame.USE_FRAME = use_frame
    try:
      gd = g.clone().requires_grad_(True)
      a = render_gaussians(gd, cam, cfg, use_sh=True).image
      b = render_gaussians(gd, cam2, cfg, use_sh=True).image
      (a.sum() + 2.0 * b.sum()).backward()
      grads.append([gd.position.grad, gd.log_scaling.grad, gd.rotation.grad, gd.alpha_logit.grad, gd.feature.grad])
    finally:
      frame.USE_FRAME = True
  assert_grads_close(*grads, modular_conditioning=True)


def test_empty_and_all_culled_scenes():
  g, cam = make_scene(100, (64, 64), seed=1, sh_degree=0)
  cfg = RasterConfig()
  r = render_gaussians(g[:0], cam, cfg, use_sh=True)
  assert r.image.shape == (64, 64, 3) and float(r.image.abs().max()) == 0.0 and len(r.points) == 0
  far = g.replace(position=g.position + 1e6)          # nothing in view
  fd = far.clone().requires_grad_(True)
  r = render_gaussians(fd, cam, cfg, use_sh=True)
  r.image.sum().backward()
  assert float(r.image.detach().abs().max()) == 0.0 and len(r.points) == 0
  assert float(fd.position.grad.abs().max()) == 0.0 and float(fd.feature.grad.abs().max()) == 0.0


def test_frame_under_hip_graph_capture():
  g, cam = make_scene(20000, (256, 256), seed=7, sh_degree=3)
  cfg = RasterConfig()
  gd = g.clone().requires_grad_(True)
  leaves = [gd.position, gd.log_scaling, gd.rotation, gd.alpha_logit, gd.feature]

  def step():
    for t in leaves:
      t.grad = None
    r = render_gaussians(gd, cam, cfg, use_sh=True)
    r.image.sum().backward()
    return r

  ref = step()
  ref_image, ref_grads = ref.image.detach().clone(), [t.grad.clone() for t in leaves]
  del ref        # an autograd graph built on the default stream must not outlive into the capture (torch's rule)
  graph = frame.FrameGraph(step, warmup=2)
  before = frame.host_syncs
  for _ in range(3):
    r = graph.replay()
  assert frame.host_syncs == before, "a captured frame must not wait on the host"
  torch.cuda.synchronize()
  assert torch.equal(r.image, ref_image)
  assert_grads_close([t.grad for t in leaves], ref_grads)
  # a new camera pose written into the captured tensors is picked up by the next replay
  with torch.no_grad():
    cam.T_camera_world[:3, 3] += 0.05
  r = graph.replay()
  torch.cuda.synchronize()
  frame.USE_FRAME = False
  try:
    want = render_gaussians(g, cam, cfg, use_sh=True).image
  finally:
    frame.USE_FRAME = True
  assert torch.allclose(r.image, want, rtol=0, atol=2e-6)      # degree-3 colours of the two SH kernels: equal to rounding
  assert not frame.frame_status(r)['overflow']


@pytest.mark.parametrize('dtype,channels,tile', [(torch.float32, 3, 16), (torch.float32, 1, 8), (torch.float64, 3, 16), (torch.float32, 3, 32)])
def test_rasterize_2d_on_the_executor_equals_modular(dtype, channels, tile):
  # rasterize() (reference rasterizer/function.py:133-165) = map_to_tiles + rasterize_with_tiles; on the executor's
  # projected-input mode it is one node without a host read of the overlap total
  from taichi_splatting_amd import rasterize
  from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
  from taichi_splatting_amd.testing import random_2d_gaussians
  torch.manual_seed(3)
  size = (200, 144)
  g = random_2d_gaussians(6000, size, num_channels=channels, scale_factor=2.0, alpha_range=(0.2, 0.9)).to(DEV)
  p, f, d = project_gaussians2d(g).to(dtype), g.feature.to(dtype).contiguous(), g.depths
  cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2), compute_visibility=True,
                     compute_point_heuristic=True)
  G = torch.randn(size[1], size[0], channels, device=DEV, dtype=dtype)
  res = []
  for use_frame in (True, False):
    frame.USE_FRAME = use_frame
    try:
      pg, fg = p.clone().requires_grad_(True), f.clone().requires_grad_(True)
      out = rasterize(pg, d, fg, size, cfg)
      (out.image * G).sum().backward()
      res.append((out, pg.grad, fg.grad))
    finally:
      frame.USE_FRAME = True
  (a, gpa, gfa), (b, gpb, gfb) = res
  assert torch.equal(a.image, b.image) and torch.equal(a.image_weight, b.image_weight)
  tol = 1e-9 if dtype == torch.float64 else 5e-5
  assert torch.allclose(a.visibility, b.visibility, rtol=1e-4, atol=1e-5)
  for x, y in ((gpa, gpb), (gfa, gfb), (a.point_heuristic, b.point_heuristic)):
    assert float((x - y).abs().max()) <= tol * float(y.abs().max()), float((x - y).abs().max()) / float(y.abs().max())
