"""-m gpu: host-side guarantees added in round 4.

* frames in flight own their pinned K word and their moments buffer per stream (no cross-talk between two frames
  that have not been settled, e.g. a viewer thread next to a training thread);
* a frame replayed from a HIP graph that overflows its overlap capacity is REPORTED: ``FrameGraph.replay`` raises
  ``frame.FrameOverflow`` at the next replay (immediately with strict=True), ``LazyPoints`` and the rank steps'
  ``poll()`` do the same — never a silent run of background-only frames;
* the rank steps honour the deterministic backward, refuse camera gradients they cannot produce, and accept an
  empty strip.
"""
import threading

import pytest
import torch

from taichi_splatting_amd import RasterConfig, frame, render_gaussians, sharded
from taichi_splatting_amd.rasterizer import function as raster_function
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def make_scene(n, size, seed, sh_degree=0, margin=0.1):
  torch.manual_seed(seed)
  cam = random_camera(image_size=size)
  g = random_3d_gaussians(n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=margin)
  g = g.replace(feature=(torch.rand(n, 3, (sh_degree + 1) ** 2) - 0.5) * 0.5)
  return g.to(DEV), cam.to(device=DEV)


def test_unsettled_frames_keep_their_own_overlap_total():
  """Two _FrameFunction forwards enqueued before either looks at K: round 3 shared ONE pinned word per device, the
  second frame's reset replaced the first frame's K.  Each frame must settle with its own total."""
  cfg = RasterConfig()
  small, cam = make_scene(3000, (192, 128), seed=1)
  big, _ = make_scene(40000, (192, 128), seed=2)
  want = []
  for g in (small, big):
    r = render_gaussians(g, cam, cfg, use_sh=True)
    want.append((frame.frame_status(r)['overlaps'], r.image.clone()))
  assert want[0][0] != want[1][0]

  def enqueue(g):
    opts = frame.FrameOptions(image_size=(192, 128), depth_range=tuple(float(x) for x in cam.depth_range), config=cfg, use_sh=True)
    state = frame.FrameState()
    out = frame._FrameFunction.apply(*g.shape_tensors(), g.feature, cam.T_camera_world.reshape(4, 4),
                                     cam.projection.reshape(4), opts, state)
    return state, out[0]

  s1, image1 = enqueue(small)
  s2, image2 = enqueue(big)
  assert s1.pending is not None and s2.pending is not None, "both frames are still waiting for their K"
  s2.settle()
  s1.settle()
  assert (s1.k, s2.k) == (want[0][0], want[1][0])
  assert torch.equal(image1, want[0][1]) and torch.equal(image2, want[1][1])


def test_frames_from_two_threads_and_streams():
  """a 'viewer' thread rendering on its own stream next to a training loop: images and gradients of both stay right"""
  cfg = RasterConfig()
  g, cam = make_scene(20000, (256, 192), seed=3, sh_degree=1)
  torch.manual_seed(5)
  cam2 = random_camera(image_size=(256, 192)).to(device=DEV)

  def train_once():
    gd = g.clone().requires_grad_(True)
    r = render_gaussians(gd, cam, cfg, use_sh=True)
    r.image.sum().backward()
    return r.image.detach().clone(), gd.position.grad.clone()

  ref_image, ref_grad = train_once()
  ref_view = render_gaussians(g, cam2, cfg, use_sh=True).image.clone()
  torch.cuda.synchronize()
  errors, views = [], []

  def viewer():
    try:
      stream = torch.cuda.Stream()
      with torch.cuda.stream(stream):
        for _ in range(30):
          gd = g.clone().requires_grad_(True)
          r = render_gaussians(gd, cam2, cfg, use_sh=True)
          r.image.mean().backward()           # a backward of its own: its moments buffer is keyed by ITS stream
          views.append(r.image.detach())
      stream.synchronize()
    except Exception as e:      # noqa: BLE001
      errors.append(e)

  t = threading.Thread(target=viewer)
  t.start()
  results = [train_once() for _ in range(30)]
  t.join()
  torch.cuda.synchronize()
  assert not errors, errors
  scale = float(ref_grad.abs().max())
  for image, grad in results:
    assert torch.equal(image, ref_image)
    assert float((grad - ref_grad).abs().max()) < 1e-4 * scale      # float-atomic noise only
  for v in views:
    assert torch.equal(v, ref_view)


def test_moments_cache_keeps_several_sizes_and_pins_captured_buffers():
  frame.release_caches(force=True)
  dev = torch.device(DEV)
  a = frame._moments_buffer(dev, 1000, False)
  b = frame._moments_buffer(dev, 2000, False)
  assert frame._moments_buffer(dev, 1000, False) is a and frame._moments_buffer(dev, 2000, False) is b, \
    "alternating two scene sizes must not reallocate (round 3: one size at a time)"
  for n in (3000, 4000, 5000, 6000):
    frame._moments_buffer(dev, n, False)
  assert len([k for k in frame._moments if k[0] == dev.index]) <= frame.MOMENTS_LRU
  frame.release_caches(force=True)


def _capture_small_step(n_capture, n_replay_scale):
  """captured step whose scene can be swapped in place for one with more overlaps"""
  cfg = RasterConfig()
  g, cam = make_scene(n_capture, (256, 256), seed=7)
  gd = g.clone().requires_grad_(True)
  leaves = [gd.position, gd.log_scaling, gd.rotation, gd.alpha_logit, gd.feature]

  def step():
    for t in leaves:
      t.grad = None
    r = render_gaussians(gd, cam, cfg, use_sh=True)
    r.image.sum().backward()
    return r

  return gd, step


def test_graph_replay_overflow_is_reported_not_silent():
  gd, step = _capture_small_step(20000, 4.0)
  r = step()
  del r
  graph = frame.FrameGraph(step, warmup=2, strict=False)
  graph.replay()
  torch.cuda.synchronize()
  graph.replay()                  # fine so far
  with torch.no_grad():
    gd.log_scaling += 1.2         # every splat 3.3 x larger: far more tile overlaps than the captured buffers hold
  r = graph.replay()              # this replay overflows on the device: background only ...
  torch.cuda.synchronize()
  assert float(r.image.detach().abs().max()) == 0.0
  with pytest.raises(frame.FrameOverflow, match="set_overlap_capacity"):
    graph.replay()                # ... and the NEXT host touch says so, without the caller asking
  graph.replay()                  # reported once; the caller decides (here: keeps going)
  torch.cuda.synchronize()


def test_graph_replay_overflow_strict_raises_in_the_step():
  gd, step = _capture_small_step(20000, 4.0)
  r = step()
  del r
  # nothing may be pinned INSIDE the capture (hipHostMalloc invalidates it): with torch's pinned-memory cache emptied a
  # pin_memory() call during capture would have to go to the driver — the captured frame's K word comes from a block
  # pinned in eager mode (frame.KSlots.graph_words); this failed at 6 M gaussians in bench.py's graph child
  empty = getattr(torch._C, '_host_emptyCache', None)
  if empty is not None:
    empty()
  graph = frame.FrameGraph(step, warmup=2, strict=True)
  graph.replay()
  with torch.no_grad():
    gd.log_scaling += 1.2
  with pytest.raises(frame.FrameOverflow):
    graph.replay()


def test_lazy_points_report_an_overflowed_replay():
  gd, step = _capture_small_step(20000, 4.0)
  r = step()
  del r
  graph = frame.FrameGraph(step, warmup=2, strict=False)
  with torch.no_grad():
    gd.log_scaling += 1.2
  r = graph.replay()
  with pytest.raises(frame.FrameOverflow):
    len(r.points)                 # LazyPoints synchronises on the visible count anyway: the overflow surfaces here


def test_rank_step_polls_overflow_and_strict_mode():
  g, cam = make_scene(6000, (160, 160), seed=11, sh_degree=1)
  cfg = RasterConfig()
  loss_fn = lambda image, rows: image.sum()      # noqa: E731
  mine = g.clone().requires_grad_(True)
  step = sharded.StripStep((160, 160), cfg, cam.depth_range, 0, 1, [0, 10])
  step.probe(mine, cam, True)
  step.strict = False
  step.step(mine, cam, loss_fn, use_sh=True)
  torch.cuda.synchronize()
  step.poll()                       # nothing to report
  step = sharded.StripStep((160, 160), cfg, cam.depth_range, 0, 1, [0, 10])
  step.probe(mine, cam, True)
  step.strict = False
  step.k_capacity = 4096            # a capacity that is too small
  image, _ = step.step(mine, cam, loss_fn, use_sh=True, backward=False)
  torch.cuda.synchronize()
  assert float(image.abs().max()) == 0.0
  with pytest.raises(frame.FrameOverflow, match="background only"):
    step.step(mine, cam, loss_fn, use_sh=True, backward=False)       # reported on entry of the NEXT step, no sync needed
  step.strict = True
  with pytest.raises(frame.FrameOverflow):
    step.step(mine, cam, loss_fn, use_sh=True, backward=False)      # strict: raises for the step itself


def test_rank_step_refuses_camera_gradients_and_honours_deterministic():
  g, cam = make_scene(5000, (128, 128), seed=12, sh_degree=0)
  cfg = RasterConfig()
  loss_fn = lambda image, rows: (image * image).sum()      # noqa: E731
  step = sharded.StripStep((128, 128), cfg, cam.depth_range, 0, 1, [0, 8])
  mine = g.clone().requires_grad_(True)
  step.probe(mine, cam, True)
  cam_grad = cam.__class__(projection=cam.projection, T_camera_world=cam.T_camera_world.clone().requires_grad_(True),
                           near_plane=cam.near_plane, far_plane=cam.far_plane, image_size=cam.image_size)
  with pytest.raises(NotImplementedError, match="camera gradients"):
    step.step(mine, cam_grad, loss_fn, use_sh=True)

  was = raster_function.DETERMINISTIC_BACKWARD
  raster_function.DETERMINISTIC_BACKWARD = True
  try:
    runs = []
    for _ in range(3):
      mine = g.clone().requires_grad_(True)
      step.step(mine, cam, loss_fn, use_sh=True)
      runs.append([t.grad.clone() for t in (mine.position, mine.log_scaling, mine.rotation, mine.alpha_logit, mine.feature)])
    for other in runs[1:]:
      for a, b in zip(runs[0], other):
        assert torch.equal(a, b), "MS_DETERMINISTIC must give bitwise reproducible gradients on the rank steps too"
  finally:
    raster_function.DETERMINISTIC_BACKWARD = was
  # and it agrees with the float-atomic result
  mine = g.clone().requires_grad_(True)
  step.step(mine, cam, loss_fn, use_sh=True)
  scale = float(mine.position.grad.abs().max())
  assert float((mine.position.grad - runs[0][0]).abs().max()) < 2e-4 * scale


def test_empty_strip_is_a_valid_rank():
  g, cam = make_scene(4000, (128, 128), seed=13, sh_degree=0)
  cfg = RasterConfig()
  loss_fn = lambda image, rows: image.sum()      # noqa: E731
  # world of 3 with an empty middle strip (bounds repeat a value), ranks run one after the other on this GPU
  bounds = [0, 5, 5, 8]
  full = render_gaussians(g, cam, cfg, use_sh=True).image
  rows = []
  for rank in range(3):
    step = sharded.StripStep((128, 128), cfg, cam.depth_range, rank, 3, bounds)
    mine = g.clone().requires_grad_(True)
    step.probe(mine, cam, True)
    image, _ = step.step(mine, cam, loss_fn, use_sh=True, backward=False)
    rows.append(image)
  assert rows[1].shape[0] == 0
  assert torch.equal(torch.cat(rows, dim=0), full)
  # the single-process frame with an empty cropped strip
  r = frame.render_frame(g, cam, cfg, True, tile_rows=(5, 5), crop_to_rows=True)
  assert r.image.shape[0] == 0


@pytest.mark.parametrize('tile_size', [8, 16])
def test_reference_tail_lists_reproduce_the_reference_loop_order(tile_size):
  """with_reference_tail(): a user diffing against Taichi output gets the reference's visiting order (fact 8) from
  the product kernels — float64 against the oracle's emulation of forward.py:86-89."""
  from oracle import mapper as omap, raster as orast
  from taichi_splatting_amd import rasterize_with_tiles
  from taichi_splatting_amd.mapper.tile_mapper import with_reference_tail
  from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
  from taichi_splatting_amd.testing import random_2d_gaussians
  size = (96, 64)
  cfg = RasterConfig(tile_size=tile_size, pixel_stride=(1, 1) if tile_size == 8 else (2, 2))
  torch.manual_seed(tile_size)
  g = random_2d_gaussians(6000, size, scale_factor=4.0, alpha_range=(0.02, 0.3))       # several groups per tile
  p, f = project_gaussians2d(g).double(), g.feature.double()
  o2p, ranges, _ = omap.map_to_tiles(p.numpy().astype('float32'), g.depths.numpy(), size, tile_size)
  o2p, ranges = torch.from_numpy(o2p), torch.from_numpy(ranges)
  counts = (ranges[..., 1] - ranges[..., 0]).flatten()
  assert int((counts > tile_size * tile_size).sum()) > 0
  plain, _, _ = orast.forward(p, f, ranges, o2p, size, cfg)
  want, want_alpha, _ = orast.forward(p, f, ranges, o2p, size, cfg, emulate_reference_loop_bound=True)
  assert float((plain - want).abs().max()) > 1e-4, "the scene must show the deviation"
  o2p_ref, ranges_ref = with_reference_tail(o2p.to(DEV), ranges.to(DEV), tile_size)
  with torch.no_grad():
    out = rasterize_with_tiles(p.to(DEV), f.to(DEV), o2p_ref, ranges_ref.view(-1, 2), size, cfg)
  assert torch.allclose(out.image.cpu(), want, atol=1e-9)
  assert torch.allclose(out.image_weight.cpu(), want_alpha, atol=1e-9)


@pytest.mark.parametrize('crop', [False, True])
def test_broadcast_image_gradient_needs_no_copy(crop):
  """image.sum().backward() hands the frame an EXPANDED scalar as dL/dimage; the moments kernel reads it as one pixel's
  values (ms_frame_grads.grad_image_broadcast).  Same gradients as with a materialised (H, W, 3) array of ones."""
  g, cam = make_scene(15000, (200, 144), seed=21, sh_degree=2)
  cfg = RasterConfig()
  out = []
  for materialise in (False, True):
    gd = g.clone().requires_grad_(True)
    r = frame.render_frame(gd, cam, cfg, True, tile_rows=(2, 7) if crop else None, crop_to_rows=crop)
    if materialise:
      (r.image * torch.full_like(r.image, 0.5)).sum().backward()
    else:
      (0.5 * r.image.sum()).backward()
    out.append([t.grad.clone() for t in (gd.position, gd.log_scaling, gd.rotation, gd.alpha_logit, gd.feature)])
  for a, b in zip(*out):
    scale = float(b.abs().max())
    assert scale > 0 and float((a - b).abs().max()) < 5e-5 * scale       # float-atomic noise only
