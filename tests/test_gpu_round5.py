"""-m gpu, round 5: scene shapes the bench scene does not have — heavy-tailed splat sizes
(the wave-cooperative walk of wide tile spans in the count / emit kernels), tiles whose lists go on long after every
pixel is opaque (the forward's spent-tile exit) — and the staging pipeline of the raster backward on tile lists of
every length class (1, 2, 3, 4+ batches; slots beyond the workgroup's thread count)."""
import math

import numpy as np
import pytest
import torch

from oracle import mapper as omap, raster as orast
from taichi_splatting_amd import RasterConfig, frame, map_to_tiles, rasterize_with_tiles, render_gaussians
from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
from taichi_splatting_amd.testing import random_2d_gaussians, random_3d_gaussians, random_camera

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def heavy_tail_2d(n, size, seed, share=0.05, factor=20.0):
  torch.manual_seed(seed)
  g = random_2d_gaussians(n, size, scale_factor=1.0, alpha_range=(0.1, 0.9))
  big = torch.rand(n) < share
  g.log_scaling[:] = g.log_scaling + math.log(factor) * big[:, None].float()
  return g, big


@pytest.mark.parametrize('tile', [8, 16, 32])
def test_heavy_tailed_sizes_count_emit_and_lists(tile):
  """5 % of the splats at 20 x scale: a wave of the count / emit kernels holds spans of 1-4 tiles next to spans of
  hundreds, which the whole wave walks (csrc/mapper.hip).  Lists against the numpy oracle (reference order: tile, depth
  bits, point index), both sequences identical, count == emit (the ranges partition [0, K))."""
  size = (512, 384)
  g, big = heavy_tail_2d(12000, size, seed=tile)
  assert int(big.sum()) > 300
  cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2))
  p, depth = project_gaussians2d(g), g.depths.reshape(-1, 1).clone()
  want_o2p, want_ranges, _ = omap.map_to_tiles(p.numpy(), depth.reshape(-1).numpy(), size, tile, cfg.alpha_threshold)
  lists = {}
  for method in ('direct', 'presort'):
    o2p, ranges = map_to_tiles(p.to(DEV), depth.to(DEV), size, cfg, method=method)
    lists[method] = (o2p.cpu(), ranges.cpu())
    r2 = ranges.view(-1, 2).cpu()
    live = r2[r2[:, 1] > r2[:, 0]]
    assert int(live[0, 0]) == 0 and int(live[-1, 1]) == o2p.shape[0] and bool((live[1:, 0] == live[:-1, 1]).all())
  assert torch.equal(lists['direct'][0], lists['presort'][0]) and torch.equal(lists['direct'][1], lists['presort'][1])
  o2p, ranges = lists['direct']
  assert o2p.shape[0] == want_o2p.shape[0], (o2p.shape[0], want_o2p.shape[0])
  assert np.array_equal(ranges.view(-1, 2).numpy(), np.asarray(want_ranges).reshape(-1, 2))
  assert np.array_equal(o2p.numpy(), np.asarray(want_o2p))
  # the wide spans really were there: a big splat covers dozens of tiles
  spans = torch.bincount(o2p.long(), minlength=p.shape[0])
  assert int(spans[big].max()) > 64 and float(spans[~big].float().mean()) < 16


def test_heavy_tailed_scene_settles_on_the_presort_sequence_with_the_same_image():
  """render_gaussians on a heavy-tailed 3D scene: the overlap total per gaussian is far above the crossover, so the
  frame executor maps with the pre-sort from the second frame on (the first frame of a shape runs the default), and the
  frames agree bit for bit (same lists, same kernels)."""
  torch.manual_seed(5)
  size = (512, 512)
  cam = random_camera(image_size=size)
  n = 40000
  g = random_3d_gaussians(n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9))
  big = torch.rand(n) < 0.05
  g = g.replace(log_scaling=g.log_scaling + math.log(20.0) * big[:, None].float(), feature=torch.rand(n, 3))
  cfg = RasterConfig()
  frame.release_caches()
  try:
    gd, camd = g.to(DEV), cam.to(device=DEV)
    images, modes = [], []
    for _ in range(3):
      with torch.no_grad():
        images.append(render_gaussians(gd, camd, cfg, use_sh=False).image.clone())
      key = frame._shape_key(torch.device(DEV), n, size, cfg, None, False)
      modes.append(frame._mapper_mode.get(key))
    from taichi_splatting_amd import _lib
    assert modes[-1] == _lib.MAPPER_PRESORT, modes
    assert torch.equal(images[0], images[1]) and torch.equal(images[1], images[2])
  finally:
    frame.release_caches()


def pile_2d(n, size, window, sigma, seed, alpha_range=(0.5, 0.95)):
  torch.manual_seed(seed)
  g = random_2d_gaussians(n, size, scale_factor=1.0, alpha_range=alpha_range)
  g.position[:] = torch.tensor([[window[0], window[1]]], dtype=torch.float32) + window[2] * torch.rand(n, 2)
  g.log_scaling[:] = torch.log(torch.full((n, 2), sigma))
  return g


@pytest.mark.parametrize('tile', [8, 16, 32])
def test_forward_stops_on_spent_tiles_without_changing_the_image(tile):
  """20 000 fairly opaque splats on a 40 x 40 px window: every pixel there is opaque after a few dozen splats and the
  product forward leaves the tile's list (csrc/raster_fast.hip, FWD_SPENT_T); the float64 generic kernel walks all of
  it, like the reference (forward.py:69-70 never sets its flag when blending).  Image, image weight and the per-splat
  visibility must agree to float32 rounding."""
  size = (160, 128)
  g = pile_2d(20000, size, (50.0, 40.0, 40.0), 1.5, seed=tile)
  cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2))
  cfg_v = RasterConfig(tile_size=tile, pixel_stride=cfg.pixel_stride, compute_visibility=True)
  p, depth, f = project_gaussians2d(g).to(DEV), g.depths.reshape(-1, 1).to(DEV), g.feature.to(DEV)
  o2p, ranges = map_to_tiles(p, depth, size, cfg)
  assert int((ranges[..., 1] - ranges[..., 0]).max()) > 3000
  with torch.no_grad():
    got = rasterize_with_tiles(p, f, o2p, ranges.view(-1, 2), size, cfg)
    got_v = rasterize_with_tiles(p, f, o2p, ranges.view(-1, 2), size, cfg_v)
    want = rasterize_with_tiles(p.double(), f.double(), o2p, ranges.view(-1, 2), size, cfg_v)
  assert (got.image.double() - want.image).abs().max().item() < 2e-6
  assert (got.image_weight.double() - want.image_weight).abs().max().item() < 2e-6
  assert torch.equal(got.image, got_v.image)
  assert float(want.image_weight.max()) > 0.9999                       # the window really is opaque
  vis, vis64 = got_v.visibility.double(), want.visibility
  assert (vis - vis64).abs().max().item() < 1e-5 * float(vis64.max())
  # and against the oracle (the reference's loop restated), every pixel
  img, alpha, _ = orast.forward(p.cpu().double(), f.cpu().double(), ranges.cpu(), o2p.cpu(), size, cfg)
  assert (got.image.cpu().double() - img).abs().max().item() < 1e-5
  assert (got.image_weight.cpu().double() - alpha).abs().max().item() < 1e-5


@pytest.mark.parametrize('per_tile', [40, 250, 268, 269, 530, 800, 1100, 2300])
def test_backward_staging_pipeline_on_every_list_length(per_tile):
  """The raster backward stages a tile's list in equal batches of <= 268 splats through a three-deep register pipeline
  whose last 12 slots travel one component per lane (csrc/raster_bwd_scan.hip, round 5).  Tile lists of 40 ... 2300
  entries — one batch with and without the extra slots, exactly 268 and 269, two, three, four and nine batches — against
  the pixel-per-lane kernels of raster_fast.hip (MS_RASTER_BWD=patch) on low-opacity splats (nothing saturates: every
  batch blends)."""
  import os
  size = (64, 48)                                    # 4 x 3 tiles of 16
  n = per_tile * 12
  torch.manual_seed(per_tile)
  g = random_2d_gaussians(n, size, scale_factor=0.6, alpha_range=(0.01, 0.03))
  # one splat per (tile, slot): centres inside their tile, small enough to stay there
  tiles = torch.arange(n) % 12
  g.position[:] = torch.stack([(tiles % 4) * 16 + 3.0 + 10.0 * torch.rand(n), (tiles // 4) * 16 + 3.0 + 10.0 * torch.rand(n)], 1)
  g.log_scaling[:] = torch.log(0.5 + 0.4 * torch.rand(n, 2))
  cfg = RasterConfig(tile_size=16)
  p0, depth, f0 = project_gaussians2d(g).to(DEV), g.depths.reshape(-1, 1).to(DEV), g.feature.to(DEV)
  o2p, ranges = map_to_tiles(p0, depth, size, cfg)
  runs = (ranges[..., 1] - ranges[..., 0]).flatten()
  assert int(runs.min()) >= per_tile and int(runs.max()) <= per_tile + 40
  torch.manual_seed(1)
  G = torch.rand(size[1], size[0], 3, device=DEV) + 0.5
  grads = {}
  for mode in ('scan', 'patch'):
    os.environ['MS_RASTER_BWD'] = mode
    try:
      p, f = p0.clone().requires_grad_(True), f0.clone().requires_grad_(True)
      out = rasterize_with_tiles(p, f, o2p, ranges.view(-1, 2), size, cfg)
      (out.image * G).sum().backward()
      grads[mode] = (p.grad.clone(), f.grad.clone())
    finally:
      os.environ.pop('MS_RASTER_BWD', None)
  # every gradient row to 2e-5 of the largest gradient, except the odd splat with a pixel on the blend gate (the two
  # kernels round alpha * g differently; one flipped pair moves that splat's row): at most two such rows (and a handful behind them at that pixel), none of the
  # size a dropped or misplaced slot would cause (those lose a whole splat: relative error of order 1 on its row)
  for a, b, what in zip(grads['scan'], grads['patch'], ('gaussians2d', 'features')):
    scale = float(b.abs().max())
    rel = (a - b).abs().max(dim=1).values / scale
    assert int((rel > 2e-5).sum()) <= max(8, 0.003 * rel.numel()) and int((rel > 1e-3).sum()) <= max(2, 0.0005 * rel.numel()), \
      (what, per_tile, int((rel > 2e-5).sum()), int((rel > 1e-3).sum()), float(rel.max()))
    assert float(rel.max()) < 0.2, (what, per_tile, float(rel.max()))
    own = (a - b).abs().max(dim=1).values / b.abs().max(dim=1).values.clamp_min(1e-3 * scale)
    assert int((own > 0.5).sum()) == 0, (what, per_tile, "a splat lost most of its gradient")
    assert float(a.abs().max()) > 0


@pytest.mark.parametrize('tile,heuristics,deterministic', [(16, False, False), (16, True, True), (32, False, True), (8, False, False)])
def test_splat_row_kernels_equal_the_dense_kernels(tile, heuristics, deterministic):
  """The product raster kernels gathering from the splat-row table (include/mi355_splat.h: ms_splat_rows_pack,
  ms_raster_fwd_rows, ms_raster_bwd_moments_rows — what the frame executor runs) against the same kernels on the dense
  points7 / colour arrays: identical arithmetic on identical values, so the forward image, alpha and visibility are bit
  for bit the same, and so are the moment rows when they are committed in fixed point (float atomics add in arrival
  order: 2e-6 of the column maximum then).  Tile 8's backward is not offered on rows (it measured slower)."""
  import ctypes
  from taichi_splatting_amd import _lib
  lib = _lib.load()
  size = (400, 304)
  torch.manual_seed(tile)
  g = random_2d_gaussians(30000, size, scale_factor=1.5, alpha_range=(0.1, 0.9))
  cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2), compute_visibility=True,
                     compute_point_heuristic=heuristics)
  p, depth, f = project_gaussians2d(g).to(DEV).contiguous(), g.depths.reshape(-1, 1).to(DEV), g.feature.to(DEV).contiguous()
  o2p, ranges = map_to_tiles(p, depth, size, cfg)
  ranges2 = ranges.view(-1, 2).contiguous()
  n, (w, h) = p.shape[0], size
  cfg_c, stream = _lib.raster_config_c(cfg), _lib.current_stream(torch.device(DEV))
  th = (h + tile - 1) // tile
  rows = torch.full((n, _lib.SPLAT_ROW), float('nan'), device=DEV)
  _lib.check(lib.ms_splat_rows_pack(p.data_ptr(), depth.data_ptr(), f.data_ptr(), n, rows.data_ptr(), stream), "pack")
  assert torch.equal(rows[:, :7], p) and torch.equal(rows[:, 7], depth.reshape(-1)) and torch.equal(rows[:, 8:11], f)

  def forward(use_rows):
    image, alpha, vis = torch.empty((h, w, 3), device=DEV), torch.empty((h, w), device=DEV), torch.zeros(n, device=DEV)
    if use_rows:
      _lib.check(lib.ms_raster_fwd_rows(rows.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), w, h, cfg_c, image.data_ptr(),
                                        alpha.data_ptr(), vis.data_ptr(), 0, th, stream), "fwd rows")
    else:
      _lib.check(lib.ms_raster_fwd(p.data_ptr(), f.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), w, h, 3, cfg_c,
                                   image.data_ptr(), alpha.data_ptr(), vis.data_ptr(), 0, th, _lib.dtype_code(torch.float32),
                                   stream), "fwd")
    return image, alpha, vis
  (img_r, a_r, v_r), (img_d, a_d, v_d) = forward(True), forward(False)
  assert torch.equal(img_r, img_d) and torch.equal(a_r, a_d)
  assert (v_r - v_d).abs().max().item() <= 1e-5 * float(v_d.max())          # float atomics
  assert float(img_d.max()) > 0.1

  torch.manual_seed(1)
  G = torch.rand_like(img_d) + 0.5
  exps = None
  if deterministic:
    exps = torch.empty(2, dtype=torch.int32, device=DEV)
    _lib.check(lib.ms_fixed_point_exponents(G.abs().amax().reshape(1).data_ptr(), exps.data_ptr(), stream), "exp")

  def backward(use_rows):
    mom = torch.zeros((n, _lib.MOMENT_ROW), dtype=torch.int64 if deterministic else torch.float32, device=DEV)
    args = (ranges2.data_ptr(), o2p.data_ptr(), img_d.data_ptr(), G.data_ptr(), w, h, cfg_c, mom.data_ptr(),
            1 if deterministic else 0, exps.data_ptr() if deterministic else None, 0, th, stream)
    if use_rows:
      rc = lib.ms_raster_bwd_moments_rows(rows.data_ptr(), *args)
    else:
      rc = lib.ms_raster_bwd_moments(p.data_ptr(), f.data_ptr(), *args)
    return rc, mom
  rc, mom_r = backward(True)
  if tile == 8:
    assert rc == -2                                                     # MS_ERR_UNSUPPORTED
    return
  _lib.check(rc, "bwd rows")
  rc, mom_d = backward(False)
  _lib.check(rc, "bwd")
  if deterministic:
    assert torch.equal(mom_r, mom_d) and int(mom_d.abs().max()) > 0
  else:
    scale = mom_d.abs().amax(dim=0).clamp_min(1e-30)
    assert float(((mom_r - mom_d).abs() / scale).max()) < 2e-6 and float(mom_d.abs().max()) > 0


def test_frame_executor_with_splat_rows_switched_on():
  """MS_SPLAT_ROWS=1 (read once per process, hence the subprocess): the frame executor fills the splat-row table in its
  projection and SH kernels and rasterizes from it — SH degree 2 (the generic SH kernel writes the colour words), degree 3
  (the streaming kernel does) and no SH (the projection kernel copies the features).  Same values gathered, same
  arithmetic: the image is bit for bit that of a process without the table, the gradients agree to the order of their
  float atomics."""
  import os, subprocess, sys, textwrap
  code = textwrap.dedent('''
    import torch
    from taichi_splatting_amd import RasterConfig, render_gaussians, frame
    from taichi_splatting_amd.testing import random_3d_gaussians, random_camera
    dev = 'cuda:0'
    torch.manual_seed(3)
    cam = random_camera(image_size=(320, 240))
    out = {}
    for name, shape, use_sh in (('deg2', (3, 9), True), ('deg3', (3, 16), True), ('rgb', (3,), False)):
      g = random_3d_gaussians(20000, cam, scale_factor=1.0, alpha_range=(0.1, 0.9))
      g = g.replace(feature=torch.rand(20000, *shape) * (0.5 if use_sh else 1.0))
      g, c = g.to(dev), cam.to(device=dev)
      g.requires_grad_(True)
      r = render_gaussians(g, c, RasterConfig(), use_sh=use_sh)
      (r.image * torch.linspace(0.5, 1.5, 320 * 240 * 3, device=dev).view(240, 320, 3)).sum().backward()
      out[name] = (r.image.detach().cpu(), g.position.grad.cpu(), g.feature.grad.cpu())
    torch.save(out, OUT)
  ''')
  import tempfile
  results = {}
  with tempfile.TemporaryDirectory() as tmp:
    for mode in ('0', '1'):
      path = os.path.join(tmp, f'rows{mode}.pt')
      env = dict(os.environ, MS_SPLAT_ROWS=mode)
      subprocess.run([sys.executable, '-c', code.replace('OUT', repr(path))], check=True, env=env, timeout=300,
                     cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
      results[mode] = torch.load(path)
  for name in ('deg2', 'deg3', 'rgb'):
    (img0, gp0, gf0), (img1, gp1, gf1) = results['0'][name], results['1'][name]
    assert torch.equal(img0, img1), name
    assert float(img0.max()) > 0.05
    for a, b in ((gp0, gp1), (gf0, gf1)):
      assert float((a - b).abs().max()) <= 2e-5 * float(a.abs().max()), name
      assert float(a.abs().max()) > 0


@pytest.mark.parametrize('tile,heuristics,visibility', [(16, False, True), (16, True, False), (32, False, True), (8, False, False)])
def test_long_tile_runs_cut_into_segments_match_the_per_tile_kernels(tile, heuristics, visibility):
  """120 000 + 60 000 splats piled onto two windows (runs of 20 000 - 120 000 entries on a handful of tiles, ordinary
  tiles around them): ms_raster_fwd_split / ms_raster_bwd_moments_split cut the runs above 16 384 entries into
  segments blended by separate workgroups (csrc/raster_common.h, "Long tile runs") and must reproduce the per-tile
  kernels — image and alpha to float32 rounding of the re-associated transmittance products, the moment rows to 2e-5 of
  the column maximum except for the odd splat behind a pixel whose transmittance sits on the backward's saturation
  test.  Opacities just above the blend gate and sub-pixel footprints: no pixel saturates before the last segment, every
  segment really blends (image_alpha ends near 0.99)."""
  from taichi_splatting_amd import _lib
  lib = _lib.load()
  size = (256, 192)
  torch.manual_seed(tile)
  g = random_2d_gaussians(200000, size, scale_factor=1.0, alpha_range=(0.0045, 0.008))
  n = g.position.shape[0]
  g.position[:120000] = torch.tensor([[34.0, 34.0]]) + 12.0 * torch.rand(120000, 2)
  g.position[120000:180000] = torch.tensor([[148.0, 100.0]]) + 8.0 * torch.rand(60000, 2)
  g.log_scaling[:] = torch.log(0.5 + 0.5 * torch.rand(n, 2))
  cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2), compute_point_heuristic=heuristics,
                     compute_visibility=visibility)
  p, depth, f = project_gaussians2d(g).to(DEV).contiguous(), g.depths.reshape(-1, 1).to(DEV), g.feature.to(DEV).contiguous()
  o2p, ranges = map_to_tiles(p, depth, size, cfg)
  ranges2 = ranges.view(-1, 2).contiguous()
  runs = (ranges2[:, 1] - ranges2[:, 0])
  assert int((runs > 16384).sum()) >= 2 and int(((runs > 0) & (runs <= 16384)).sum()) >= 4, runs.max()
  k, (w, h) = o2p.shape[0], size
  cfg_c, stream = _lib.raster_config_c(cfg), _lib.current_stream(torch.device(DEV))
  th = (h + tile - 1) // tile
  scratch = torch.empty((lib.ms_raster_split_scratch_bytes(k, tile, 0, 0),), dtype=torch.uint8, device=DEV)
  assert scratch.data_ptr() % 256 == 0

  def forward(split):
    image, alpha = torch.full((h, w, 3), float('nan'), device=DEV), torch.full((h, w), float('nan'), device=DEV)
    vis = torch.zeros(n, device=DEV) if visibility else None
    if split:
      _lib.check(lib.ms_raster_fwd_split(p.data_ptr(), f.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), k, w, h, cfg_c,
                                         image.data_ptr(), alpha.data_ptr(), _lib.ptr(vis), scratch.data_ptr(), 0, 0, 0, th, stream),
                 "fwd split")
    else:
      _lib.check(lib.ms_raster_fwd(p.data_ptr(), f.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), w, h, 3, cfg_c,
                                   image.data_ptr(), alpha.data_ptr(), _lib.ptr(vis), 0, th, _lib.dtype_code(torch.float32),
                                   stream), "fwd")
    return image, alpha, vis
  (img_s, a_s, vis_s), (img_d, a_d, vis_d) = forward(True), forward(False)
  if visibility:
    # per-splat sums of the blend weights: the segments are walked a second time from their true transmittance
    assert float(vis_d.max()) > 0 and (vis_s - vis_d).abs().max().item() <= 2e-5 * float(vis_d.max())
  counts = scratch[:16].view(torch.int32).cpu()
  assert int(counts[1]) == int((runs > 16384).sum()) and int(counts[0]) >= 2 * int(counts[1]) and int(counts[2]) == 0
  assert not bool(torch.isnan(img_s).any()) and not bool(torch.isnan(a_s).any())
  assert (img_s - img_d).abs().max().item() < 3e-6 and (a_s - a_d).abs().max().item() < 3e-6
  assert float(a_d.max()) > 0.5

  torch.manual_seed(1)
  G = torch.rand_like(img_d) + 0.5

  def backward(split, image):
    mom = torch.zeros((n, _lib.MOMENT_ROW), device=DEV)
    if split:
      _lib.check(lib.ms_raster_bwd_moments_split(p.data_ptr(), f.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), k,
                                                 image.data_ptr(), G.data_ptr(), w, h, cfg_c, mom.data_ptr(), 0, None,
                                                 scratch.data_ptr(), 0, 0, 0, th, stream), "bwd split")
    else:
      _lib.check(lib.ms_raster_bwd_moments(p.data_ptr(), f.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), image.data_ptr(),
                                           G.data_ptr(), w, h, cfg_c, mom.data_ptr(), 0, None, 0, th, stream), "bwd")
    return mom
  mom_s, mom_d = backward(True, img_s), backward(False, img_d)
  scale = mom_d.abs().amax(dim=0).clamp_min(1e-30)
  rel = ((mom_s - mom_d).abs() / scale).amax(dim=1)
  assert float(mom_d.abs().max()) > 0
  assert int((rel > 2e-5).sum()) <= max(8, 0.002 * n), (int((rel > 2e-5).sum()), float(rel.max()))
  assert float(rel.max()) < 0.05, float(rel.max())


def test_frame_executor_cuts_long_runs_from_the_second_frame_on():
  """render_gaussians on a scene whose splats pile onto a few tiles: the first frame of the shape rasterizes every tile
  with one workgroup and notes the long run in the shape's pinned word (csrc/raster_fast.hip); from the next frame on
  the shape maps with the pre-sort AND has its long runs blended in segments (frame.py, ms_frame_desc.split_long_runs).
  Image and gradients of the later frames against the first: float32 rounding of re-associated products."""
  torch.manual_seed(11)
  size = (256, 192)
  cam = random_camera(image_size=size)
  n = 150000
  g = random_3d_gaussians(n, cam, scale_factor=0.3, alpha_range=(0.005, 0.01))
  # the generator's positions squeezed towards the view axis: everything lands on a few tiles around the image centre
  centre = g.position.mean(dim=0, keepdim=True)
  g = g.replace(position=centre + (g.position - centre) * torch.tensor([[0.04, 0.04, 1.0]]), feature=torch.rand(n, 3))
  cfg = RasterConfig()
  frame.release_caches()
  try:
    out = []
    for i in range(3):
      gd, camd = g.to(DEV), cam.to(device=DEV)
      gd.requires_grad_(True)
      r = render_gaussians(gd, camd, cfg, use_sh=False)
      (r.image * torch.linspace(0.5, 1.5, size[0] * size[1] * 3, device=DEV).view(size[1], size[0], 3)).sum().backward()
      out.append((r.image.detach().clone(), gd.position.grad.clone(), gd.feature.grad.clone(), gd.log_scaling.grad.clone()))
      torch.cuda.synchronize()
      key = frame._shape_key(torch.device(DEV), n, size, cfg, None, False)
      if i == 0:
        st = frame.frame_status(r)
        assert st['overlaps'] > 100000
    assert key in frame._presort_sticky, "no run above 16 384 entries: the scene does not test anything"
    assert float(out[0][0].max()) > 0.05
    for later in out[1:]:
      assert (later[0] - out[0][0]).abs().max().item() < 3e-6
      for a, b in zip(later[1:], out[0][1:]):
        scale = float(b.abs().max())
        assert scale > 0
        rel = (a - b).abs().flatten(1).amax(dim=1) / scale
        assert int((rel > 2e-5).sum()) <= max(8, 0.002 * n) and float(rel.max()) < 0.05, (int((rel > 2e-5).sum()), float(rel.max()))
    assert torch.equal(out[1][0], out[2][0])
    # the same frame captured in a HIP graph (the plan's memset and the segment launches are part of the capture)
    gd, camd = g.to(DEV), cam.to(device=DEV)
    gd.requires_grad_(True)
    leaves = [gd.position, gd.log_scaling, gd.rotation, gd.alpha_logit, gd.feature]
    weight = torch.linspace(0.5, 1.5, size[0] * size[1] * 3, device=DEV).view(size[1], size[0], 3)

    def step():
      for t in leaves:
        t.grad = None
      r = render_gaussians(gd, camd, cfg, use_sh=False)
      (r.image * weight).sum().backward()
      return r
    del r
    graph = frame.FrameGraph(step, warmup=2)
    for _ in range(2):
      r = graph.replay()
    torch.cuda.synchronize()
    assert int(r.frame.desc.split_long_runs) == 1
    assert torch.equal(r.image, out[1][0])
    rel = (gd.position.grad - out[1][1]).abs().flatten(1).amax(dim=1) / float(out[1][1].abs().max())
    assert float(rel.max()) < 1e-4
  finally:
    frame.release_caches()
