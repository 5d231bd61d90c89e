"""-m gpu: the frame executor's direct-order mapper (storage-order emission -> stable sort by tile -> per-tile depth
sort, csrc/tile_sort.hip) against the pre-sort construction of the same lists (gaussians sorted by depth, overlaps
sorted by tile id: map_to_tiles(method='presort'); both are held against the numpy oracle in test_gpu_mapper.py, which
restates the reference's one sort of tile << 32 | depth bits, mapper/tile_mapper.py:115-170).  Integer outputs:
overlap_to_point and tile_ranges must be IDENTICAL — same (tile, depth key, point index) order, ties included.

The scenes aim at the branches of tile_sort.hip: runs in each of the three LDS size classes (up to 1024 / 2560 / 5120 entries) and above them,
runs of one key, keys that pile up in few buckets (the bounded radix path), 16 bit keys, zero / denormal depths."""
import pytest
import torch

from taichi_splatting_amd import RasterConfig, frame, map_to_tiles
from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
from taichi_splatting_amd.testing import random_2d_gaussians

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def executor_map(p, depth, features, size, cfg, use_depth16=False):
  state = frame.FrameState()
  frame._RasterizeFrameFunction.apply(p, depth, features, tuple(size), cfg, bool(use_depth16), state)
  state.settle()
  k, live, overflow = state.counters()[:3].tolist()
  assert not overflow and live == k
  return state.overlap_to_point()[:k].clone(), state.tile_ranges().clone()


def check(p, depth, size, cfg, use_depth16=False):
  p, depth = p.to(DEV), depth.to(DEV)
  features = torch.rand(p.shape[0], 3, device=DEV)
  want_o2p, want_ranges = map_to_tiles(p, depth.reshape(-1, 1), size, cfg, use_depth16=use_depth16, method='presort')
  o2p, ranges = executor_map(p, depth, features, size, cfg, use_depth16)
  m_o2p, m_ranges = map_to_tiles(p, depth.reshape(-1, 1), size, cfg, use_depth16=use_depth16, method='direct')
  assert torch.equal(m_ranges, want_ranges) and torch.equal(m_o2p, want_o2p)       # the modular operator's own direct path
  assert torch.equal(ranges, want_ranges)
  assert o2p.shape == want_o2p.shape
  bad = (o2p != want_o2p).nonzero()
  assert bad.numel() == 0, f"{bad.numel()} of {o2p.numel()} entries differ, first at {int(bad[0])}"
  longest = int((want_ranges[..., 1] - want_ranges[..., 0]).max())
  return longest


def scene(n, size, scale, seed, tile=16):
  torch.manual_seed(seed)
  g = random_2d_gaussians(n, size, scale_factor=scale, alpha_range=(0.2, 0.9))
  return project_gaussians2d(g), g.depths.reshape(-1).clone(), RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2))


@pytest.mark.parametrize('n,size,scale,tile', [
  (20000, (320, 240), 1.0, 16),        # 100-200 per tile: the small class
  (60000, (256, 192), 8.0, 16),        # 1300-3400 per tile: the middle and the long class
  (30000, (64, 64), 12.0, 16),         # every gaussian in every tile: runs far above 5120 (global radix path)
  (5000, (130, 70), 2.0, 8),
  (40000, (512, 256), 3.0, 32),
])
def test_direct_mapper_equals_modular(n, size, scale, tile):
  p, depth, cfg = scene(n, size, scale, 11, tile)
  longest = check(p, depth, size, cfg)
  assert longest > 1


def test_run_lengths_cover_the_size_classes_and_the_global_path():
  # the parametrised scenes above are only worth their names if the runs have the lengths they claim
  for (n, size, scale), (lo, hi) in [((20000, (320, 240), 1.0), (2, 1024)), ((60000, (256, 192), 8.0), (1025, 2560)), ((60000, (256, 192), 8.0), (2561, 5120)),
                                     ((30000, (64, 64), 12.0), (5121, 1 << 30))]:
    p, depth, cfg = scene(n, size, scale, 11)
    _, ranges = map_to_tiles(p.to(DEV), depth.to(DEV).reshape(-1, 1), size, cfg)
    lengths = (ranges[..., 1] - ranges[..., 0]).flatten()
    inside = ((lengths >= lo) & (lengths <= hi)).sum().item()
    assert inside >= lengths.numel() // 4, (n, size, lengths.min().item(), lengths.max().item())


@pytest.mark.parametrize('use_depth16', [False, True])
def test_ties_keep_point_order(use_depth16):
  # few distinct depths: long runs of equal keys inside every tile (16 bit keys quantise further)
  p, depth, cfg = scene(30000, (192, 128), 3.0, 5)
  depth = (depth * 7).floor() / 7 + 0.05
  check(p, depth, (192, 128), cfg, use_depth16)
  depth[:] = 0.25                                  # one key everywhere: the runs stay in point order
  check(p, depth, (192, 128), cfg, use_depth16)


@pytest.mark.parametrize('use_depth16', [False, True])
def test_skewed_depths_take_the_bounded_path(use_depth16):
  # nearly all keys in one bucket of the monotone map, a few outliers stretching the range
  p, depth, cfg = scene(40000, (128, 128), 4.0, 9)
  depth = 0.5 + depth * 1e-4
  depth[::997] = 1e-3
  depth[5::1013] = 0.999
  check(p, depth, (128, 128), cfg, use_depth16)
  # two clusters
  p, depth, cfg = scene(40000, (128, 128), 4.0, 10)
  depth = torch.where(torch.arange(depth.numel()) % 2 == 0, 0.2 + depth * 1e-6, 0.9 + depth * 1e-6)
  check(p, depth, (128, 128), cfg, use_depth16)


def test_zero_denormal_and_large_depths():
  p, depth, cfg = scene(20000, (160, 96), 3.0, 3)
  depth = depth.clone()
  depth[::5] = 0.0
  depth[1::5] = 1e-41                              # denormal float bits
  depth[2::5] *= 1e30
  check(p, depth, (160, 96), cfg)


def test_exponent_spread():
  # keys over many binades: the value map leaves the low binades almost empty and the high ones crowded
  p, depth, cfg = scene(50000, (128, 96), 4.0, 21)
  depth = torch.exp2(-20 * depth)
  check(p, depth, (128, 96), cfg)


@pytest.mark.parametrize('use_depth16', [False, True])
def test_render_path_with_culled_rows_and_ndc_keys(use_depth16):
  # render_gaussians: ndc depth keys made in the emit kernel and culled gaussians left in place (the executor), against
  # projection -> compaction -> map_to_tiles_strip on the visible rows (the modular path): same lists through idx
  from taichi_splatting_amd import render_gaussians
  from taichi_splatting_amd.mapper.tile_mapper import map_to_tiles_strip
  from taichi_splatting_amd.testing import random_3d_gaussians, random_camera
  torch.manual_seed(2)
  cam = random_camera(image_size=(320, 208))
  g = random_3d_gaussians(30000, cam, scale_factor=1.5, margin=0.3)
  centre = torch.inverse(cam.T_camera_world)[0:3, 3].reshape(1, 3)
  g.position[-500:] = 2 * centre - g.position[-500:]                        # mirrored through the camera centre: behind it
  g = g.to(DEV)
  cam = cam.to(DEV)
  cfg = RasterConfig(tile_size=16)
  r = render_gaussians(g, cam, cfg, use_sh=False, use_depth16=use_depth16)
  k = frame.frame_status(r)['overlaps']
  state = r.frame
  o2p, ranges = state.overlap_to_point()[:k], state.tile_ranges()
  pts = r.points
  assert 0 < pts.idx.numel() < g.position.shape[0]                          # something was culled
  want_o2p, want_ranges = map_to_tiles_strip(pts.gaussians2d.detach(), pts.depths.detach().reshape(-1, 1), image_size=cam.image_size,
                                             config=cfg, use_depth16=use_depth16, ndc_range=(cam.near_plane, cam.far_plane),
                                             method='presort')
  assert torch.equal(ranges, want_ranges.view_as(ranges))
  assert torch.equal(o2p, pts.idx.reshape(-1).to(torch.int32)[want_o2p.long()])


# ---- the primitive through the C-ABI (ms_tile_depth_sort) against a stable composite sort in torch -------------------

def run_primitive(lengths, keys):
  import ctypes
  from taichi_splatting_amd import _lib
  lib = _lib.load()
  lengths = torch.as_tensor(lengths, dtype=torch.int64)
  ends = torch.cumsum(lengths, 0)
  ranges = torch.stack([ends - lengths, ends], dim=1).to(torch.int32)
  ranges[lengths == 0] = 0                                         # empty tiles are [0, 0) (find_ranges)
  k = int(ends[-1])
  tile = torch.repeat_interleave(torch.arange(lengths.numel()), lengths)
  keys = keys.to(torch.int64) & 0xffffffff
  ids = torch.arange(k, dtype=torch.int32) * 3 + 1                 # ascending inside every run, not the position
  composite = (tile << 32) | keys
  want = ids[torch.sort(composite, stable=True).indices]
  srt = composite.to(DEV)
  o2p = ids.to(DEV)
  scratch = torch.empty(k, dtype=torch.int64, device=DEV)
  _lib.check(lib.ms_tile_depth_sort(ranges.to(DEV).data_ptr(), lengths.numel(), srt.data_ptr(), o2p.data_ptr(),
                                    scratch.data_ptr(), _lib.current_stream(torch.device(DEV))), "ms_tile_depth_sort")
  torch.cuda.synchronize()
  bad = (o2p.cpu() != want).nonzero()
  assert bad.numel() == 0, f"{bad.numel()} of {k} entries differ, first at {int(bad[0])} (tile {int(tile[bad[0]])})"


LENGTHS = [0, 1, 2, 63, 255, 256, 257, 1023, 1024, 1025, 0, 2559, 2560, 2561, 5119, 5120, 5121, 20000, 3, 0]


@pytest.mark.parametrize('kind', ['u32', 'float01', 'narrow', 'two_values', 'constant', 'sign_bits', 'one_outlier'])
def test_primitive_every_size_class_and_key_shape(kind):
  torch.manual_seed(17)
  k = sum(LENGTHS)
  if kind == 'u32':
    keys = torch.randint(0, 1 << 32, (k,), dtype=torch.int64)
  elif kind == 'float01':
    keys = torch.rand(k).view(torch.int32).to(torch.int64)
  elif kind == 'narrow':
    keys = 0x3f000000 + torch.randint(0, 40, (k,), dtype=torch.int64)         # many ties
  elif kind == 'two_values':
    keys = torch.where(torch.rand(k) < 0.5, 7, 0x3f7fffff).to(torch.int64)
  elif kind == 'constant':
    keys = torch.full((k,), 0x3e99999a, dtype=torch.int64)
  elif kind == 'sign_bits':
    keys = (torch.randn(k) * 1e3).view(torch.int32).to(torch.int64)            # negative floats, large magnitudes
  else:
    keys = (0.5 + torch.rand(k) * 1e-5).view(torch.int32).to(torch.int64)
    keys[::4001] = 1                                                           # stretches every run's range
  run_primitive(LENGTHS, keys)


def test_primitive_many_small_tiles_and_one_huge():
  torch.manual_seed(23)
  lengths = torch.randint(0, 40, (5000,)).tolist() + [300000] + torch.randint(900, 1100, (64,)).tolist()
  k = sum(lengths)
  run_primitive(lengths, torch.rand(k).view(torch.int32).to(torch.int64))
  run_primitive(lengths, torch.randint(0, 65536, (k,), dtype=torch.int64))     # 16 bit keys


def test_both_launch_sequences_give_the_same_lists_and_the_policy_switches():
  # frame.py remembers per scene shape which sequence the next frame uses (overlaps per gaussian); forced here
  p, depth, cfg = scene(30000, (192, 128), 3.0, 13)
  p, depth = p.to(DEV), depth.to(DEV)
  features = torch.rand(p.shape[0], 3, device=DEV)
  frame.release_caches()
  keep = frame.PRESORT_ABOVE, frame.DIRECT_BELOW
  try:
    results = []
    for above, below, want_second in ((1e9, 1e9, 0), (-1.0, -1.0, 1)):
      frame.release_caches()
      frame.PRESORT_ABOVE, frame.DIRECT_BELOW = above, below
      modes = []
      for _ in range(2):                       # the first frame of a shape maps with the pre-sort; the second as decided
        state = frame.FrameState()
        frame._RasterizeFrameFunction.apply(p, depth, features, (192, 128), cfg, False, state)
        state.settle()
        modes.append(int(state.desc.mapper))
      assert modes == [1, want_second]
      k = int(state.counters()[0])
      results.append((state.overlap_to_point()[:k].clone(), state.tile_ranges().clone()))
    assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])
  finally:
    frame.PRESORT_ABOVE, frame.DIRECT_BELOW = keep
    frame.release_caches()


def test_a_giant_run_moves_the_scene_shape_to_the_presort_sequence():
  # 60 000 small splats inside ONE tile: the direct sequence sorts that run with one workgroup and reports its length
  # through a pinned word; frame.py then maps the shape with the pre-sort, whose cost does not depend on the spread
  torch.manual_seed(31)
  size, n = (256, 256), 60000
  g = random_2d_gaussians(n, size, scale_factor=0.2, alpha_range=(0.3, 0.9))
  g.position[:] = 84.0 + 6.0 * torch.rand(n, 2)                      # all inside tile (5, 5) of the 16 px grid
  g.log_scaling[:] = torch.log(torch.full((n, 2), 0.4))
  p, depth = project_gaussians2d(g).to(DEV), g.depths.reshape(-1).to(DEV)
  features = torch.rand(n, 3, device=DEV)
  cfg = RasterConfig(tile_size=16)
  want_o2p, want_ranges = map_to_tiles(p, depth.reshape(-1, 1), size, cfg, method='presort')
  assert int((want_ranges[..., 1] - want_ranges[..., 0]).max()) > frame.LONG_RUN_LIMIT
  frame.release_caches()
  try:
    # a shape that had settled on the direct sequence (few overlaps per gaussian) and now piles up: the direct sequence's
    # per-tile sort reports the run, the next frame maps with the pre-sort.  (The FIRST frame of an unknown shape maps with
    # the pre-sort anyway since round 6.)
    key = ('2d',) + frame._shape_key(torch.device(DEV), n, size, cfg, None, False)
    frame._mapper_mode[key] = _lib_direct()
    modes = []
    for _ in range(3):
      state = frame.FrameState()
      frame._RasterizeFrameFunction.apply(p, depth, features, size, cfg, False, state)
      state.settle()
      torch.cuda.synchronize()                                       # the word of this frame is written
      modes.append(int(state.desc.mapper))
      k = int(state.counters()[0])
      assert torch.equal(state.tile_ranges(), want_ranges) and torch.equal(state.overlap_to_point()[:k], want_o2p)
    assert modes == [0, 1, 1]
    frame.release_caches()
    modes = []
    for _ in range(2):
      state = frame.FrameState()
      frame._RasterizeFrameFunction.apply(p, depth, features, size, cfg, False, state)
      state.settle()
      torch.cuda.synchronize()
      modes.append((int(state.desc.mapper), int(state.desc.split_long_runs)))
    assert modes == [(1, 1), (1, 1)] and key in frame._presort_sticky
  finally:
    frame.release_caches()


def _lib_direct():
  from taichi_splatting_amd import _lib
  return _lib.MAPPER_DIRECT
