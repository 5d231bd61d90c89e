"""CPU tests of the host-side pieces added in round 3: reference module paths kept importable, strip balancing with
rows the mapper culls, the shared identity index list under inference mode, the frame executor's layout query."""
import ctypes

import torch

import taichi_splatting_amd as pkg
from taichi_splatting_amd import RasterConfig


def test_reference_module_paths_stay_importable():
  pkg.install_as_taichi_splatting()
  from taichi_splatting.optim.util import (get_vector_state, get_scalar_state, get_total_weight, get_running_vis,
                                           flatten_param)
  from taichi_splatting.benchmarks.util import benchmarked            # noqa: F401
  import taichi_splatting.benchmarks.bench_projection as bp
  import taichi_splatting.benchmarks.bench_rasterizer as br
  import taichi_splatting.benchmarks.bench_tilemapper as bt
  import taichi_splatting.benchmarks.bench_sh as bs
  assert all(callable(m.main) for m in (bp, br, bt, bs))
  p = torch.zeros(5, 2, 3, requires_grad=True)
  state = {}
  first, second = get_vector_state(state, p)
  assert first.shape == (5, 6) and second.shape == (5,) and state['v'] is first and state['m'] is second
  first, second = get_scalar_state({}, p)
  assert first.shape == (5, 6) and second.shape == (5, 6)
  assert get_total_weight(state, 5, p.device).shape == (5,) and get_running_vis(state, (5,), p.device).shape == (5,)
  p.grad = torch.ones_like(p)
  flat, grad = flatten_param(p)
  assert flat.shape == (5, 6) and grad.shape == (5, 6)


def test_overlap_balanced_bounds_ignores_rows_the_mapper_culls():
  # alpha below the threshold / NaN / inf rows made NaN tile indexes (INT64_MIN after the cast)
  from taichi_splatting_amd.distributed import overlap_balanced_bounds
  torch.manual_seed(0)
  g = torch.rand(400, 7)
  g[:, 0:2] *= 256
  g[:, 4:6] += 1.0
  g[:, 6] = 0.5
  clean = overlap_balanced_bounds(g, (256, 256), RasterConfig(), 4)
  dirty = g.clone()
  extra = g[:6].clone()
  extra[0, 6] = 1e-4; extra[1, 6] = float('nan'); extra[2, 4] = float('inf'); extra[3, 1] = float('nan')
  extra[4, 6] = 0.0; extra[5, 5] = float('-inf')
  assert overlap_balanced_bounds(torch.cat([dirty, extra]), (256, 256), RasterConfig(), 4) == clean
  assert clean[0] == 0 and clean[-1] == 16 and all(b1 >= b0 for b0, b1 in zip(clean, clean[1:]))


def test_identity_indexes_made_under_inference_mode_serve_training_frames():
  from taichi_splatting_amd import frame
  frame._identity.clear()
  dev = torch.device('cpu')
  with torch.inference_mode():
    idx = frame.identity_indexes(7, dev)
  assert not idx.is_inference()
  feature = torch.rand(7, 3, requires_grad=True)
  feature[frame.identity_indexes(7, dev)].sum().backward()       # "Inference tensors cannot be saved for backward"
  assert feature.grad is not None
  assert frame.identity_indexes(7, dev) is idx


def test_frame_layout_query_and_dispatch_without_a_gpu():
  from taichi_splatting_amd import _lib
  lib = _lib.load()
  cfg = RasterConfig()
  d = _lib.FrameDescC(n=1000, k_capacity=5000, image_w=250, image_h=130, dtype=_lib.MS_F32, f=3, sh_degree=3, depth16=0,
                      tile_row_begin=0, tile_row_end=1 << 30, projected_input=0, raster=_lib.raster_config_c(cfg))
  lay = _lib.FrameLayoutC()
  assert lib.ms_frame_layout_query(ctypes.byref(d), ctypes.byref(lay)) == 0
  tiles = ((250 + 15) // 16) * ((130 + 15) // 16)
  assert lay.keep_n_bytes >= 1000 * (28 + 4 + 12) + tiles * 8 and lay.keep_k_bytes >= 5000 * 4
  offsets = [lay.points7, lay.depth, lay.colours, lay.camera_position, lay.counters, lay.tile_ranges]
  assert offsets == sorted(offsets) and all(o % 256 == 0 for o in offsets)
  assert lib.ms_frame_uses_moments(ctypes.byref(d), 0) == 1
  d.raster.tile_size = 32                # the scan backward serves every tile size since the 1024-thread tile-32 variant
  assert lib.ms_frame_uses_moments(ctypes.byref(d), 0) == 1 and lib.ms_frame_uses_moments(ctypes.byref(d), 1) == 1
  d.raster.tile_size = 16; d.raster.antialias = 1
  assert lib.ms_frame_uses_moments(ctypes.byref(d), 0) == 0
  d.f = 7                        # no instantiation: argument error, not a crash
  assert lib.ms_frame_layout_query(ctypes.byref(d), ctypes.byref(lay)) == -2
  d.f = 3; d.projected_input = 1  # projected input cannot carry SH
  assert lib.ms_frame_layout_query(ctypes.byref(d), ctypes.byref(lay)) == -1


def test_parked_gc_restores_the_collector():
  """frame.parked_gc: the collector is off inside the block and back to its previous state after it, also on error"""
  import gc
  from taichi_splatting_amd import frame
  assert gc.isenabled()
  with frame.parked_gc():
    assert not gc.isenabled()
  assert gc.isenabled()
  try:
    with frame.parked_gc():
      raise ValueError("boom")
  except ValueError:
    pass
  assert gc.isenabled()
  gc.disable()
  try:
    with frame.parked_gc():
      pass
    assert not gc.isenabled()            # it was off before: stays off
  finally:
    gc.enable()
