"""One table-driven harness for the per-stage benchmarks.

``WORKLOADS`` maps a stage name to (description, builder); a builder creates the stage's inputs on the device
once and returns ``{case name: zero-argument callable}``.  The inputs reproduce the reference's component
benchmarks (benchmarks/bench_projection.py, bench_sh.py, bench_tilemapper.py, bench_rasterizer.py: same
generators, sizes and distributions — SURVEY.md 2 row 15 asks for the harness to reproduce, i.e. inputs and
protocol) plus BASELINE.json's config D for the two raster passes.  Protocol (benchmarks/util.py:23-37 of the
reference): warm up, then time ``iters`` back-to-back calls between two events, one
synchronisation at the end.
"""
from __future__ import annotations

import argparse
from dataclasses import replace
from typing import Callable, Dict, Tuple

import torch

from ..data_types import RasterConfig
from ..mapper.tile_mapper import map_to_tiles
from ..misc.renderer2d import project_gaussians2d
from ..perspective.projection import project_to_image
from ..rasterizer import rasterize_with_tiles
from ..rendering import ndc_depth
from ..spherical_harmonics import evaluate_sh_at
from ..testing import random_2d_gaussians, random_3d_gaussians, random_camera

Cases = Dict[str, Callable[[], object]]


def time_ms(fn: Callable[[], object], iters: int = 100, warmup: int = 10, min_seconds: float = 0.25) -> float:
  """Milliseconds per call by HIP events around ``iters`` calls (the reference's protocol, benchmarks/util.py:23-37:
  10 warm-ups, then events).  Sub-millisecond cases are additionally warmed up and repeated until ``min_seconds``
  have passed: 100 calls of a 0.1 ms case end before the clocks and the caching allocator have settled, and read
  2-5x high."""
  import time
  t0 = time.perf_counter()
  done = 0
  while done < max(1, min(warmup, iters // 4)) or time.perf_counter() - t0 < min_seconds:
    fn()
    done += 1
  torch.cuda.synchronize()
  total_ms, calls = 0.0, 0
  t0 = time.perf_counter()
  while calls < iters or time.perf_counter() - t0 < min_seconds:
    begin, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    begin.record()
    for _ in range(iters):
      fn()
    end.record()
    torch.cuda.synchronize()
    total_ms += begin.elapsed_time(end)
    calls += iters
  return total_ms / calls


def _backward_case(leaves, forward):
  """fwd + bwd of ``forward()`` (summing every floating output) w.r.t. the tensors in ``leaves``."""
  def case():
    for t in leaves:
      t.grad = None
    out = forward()
    outs = out if isinstance(out, (tuple, list)) else (out,)
    sum(o.sum() for o in outs if torch.is_tensor(o) and o.is_floating_point() and o.requires_grad).backward()
  return case


def _with_grad(tensors, wanted, body):
  """Run ``body`` with requires_grad set on exactly ``wanted`` (evaluated lazily at call time)."""
  def case():
    for t in tensors:
      t.requires_grad_(any(t is w for w in wanted))
    body()
    for t in tensors:
      t.requires_grad_(False)
  return case


def projection_cases(device, n=2_000_000, margin=0.5) -> Cases:
  camera = random_camera()
  g = random_3d_gaussians(n, camera, margin=margin).to(device)       # ~half of them outside the view
  camera = camera.to(device=device)
  cfg = RasterConfig()
  shape = list(g.shape_tensors())
  cam = [camera.T_camera_world, camera.projection]
  fwd = lambda: project_to_image(g, camera, cfg)[:2]
  bwd = _backward_case(shape + cam, fwd)
  visible = project_to_image(g, camera, cfg)[2].shape[0]
  print(f"projection: {n} gaussians, {visible} visible")
  return {'forward': lambda: _no_grad(fwd),
          'backward (gaussians)': _with_grad(shape + cam, shape, bwd),
          'backward (extrinsics)': _with_grad(shape + cam, cam[:1], bwd),
          'backward (intrinsics)': _with_grad(shape + cam, cam[1:], bwd),
          'backward (everything)': _with_grad(shape + cam, shape + cam, bwd)}


def _no_grad(fn):
  with torch.no_grad():
    return fn()


def sh_cases(device, n=1_000_000, degree=3) -> Cases:
  camera_pos = random_camera().to(device=device).camera_position.clone()
  params = torch.rand(n, 3, (degree + 1) ** 2, device=device)
  points = torch.randn(n, 3, device=device)
  indexes = torch.arange(n, device=device)
  leaves = [params, points, camera_pos]
  fwd = lambda: evaluate_sh_at(params, points, indexes, camera_pos)
  bwd = _backward_case(leaves, fwd)
  return {'forward': lambda: _no_grad(fwd),
          'backward (sh_features)': _with_grad(leaves, [params], bwd),
          'backward (all)': _with_grad(leaves, leaves, bwd)}


def _dense_2d_scene(device, n, size, scale, alpha, tile):
  g = random_2d_gaussians(n, size, num_channels=3, scale_factor=scale, alpha_range=alpha, depth_range=(0.1, 100.)).to(device)
  cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2))
  return project_gaussians2d(g), g.depths, g.feature, cfg


def tilemapper_cases(device, n=1_000_000, size=(1024, 768), scale=2.0, tile=16) -> Cases:
  p, depth, _, cfg = _dense_2d_scene(device, n, size, scale, (0.5, 1.0), tile)
  o2p, ranges = map_to_tiles(p, depth, size, cfg)
  print(f"tile_mapper: n={n} K={o2p.shape[0]} K/N={o2p.shape[0] / n:.2f} K/tile={o2p.shape[0] / ranges[..., 0].numel():.1f}")
  return {'tile_mapper': lambda: map_to_tiles(p, depth, size, cfg),
          'tile_mapper (depth16)': lambda: map_to_tiles(p, depth, size, cfg, use_depth16=True)}


def _raster_cases(p, feats, o2p, ranges, size, cfg) -> Cases:
  leaves = [p, feats]
  render = lambda c=cfg: rasterize_with_tiles(p, feats, o2p, ranges, size, c).image
  bwd = lambda c=cfg: _backward_case(leaves, lambda: render(c))
  return {'forward': lambda: _no_grad(render),
          'forward_vis': lambda: _no_grad(lambda: render(replace(cfg, compute_visibility=True))),
          'backward (features)': _with_grad(leaves, [feats], bwd()),
          'backward (gaussians)': _with_grad(leaves, [p], bwd()),
          'backward (all)': _with_grad(leaves, leaves, bwd()),
          'backward (compute_point_heuristic)': _with_grad(leaves, leaves, bwd(replace(cfg, compute_point_heuristic=True)))}


def rasterizer_cases(device, n=1_000_000, size=(1024, 768), scale=4.0, tile=16) -> Cases:
  p, depth, feats, cfg = _dense_2d_scene(device, n, size, scale, (0.75, 1.0), tile)
  o2p, ranges = map_to_tiles(p, depth, size, cfg)
  print(f"rasterizer (dense 2D): n={n} K={o2p.shape[0]} K/tile={o2p.shape[0] / ranges[..., 0].numel():.1f}")
  return _raster_cases(p, feats.contiguous(), o2p, ranges.view(-1, 2), size, cfg)


def rasterizer_config_d_cases(device, n=6_000_000, size=(2048, 2048), tile=16) -> Cases:
  camera = random_camera(image_size=size)
  g = random_3d_gaussians(n, camera, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.0).to(device)
  camera = camera.to(device=device)
  cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2))
  with torch.no_grad():
    p, depth, idx = project_to_image(g, camera, cfg)
    o2p, ranges = map_to_tiles(p, ndc_depth(depth, camera.near_plane, camera.far_plane), size, cfg)
  print(f"rasterizer (config D): n={n} K={o2p.shape[0]} K/tile={o2p.shape[0] / ranges[..., 0].numel():.1f}")
  return _raster_cases(p, g.feature[idx].contiguous(), o2p, ranges.view(-1, 2), size, cfg)


WORKLOADS: Dict[str, Tuple[str, Callable[..., Cases]]] = {
  'projection': ("2 M random 3D gaussians, random camera, about half outside the view (margin 0.5)", projection_cases),
  'sh': ("1 M points, SH degree 3, RGB", sh_cases),
  'tilemapper': ("1 M random 2D gaussians, 1024x768, scale_factor 2, alpha in (0.5, 1), tile 16", tilemapper_cases),
  'rasterizer': ("1 M random 2D gaussians, 1024x768, scale_factor 4, alpha in (0.75, 1), depth in (0.1, 100), tile 16", rasterizer_cases),
  'rasterizer_d': ("BASELINE config D raster stage: 6 M random 3D gaussians projected to 2048x2048, tile 16", rasterizer_config_d_cases),
}


def run(stage: str, iters: int = 100, device: str = 'cuda:0', seed: int = 0, **overrides) -> Dict[str, float]:
  """Build the stage's workload and time every case; returns {case: ms per call}."""
  description, builder = WORKLOADS[stage]
  torch.manual_seed(seed)
  print(f"== {stage}: {description}")
  results = {}
  for name, case in builder(torch.device(device), **overrides).items():
    results[name] = time_ms(case, iters=iters)
    print(f"  {name:36s} {results[name]:9.3f} ms   ({1e3 / results[name]:8.1f} /s)")
  return results


def main(argv=None):
  ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
  ap.add_argument('stages', nargs='*', default=[], help=f"any of {sorted(WORKLOADS)} (default: all but rasterizer_d)")
  ap.add_argument('--iters', type=int, default=100)
  ap.add_argument('--device', default='cuda:0')
  ap.add_argument('--seed', type=int, default=0)
  args = ap.parse_args(argv)
  for stage in args.stages or [s for s in WORKLOADS if s != 'rasterizer_d']:
    run(stage, iters=args.iters, device=args.device, seed=args.seed)


if __name__ == '__main__':
  main()
