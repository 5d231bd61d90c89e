#!/usr/bin/env python
"""A/B of the two float RGB raster backward organisations on one scene (development tool):
pixel-per-lane kernels of raster_fast.hip (ms_raster_bwd) vs splat-per-lane scan kernel of
raster_bwd_scan.hip (ms_raster_bwd_moments + ms_raster_moments_finalize).  Prints agreement and times.

    python tools/ab_raster_bwd.py [--n 6000000 --size 2048 --tile 16] [--heur] [--dense]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
  p = argparse.ArgumentParser()
  p.add_argument('--n', type=int, default=6_000_000)
  p.add_argument('--size', type=int, default=2048)
  p.add_argument('--tile', type=int, default=16)
  p.add_argument('--heur', action='store_true')
  p.add_argument('--dense', action='store_true', help='reference bench_rasterizer workload (1M 2D, 1024x768, scale 4)')
  p.add_argument('--iters', type=int, default=10)
  args = p.parse_args()

  from taichi_splatting_amd import RasterConfig, _lib
  from taichi_splatting_amd.testing import random_camera, random_3d_gaussians, random_2d_gaussians
  from taichi_splatting_amd.perspective.projection import project_to_image
  from taichi_splatting_amd.spherical_harmonics import evaluate_sh_at
  from taichi_splatting_amd.mapper.tile_mapper import map_to_tiles
  from taichi_splatting_amd.rasterizer.function import rasterize_with_tiles
  from taichi_splatting_amd.rendering import ndc_depth
  from taichi_splatting_amd.misc.renderer2d import project_gaussians2d

  dev = torch.device('cuda', 0)
  lib = _lib.load()
  cfg = RasterConfig(tile_size=args.tile, pixel_stride=(1, 1) if args.tile == 8 else (2, 2),
                     compute_point_heuristic=args.heur)
  torch.manual_seed(0)
  with torch.no_grad():
    if args.dense:
      size = (1024, 768)
      g = random_2d_gaussians(1_000_000, size, num_channels=3, scale_factor=4.0, alpha_range=(0.75, 1.0),
                              depth_range=(0.1, 100.0)).to(dev)
      g2d, feats, depth = project_gaussians2d(g), g.feature.contiguous(), g.depths.contiguous()
      o2p, ranges = map_to_tiles(g2d, depth, size, cfg)
      w, h = size
    else:
      size = (args.size, args.size)
      cam = random_camera(image_size=size)
      g = random_3d_gaussians(args.n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.0)
      g = g.replace(feature=(torch.rand(args.n, 3, 16) - 0.5) * 0.5).to(dev)
      cam = cam.to(device=dev)
      g2d, depths, idx = project_to_image(g, cam, cfg)
      feats = evaluate_sh_at(g.feature, g.position, idx, cam.camera_position)
      o2p, ranges = map_to_tiles(g2d, ndc_depth(depths, cam.near_plane, cam.far_plane), size, cfg)
      w, h = size
    ranges2 = ranges.view(-1, 2)
    image = rasterize_with_tiles(g2d, feats, o2p, ranges2, size, cfg).image
    torch.manual_seed(1)
    grad_image = torch.rand_like(image) + 0.5
    n = g2d.shape[0]
    print(f"V={n} K={o2p.shape[0]} tiles={ranges2.shape[0]} K/tile={o2p.shape[0] / ranges2.shape[0]:.1f}")

    cfg_c = _lib.raster_config_c(cfg)
    stream = _lib.current_stream(dev)
    th = (h + cfg.tile_size - 1) // cfg.tile_size

    gp0, gf0 = torch.zeros_like(g2d), torch.zeros_like(feats)
    he0 = torch.zeros((n, 2), device=dev) if args.heur else None

    def old():
      _lib.check(lib.ms_raster_bwd(g2d.data_ptr(), feats.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), image.data_ptr(),
                                   grad_image.data_ptr(), w, h, 3, cfg_c, gp0.data_ptr(), gf0.data_ptr(), _lib.ptr(he0),
                                   0, th, 0, stream), "old")

    mom = torch.zeros((n, _lib.MOMENT_ROW), device=dev)
    gp1, gf1 = torch.empty_like(g2d), torch.empty_like(feats)
    he1 = torch.empty((n, 2), device=dev) if args.heur else None

    def new_main():
      _lib.check(lib.ms_raster_bwd_moments(g2d.data_ptr(), feats.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(),
                                           image.data_ptr(), grad_image.data_ptr(), w, h, cfg_c, mom.data_ptr(), 0, None, 0, th,
                                           stream), "new")

    def new_fin():
      _lib.check(lib.ms_raster_moments_finalize(g2d.data_ptr(), mom.data_ptr(), 0, None, n, gp1.data_ptr(), gf1.data_ptr(),
                                                _lib.ptr(he1), stream), "fin")

    import ctypes
    stats_fn = getattr(lib, 'ms_debug_scan_stats', None)       # only in -DMS_SCAN_STATS builds (tools/build_variant.sh)
    if stats_fn is not None:
      stats_fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
      stats_fn(None, 1)
    old(); new_main(); new_fin()
    torch.cuda.synchronize()
    if stats_fn is not None:
      out = (ctypes.c_ulonglong * 12)()
      stats_fn(ctypes.cast(out, ctypes.c_void_p), 1)
      visits, hits, chunks, lanes, steps, pairs = [int(v) for v in out[:6]]
      print(f"stats: (wave,batch) visits={visits} (sub-patch,splat) hits={hits} ({hits / o2p.shape[0]:.2f}/overlap) "
            f"chunks={chunks} fill={lanes / max(chunks, 1):.1f}/64 steps={steps} ({steps / max(chunks, 1):.1f}/chunk) "
            f"contributing pairs={pairs} ({pairs / max(steps, 1):.1f}/step)")
      print(f"stats: chunks of <= 16 splats {int(out[10])} ({int(out[10]) / max(chunks, 1):.3f}), of 17..32 splats {int(out[11])} ({int(out[11]) / max(chunks, 1):.3f})")
      rows_fn = getattr(lib, 'ms_debug_rows_stats', None)      # raster_bwd_rows.hip (round 4) counts separately
      if rows_fn is not None:
        rows_fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
        ro = (ctypes.c_ulonglong * 12)()
        rows_fn(ctypes.cast(ro, ctypes.c_void_p), 1)
        if ro[0]:
          passes, hits, chunks, lanes, steps, pairs, phits = [int(v) for v in ro[:7]]
          print(f"rows stats: passes={passes} (patch,splat) hits={phits} ({phits / o2p.shape[0]:.2f}/overlap) (quad,splat) hits={hits} "
                f"({hits / o2p.shape[0]:.2f}/overlap) chunks={chunks} fill={lanes / max(chunks, 1):.1f}/64 pixel steps={steps} "
                f"({steps / max(chunks, 1):.2f}/chunk) contributing pairs={pairs} ({pairs / max(steps, 1):.1f}/step)")
      if out[9]:
        print(f"stats: wave balance inside a workgroup: chunks run {int(out[8])}, slots until the batch barrier {int(out[9])} "
              f"-> {int(out[8]) / int(out[9]):.3f} of the barrier-to-barrier wave time is blend work")

    def report(name, a, b):
      d = (a - b).abs()
      scale = b.abs().max().item()
      rel = d / (b.abs() + 1e-3 * scale)
      print(f"  {name}: max|d|={d.max().item():.3e} (scale {scale:.3e})  q99.9 rel={rel.flatten().float().quantile(0.999).item() if rel.numel() < 16_000_000 else rel.flatten()[:16_000_000].quantile(0.999).item():.3e}"
            f"  nan={int(torch.isnan(a).sum())}")
    names = ['mean.x', 'mean.y', 'axis.x', 'axis.y', 'sigma.x', 'sigma.y', 'alpha']
    for k in range(7):
      report(names[k], gp1[:, k], gp0[:, k])
    report('features', gf1, gf0)
    if args.heur:
      report('heur0', he1[:, 0], he0[:, 0]); report('heur1', he1[:, 1], he0[:, 1])

    def time_ms(fn, pre=None):
      for _ in range(2):
        if pre: pre()
        fn()
      torch.cuda.synchronize()
      tot = 0.0
      for _ in range(args.iters):
        if pre: pre()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
      return tot / args.iters

    print(f"old ms_raster_bwd           : {time_ms(old):.3f} ms")
    print(f"new ms_raster_bwd_moments   : {time_ms(new_main, pre=lambda: mom.zero_()):.3f} ms")
    print(f"new ms_raster_moments_final : {time_ms(new_fin):.3f} ms")


if __name__ == '__main__':
  main()
