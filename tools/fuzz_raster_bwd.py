#!/usr/bin/env python
"""Randomised cross-check of the float RGB raster kernels (development tool): the float32 forward (+ visibility) against
the float64 generic forward, and the two backward organisations against each other: the splat-per-lane
scan kernel (ms_raster_bwd_moments + finalize, plain and deterministic) against the pixel-per-lane kernel
(ms_raster_bwd) on random 2D scenes — image sizes that are not tile multiples, tiny and huge splats, alphas
around the thresholds, empty and crowded tiles, strips of tile rows.  Both run in float32; they must agree to
rounding (median splat < 1e-5 of the largest gradient) except for the handful of splats per scene that sit on a
float32 gate decision (see the comment at the criterion; --diag brings in the float64 kernel and the gate margins).

    python tools/fuzz_raster_bwd.py [--seeds 200] [--first 0]
"""
import argparse
import sys
from dataclasses import replace
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def main():
  p = argparse.ArgumentParser()
  p.add_argument('--seeds', type=int, default=200)
  p.add_argument('--first', type=int, default=0)
  p.add_argument('--tol', type=float, default=1e-3)
  p.add_argument('--diag', action='store_true', help='also run the float64 generic kernel and report each float32 kernel against it')
  args = p.parse_args()

  from taichi_splatting_amd import RasterConfig, _lib
  from taichi_splatting_amd.testing import random_2d_gaussians
  from taichi_splatting_amd.mapper.tile_mapper import map_to_tiles
  from taichi_splatting_amd.rasterizer.function import rasterize_with_tiles
  from taichi_splatting_amd.misc.renderer2d import project_gaussians2d

  dev = torch.device('cuda', 0)
  lib = _lib.load()
  worst, failures, most_off = 0.0, [], 0
  for seed in range(args.first, args.first + args.seeds):
    gen = torch.Generator().manual_seed(seed)
    r = lambda lo, hi: float(torch.empty(1).uniform_(lo, hi, generator=gen))
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=gen))
    tile = (8, 16, 32)[ri(0, 2)]
    w, h = ri(1, 420), ri(1, 300)
    n = int(10 ** r(0, 5.0))
    scale = 10 ** r(-1.0, 1.5)
    a_lo = r(0.01, 0.9)
    heur = ri(0, 1) == 1
    cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2), compute_point_heuristic=heur,
                       alpha_threshold=(1 / 255, 0.02, 0.2)[ri(0, 2)], clamp_max_alpha=(0.99, 0.7)[ri(0, 1)],
                       saturate_threshold=(0.9999, 0.95)[ri(0, 1)])
    torch.manual_seed(seed)
    with torch.no_grad():
      g = random_2d_gaussians(n, (w, h), num_channels=3, scale_factor=scale, alpha_range=(a_lo, min(0.999, a_lo + r(0.05, 0.6))),
                              depth_range=(0.1, 100.0)).to(dev)
      g2d, feats, depth = project_gaussians2d(g).contiguous(), g.feature.contiguous(), g.depths.contiguous()
      o2p, ranges = map_to_tiles(g2d, depth, (w, h), cfg)
      ranges2 = ranges.view(-1, 2)
      image = rasterize_with_tiles(g2d, feats, o2p, ranges2, (w, h), cfg).image
      # forward: the float32 product kernel (+ visibility) against the float64 generic kernel on the same lists
      cfg_vis = replace(cfg, compute_visibility=True, compute_point_heuristic=False)
      out32 = rasterize_with_tiles(g2d, feats, o2p, ranges2, (w, h), cfg_vis)
      out64 = rasterize_with_tiles(g2d.double(), feats.double(), o2p, ranges2, (w, h), cfg_vis)
      px_off = int(((out32.image.double() - out64.image).abs().amax(-1) > 1e-4).sum())
      px_allowed = max(8, int(2e-3 * w * h))
      vis_scale = float(out64.visibility.abs().max())
      vis_off = int(((out32.visibility.double() - out64.visibility).abs() > 1e-3 * max(vis_scale, 1e-30)).sum())
      fwd_typical = float((out32.image.double() - out64.image).abs().median())
      if not (torch.isfinite(out32.image).all() and px_off <= px_allowed and vis_off <= max(24, int(2e-3 * n)) and fwd_typical <= 1e-5
              and float((out32.image - image).abs().max()) <= 2e-6):     # with and without visibility: same blend
        tag = (f"seed {seed} forward: tile {tile} {w}x{h} n={n} K={o2p.shape[0]} scale={scale:.2f} thr={cfg.alpha_threshold:.3f} "
               f"pixels off={px_off} (allowed {px_allowed}) visibility off={vis_off} median={fwd_typical:.1e}")
        failures.append(tag)
        print("FAIL", tag, flush=True)
      grad_image = torch.rand(image.shape, generator=gen).to(dev) - 0.3
      th = (h + tile - 1) // tile
      row0 = ri(0, th - 1) if ri(0, 3) == 0 else 0            # sometimes a strip of tile rows
      row1 = ri(row0 + 1, th) if row0 > 0 else th
      cfg_c = _lib.raster_config_c(cfg)
      stream = _lib.current_stream(dev)
      gp0, gf0 = torch.zeros_like(g2d), torch.zeros_like(feats)
      he0 = torch.zeros((n, 2), device=dev) if heur else None
      _lib.check(lib.ms_raster_bwd(g2d.data_ptr(), feats.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), image.data_ptr(),
                                   grad_image.data_ptr(), w, h, 3, cfg_c, gp0.data_ptr(), gf0.data_ptr(), _lib.ptr(he0),
                                   row0, row1, 0, stream), "old")
      for det in (0, 1):
        mom = torch.zeros((n, _lib.MOMENT_ROW), device=dev, dtype=torch.int64 if det else torch.float32)
        gp1, gf1 = torch.empty_like(g2d), torch.empty_like(feats)
        he1 = torch.empty((n, 2), device=dev) if heur else None
        fexp = _lib.fixed_point_exponents(grad_image) if det else None
        _lib.check(lib.ms_raster_bwd_moments(g2d.data_ptr(), feats.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(),
                                             image.data_ptr(), grad_image.data_ptr(), w, h, cfg_c, mom.data_ptr(), det,
                                             _lib.ptr(fexp), row0, row1, stream), "moments")
        _lib.check(lib.ms_raster_moments_finalize(g2d.data_ptr(), mom.data_ptr(), det, _lib.ptr(fexp), n, gp1.data_ptr(), gf1.data_ptr(),
                                                  _lib.ptr(he1), stream), "finalize")
        torch.cuda.synchronize()
        # per splat: largest difference over its outputs, each relative to the largest gradient of that output
        pairs = [(gp1[:, k:k + 1], gp0[:, k:k + 1]) for k in range(7)] + [(gf1, gf0)]
        if heur:
          pairs += [(he1[:, 0:1], he0[:, 0:1]), (he1[:, 1:2], he0[:, 1:2])]
        per_splat = torch.zeros((n,), device=dev)
        finite = True
        for a, b in pairs:
          finite = finite and bool(torch.isfinite(a).all())
          scale_b = float(b.abs().max())
          d = (a - b).abs().amax(1)
          per_splat = torch.maximum(per_splat, d / scale_b if scale_b > 0 else torch.where(d > 0, torch.inf, 0.0))
        # Both kernels gate in float32 (alpha > alpha_threshold, T > 1 - saturate_threshold) with differently rounded
        # alphas and transmittances: a (pixel, splat) pair within ~1e-7 of a gate may fall on either side, which moves
        # that splat and the ones behind it at that pixel (checked against the float64 kernel and the oracle's gate
        # margins with --diag).  A handful of such splats per scene is expected; an error in the kernel moves whole
        # tiles.
        off = int((per_splat > (args.tol if n >= 64 else 5 * args.tol)).sum())
        allowed = max(24, int(2e-3 * n)) if n >= 64 else 2      # tiny scenes: a splat or two on a gate (seeds 110996, 202296)
        err = float(per_splat.max()) if finite else float('inf')
        typical = float(per_splat.median())
        most_off = max(most_off, off)
        worst = max(worst, err)
        if args.diag and det == 0:
          # float64 generic kernel on the same scene: which float32 kernel moved?
          p64, f64 = g2d.double(), feats.double()
          img64 = rasterize_with_tiles(p64, f64, o2p, ranges2, (w, h), cfg).image
          gp64, gf64 = torch.zeros_like(p64), torch.zeros_like(f64)
          he64 = torch.zeros((n, 2), device=dev, dtype=torch.float64) if heur else None
          _lib.check(lib.ms_raster_bwd(p64.data_ptr(), f64.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), img64.data_ptr(),
                                       grad_image.double().data_ptr(), w, h, 3, cfg_c, gp64.data_ptr(), gf64.data_ptr(),
                                       _lib.ptr(he64), row0, row1, 1, stream), "f64")
          torch.cuda.synchronize()
          def rel(a, b):
            sc = float(b.abs().max())
            return float((a.double() - b).abs().max()) / sc if sc > 0 else 0.0
          e_old = max(rel(gp0, gp64), rel(gf0, gf64))
          e_new = max(rel(gp1, gp64), rel(gf1, gf64))
          bad_old = int(((gp0.double() - gp64).abs().amax(1) > 1e-4 * gp64.abs().max()).sum())
          bad_new = int(((gp1.double() - gp64).abs().amax(1) > 1e-4 * gp64.abs().max()).sum())
          from oracle import raster as orast          # development tool: the CPU oracle explains the outliers
          margin = orast.gate_margin(p64.cpu(), ranges2.cpu(), o2p.cpu(), (w, h), cfg, tile_rows=(row0, row1))
          for name, gp in (("pixel-per-lane", gp0), ("scan", gp1)):
            off = ((gp.double() - gp64).abs().amax(1) > 1e-4 * gp64.abs().max()).nonzero().flatten().cpu()
            if off.numel():
              print(f"  {name}: splats off {off[:8].tolist()} gate margins {[f'{float(margin[i]):.1e}' for i in off[:8]]}")
          print(f"diag seed {seed}: pixel-per-lane vs f64 {e_old:.2e} ({bad_old} splats off), scan vs f64 {e_new:.2e} ({bad_new} splats off), "
                f"image f32 vs f64 max {float((image.double() - img64).abs().max()):.2e}", flush=True)
        tag = (f"seed {seed} det {det}: tile {tile} {w}x{h} n={n} K={o2p.shape[0]} scale={scale:.2f} heur={heur} rows {row0}:{row1} "
               f"thr={cfg.alpha_threshold:.3f} worst={err:.2e} median={typical:.1e} splats off={off} (allowed {allowed})")
        if not (finite and off <= allowed and (typical <= 1e-5 or (n < 64 and err <= 5e-2))):
          failures.append(tag)
          print("FAIL", tag, flush=True)
    if seed % 20 == 0:
      print(f"seed {seed}: worst so far {worst:.2e}, most splats off in one scene {most_off}, failures {len(failures)}", flush=True)
  print(f"done: {args.seeds} scenes x 2 modes, worst relative difference {worst:.3e}, most splats off in one scene {most_off}, failures {len(failures)}")
  sys.exit(1 if failures else 0)


if __name__ == '__main__':
  main()
