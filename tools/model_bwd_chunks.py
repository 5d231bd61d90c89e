#!/usr/bin/env python
"""CPU model of the raster backward's chunk structure (raster_bwd_scan.hip) on a config-D-density scene.

Counts, for a scene drawn with bench.py's generator, what the splat-per-lane kernel would run under different
list granularities WITHOUT running it: (sub-rectangle, splat) hits with the conservative oriented-box test and with
the exact "some pixel centre passes the blend gate" test, lists per (tile, batch, wave), and the number of 64-lane
chunks when a list is consumed (a) in whole 64-lane chunks per list (the round-2/3 kernel) or (b) packed by 16-lane
DPP rows (lists of one wave share chunks, a list occupies ceil(n / 16) rows).  The VALU estimate is
chunks x (overhead + pair steps x per-pair cost).

    python tools/model_bwd_chunks.py [--n 300000 --size 888 --batch 268 --cap 128]
"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--n', type=int, default=300000)
  ap.add_argument('--size', type=int, default=888)
  ap.add_argument('--batch', type=int, default=268)
  ap.add_argument('--target', type=int, default=256)
  ap.add_argument('--cap', type=int, default=128)
  args = ap.parse_args()

  from oracle import mapper as omap, projection as oproj
  from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
  from taichi_splatting_amd import RasterConfig
  torch.manual_seed(0)
  size = (args.size, args.size)
  cam = random_camera(image_size=size)
  g = random_3d_gaussians(args.n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.0)
  cfg = RasterConfig()
  points, depths, idx = oproj.apply(g.position, g.log_scaling, g.rotation, g.alpha_logit, cam.T_camera_world,
                                    cam.projection, size, cam.depth_range, cfg.blur_cov, cfg.clamp_margin, cfg.alpha_threshold)
  ndc = oproj.ndc_depth(depths, *cam.depth_range)
  pts = points.numpy().astype(np.float32)
  o2p, ranges, _ = omap.map_to_tiles(pts, ndc.numpy(), size, 16, cfg.alpha_threshold)
  ranges = ranges.reshape(-1, 2)
  K = o2p.shape[0]
  tiles_wide = (args.size + 15) // 16
  print(f"V = {pts.shape[0]}, K = {K}, tiles = {ranges.shape[0]}, overlaps per tile = {K / ranges.shape[0]:.0f}")

  # per overlap: tile, position in the tile's list, batch (equal batches as in the kernel)
  tile_of = np.repeat(np.arange(ranges.shape[0]), ranges[:, 1] - ranges[:, 0])
  pos = np.arange(K) - ranges[tile_of, 0]
  total = ranges[:, 1] - ranges[:, 0]
  nb = np.maximum((total + args.target // 2) // args.target, 1)
  bsz = (total + nb - 1) // nb
  over = bsz > args.batch
  nb2 = (total + args.batch - 1) // args.batch
  nb = np.where(over, nb2, nb)
  bsz = np.maximum((total + nb - 1) // np.maximum(nb, 1), 1)
  batch = pos // bsz[tile_of]
  print(f"batches per tile = {nb.mean():.2f}")

  p = pts[o2p].astype(np.float64)
  mx, my, ax, ay, sx, sy, alpha = p.T
  ox = (tile_of % tiles_wide) * 16.0
  oy = (tile_of // tiles_wide) * 16.0
  thr = cfg.alpha_threshold

  # exact contribution masks (K, 16, 16): alpha g > threshold at the pixel centre, inside the image
  ys, xs = np.meshgrid(np.arange(16) + 0.5, np.arange(16) + 0.5, indexing='ij')
  contrib = np.zeros((K, 16, 16), dtype=bool)
  step = 100000
  for s in range(0, K, step):
    e = slice(s, s + step)
    dx = ox[e, None, None] + xs[None] - mx[e, None, None]
    dy = oy[e, None, None] + ys[None] - my[e, None, None]
    X = (ax[e, None, None] * dx + ay[e, None, None] * dy) / sx[e, None, None]
    Y = (-ay[e, None, None] * dx + ax[e, None, None] * dy) / sy[e, None, None]
    a = alpha[e, None, None] * np.exp(-0.5 * (X * X + Y * Y))
    inside = (ox[e, None, None] + xs[None] < args.size) & (oy[e, None, None] + ys[None] < args.size)
    contrib[e] = (a > thr) & inside
  print(f"contributing (pixel, splat) pairs = {contrib.sum()} = {contrib.sum() / K:.1f} per overlap")

  gs = np.sqrt(2.0 * np.log(alpha / thr)) * 1.001
  v1x, v1y, v2x, v2y = ax * sx * gs, ay * sx * gs, -ay * sy * gs, ax * sy * gs
  ex = (np.sqrt(v1x ** 2 + v2x ** 2) + 0.01) * 1.002
  ey = (np.sqrt(v1y ** 2 + v2y ** 2) + 0.01) * 1.002
  R = gs * 1.002
  A, B, C, D = ax / sx, ay / sx, -ay / sy, ax / sy

  def obb_hit(x0, y0, w, h):
    """scan_rect_hit() on the rectangle of pixel centres of the w x h pixel block at tile-local (x0, y0)"""
    rcx, rcy, hx, hy = ox + x0 + w / 2.0, oy + y0 + h / 2.0, (w - 1) / 2.0, (h - 1) / 2.0
    dx, dy = rcx - mx, rcy - my
    hit = (np.abs(dx) <= ex + hx) & (np.abs(dy) <= ey + hy)
    p1, e1 = A * dx + B * dy, np.abs(A) * hx + np.abs(B) * hy
    p2, e2 = C * dx + D * dy, np.abs(C) * hx + np.abs(D) * hy
    return hit & (np.abs(p1) - e1 <= R) & (np.abs(p2) - e2 <= R)

  # level 1: 8x8 patches (wave = patch), conservative test (as the kernel)
  patch_hit = np.stack([obb_hit((wv % 2) * 8, (wv // 2) * 8, 8, 8) for wv in range(4)], axis=1)     # (K, 4)
  print(f"(8x8 patch, splat) hits = {patch_hit.sum()} = {patch_hit.sum() / K:.2f} per overlap")

  key_tb = tile_of * 64 + batch                    # (tile, batch)
  n_tb = int(key_tb.max()) + 1
  pc = np.stack([np.bincount(key_tb[patch_hit[:, wv]], minlength=n_tb) for wv in range(4)], axis=1)   # patch-list lengths
  active = pc > 0
  print(f"(wave, batch) passes with hits = {active.sum()}, mean patch list = {pc[active].mean():.1f}, over CAP: {(pc > args.cap).mean() * 100:.2f} %")

  O, PAIR = 60, 92
  results = []
  for (w, h, name) in ((4, 4, '4x4'), (4, 2, '4x2'), (2, 4, '2x4'), (2, 2, '2x2'), (4, 1, '4x1'), (2, 1, '2x1')):
    nx, ny = 8 // w, 8 // h
    nq = nx * ny
    pair_steps = (w * h) // 2
    for exact in (False, True):
      counts = np.zeros((n_tb, 4, nq), dtype=np.int64)
      hits = 0
      for wv in range(4):
        for q in range(nq):
          x0, y0 = (wv % 2) * 8 + (q % nx) * w, (wv // 2) * 8 + (q // nx) * h
          if exact:
            hq = contrib[:, y0:y0 + h, x0:x0 + w].any(axis=(1, 2))
          else:
            hq = obb_hit(x0, y0, w, h) & patch_hit[:, wv]
          hits += int(hq.sum())
          counts[:, wv, q] = np.bincount(key_tb[hq], minlength=n_tb)
      chunks_list = ((counts + 63) // 64).sum()                       # one list at a time, 64-lane chunks
      rows = ((counts + 15) // 16).sum(axis=2)                         # 16-lane rows per (batch, wave)
      chunks_rows = ((rows + 3) // 4).sum()
      chunks_ideal = rows.sum() / 4.0
      cover = contrib.sum() / max(hits * w * h, 1)
      valu_list = chunks_list * (O + pair_steps * PAIR)
      valu_rows = chunks_rows * (O + 10 + pair_steps * (PAIR + 5))
      results.append((name, exact, hits, cover, chunks_list, valu_list, chunks_rows, valu_rows))
      print(f"{name} {'exact' if exact else 'obb  '}: hits {hits / K:.2f}/overlap, pixel coverage {cover:.2f}; "
            f"per-list chunks {chunks_list / K * 1e3:.1f}/k-overlap fill {hits / chunks_list:.1f} VALU {valu_list / K:.1f}/overlap; "
            f"row-packed chunks {chunks_rows / K * 1e3:.1f} (ideal {chunks_ideal / K * 1e3:.1f}) fill {hits / chunks_rows:.1f} VALU {valu_rows / K:.1f}/overlap")
  base = results[0][5]
  print("\nrelative blend VALU vs today's organisation (4x4, obb, per-list chunks):")
  for name, exact, hits, cover, cl, vl, cr, vr in results:
    print(f"  {name} {'exact' if exact else 'obb  '}: per-list {vl / base:.2f}   row-packed {vr / base:.2f}")


if __name__ == '__main__':
  main()
