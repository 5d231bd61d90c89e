#!/usr/bin/env python
"""Quantifies the deliberate deviation from the reference's in-group loop bound (SURVEY.md fact 8;
rasterizer/forward.py:86-89, backward.py:138-141) on a config-D-like tile population, CPU oracle only.

The reference re-blends stale shared-memory entries after the valid splats of a tile's last, partially filled
group; this library visits every splat once.  Reported: pixel delta between the two forwards, and the gradient
delta caused by the polluted saved image (the stale entries' own gradients are dropped by the reference,
backward.py:214, and come after every valid splat, so nothing else changes).

    python tools/fact8_deviation.py [--tiles 6 --per-tile 780 --seed 0]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import mapper as omap, projection as oproj, raster as orast  # noqa: E402
from taichi_splatting_amd import RasterConfig                           # noqa: E402
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians   # noqa: E402


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--tiles', type=int, default=8, help='image = tiles x tiles tiles of 16x16')
  ap.add_argument('--per-tile', type=float, default=780.0, help='target overlaps per tile (config D: 779)')
  ap.add_argument('--seed', type=int, default=0)
  args = ap.parse_args()
  side = args.tiles * 16
  size = (side, side)
  torch.manual_seed(args.seed)
  # config D = random_3d_gaussians(6M, 2048^2): the generator sizes the gaussians as w / sqrt(n) pixels, so the
  # same generator at the same gaussians-per-pixel density (1.43) reproduces its per-tile population
  n = int(round(6_000_000 * (side / 2048) ** 2 * args.per_tile / 779.0))
  cam = random_camera(image_size=size)
  g3 = random_3d_gaussians(n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.0)
  rc = RasterConfig()
  p, depth, idx = oproj.apply(g3.position.double(), g3.log_scaling.double(), g3.rotation.double(), g3.alpha_logit.double(),
                              cam.T_camera_world.double(), cam.projection.double(), size, cam.depth_range, rc.blur_cov,
                              rc.clamp_margin, rc.alpha_threshold)
  f = g3.feature.double()[idx]
  ndc = oproj.ndc_depth(depth, *cam.depth_range)
  o2p, ranges, _ = omap.map_to_tiles(p.float().numpy(), ndc.float().numpy(), size, 16)
  o2p, ranges = torch.from_numpy(o2p), torch.from_numpy(ranges)
  counts = (ranges[..., 1] - ranges[..., 0]).reshape(-1).float()
  print(f"{n} gaussians, {side}x{side}, K={o2p.shape[0]} (K/N {o2p.shape[0] / n:.2f}), per tile mean {counts.mean():.0f} "
        f"min {int(counts.min())} max {int(counts.max())}")
  cfg = orast.Cfg()
  img, alpha, _ = orast.forward(p, f, ranges, o2p, size, cfg)
  img_r, alpha_r, _ = orast.forward(p, f, ranges, o2p, size, cfg, emulate_reference_loop_bound=True)
  d = (img_r - img).abs().max(-1).values
  print(f"forward image: max |delta| {d.max():.3e}, mean {d.mean():.3e}, pixels with |delta| > 1e-4: {(d > 1e-4).float().mean() * 100:.2f} %, "
        f"mean final alpha {alpha.mean():.4f} (transmittance left for the stale tail {1 - alpha.mean():.4f})")
  G = torch.ones_like(img)
  gp, gf, _ = orast.backward(p, f, ranges, o2p, img, G, size, cfg)
  gp_r, gf_r, _ = orast.backward(p, f, ranges, o2p, img_r, G, size, cfg)     # reference: same walk, polluted saved image
  for name, a, b in (('grad gaussians2d', gp_r, gp), ('grad features', gf_r, gf)):
    scale = float(b.abs().max())
    dd = (a - b).abs()
    print(f"{name}: max |delta| {dd.max():.3e} ({dd.max() / scale:.3e} of the largest gradient), mean {dd.mean():.3e}")


if __name__ == '__main__':
  main()
