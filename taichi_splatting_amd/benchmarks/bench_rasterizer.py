"""Rasterizer forward / backward on a dense 2D scene (reference benchmarks/bench_rasterizer.py: 1 M random 2D
gaussians, 1024x768, scale_factor 4, alpha in (0.75, 1), depth in (0.1, 100), tile 16)."""
import argparse
from dataclasses import replace

import torch

from ..data_types import RasterConfig
from ..mapper.tile_mapper import map_to_tiles
from ..misc.renderer2d import project_gaussians2d
from ..rasterizer import rasterize_with_tiles
from ..testing import random_2d_gaussians
from .util import benchmarked


def parse_args(args=None):
  parser = argparse.ArgumentParser()
  parser.add_argument('--profile', action='store_true')
  parser.add_argument('--image_size', type=str, default='1024,768')
  parser.add_argument('--device', type=str, default='cuda:0')
  parser.add_argument('--n', type=int, default=1000000)
  parser.add_argument('--num_channels', type=int, default=3)
  parser.add_argument('--scale_factor', type=int, default=4)
  parser.add_argument('--tile_size', type=int, default=16)
  parser.add_argument('--seed', type=int, default=0)
  parser.add_argument('--iters', type=int, default=1000)
  parser.add_argument('--antialias', action='store_true')
  parser.add_argument('--debug', action='store_true')
  parser.add_argument('--skip_forward', action='store_true')
  parser.add_argument('--saturate_threshold', type=float, default=0.9999)
  parser.add_argument('--alpha_threshold', type=float, default=1 / 255)
  parser.add_argument('--pixel_stride', type=str, default='2,2')
  args = parser.parse_args(args)
  args.image_size = tuple(map(int, args.image_size.split(',')))
  args.pixel_stride = tuple(map(int, args.pixel_stride.split(',')))
  return args


def bench_rasterizer(args):
  torch.manual_seed(args.seed)
  gaussians = random_2d_gaussians(args.n, args.image_size, num_channels=args.num_channels,
                                  scale_factor=args.scale_factor, alpha_range=(0.75, 1.0),
                                  depth_range=(0.1, 100.)).to(args.device)
  config = RasterConfig(tile_size=args.tile_size, antialias=args.antialias, pixel_stride=args.pixel_stride,
                        saturate_threshold=args.saturate_threshold, alpha_threshold=args.alpha_threshold)
  gaussians2d = project_gaussians2d(gaussians)
  overlap_to_point, tile_ranges = map_to_tiles(gaussians2d, gaussians.depths, args.image_size, config)
  points_per_tile = tile_ranges[:, :, 1] - tile_ranges[:, :, 0]
  print(overlap_to_point.shape)
  print(f'scale_factor={args.scale_factor}, n={args.n}, tile_size={args.tile_size} '
        f'point_overlap={points_per_tile.sum() / args.n:.2f} tile_points={points_per_tile.float().mean():.2f}')
  print('----------------------------------------------------------')
  ranges = tile_ranges.view(-1, 2)

  def render(points, features, cfg=config):
    return rasterize_with_tiles(points, features, overlap_to_point, ranges, args.image_size, cfg)

  if not args.skip_forward:
    with torch.no_grad():
      benchmarked('forward', lambda: render(gaussians2d, gaussians.feature), profile=args.profile, iters=args.iters * 4)
      vis = replace(config, compute_visibility=True)
      benchmarked('forward_vis', lambda: render(gaussians2d, gaussians.feature, vis), profile=args.profile,
                  iters=args.iters * 4)

  def backward(points_grad, features_grad, cfg=config):
    points = gaussians2d.detach().requires_grad_(points_grad)
    features = gaussians.feature.detach().requires_grad_(features_grad)
    render(points, features, cfg).image.sum().backward()

  benchmarked('backward (features)', lambda: backward(False, True), profile=args.profile, iters=args.iters)
  benchmarked('backward (gaussians)', lambda: backward(True, False), profile=args.profile, iters=args.iters)
  benchmarked('backward (all)', lambda: backward(True, True), profile=args.profile, iters=args.iters)
  heur = replace(config, compute_point_heuristic=True)
  benchmarked('backward (compute_point_heuristic)', lambda: backward(True, True, heur), profile=args.profile,
              iters=args.iters)


def main():
  bench_rasterizer(parse_args())


if __name__ == '__main__':
  main()
