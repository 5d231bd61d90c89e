"""Projection forward / backward (reference benchmarks/bench_projection.py: 2 M random 3D gaussians, half of
them outside the view with --margin 0.5)."""
import argparse
from functools import partial

import torch

from ..data_types import RasterConfig
from ..perspective import projection
from ..testing import random_3d_gaussians, random_camera
from .util import benchmarked


def parse_args(args=None):
  parser = argparse.ArgumentParser()
  parser.add_argument('--profile', action='store_true')
  parser.add_argument('--image_size', type=str, default='1024,768')
  parser.add_argument('--device', type=str, default='cuda:0')
  parser.add_argument('--n', type=int, default=2000000)
  parser.add_argument('--seed', type=int, default=0)
  parser.add_argument('--iters', type=int, default=1000)
  parser.add_argument('--margin', type=float, default=0.5, help="controls random points (non visible) margin")
  parser.add_argument('--debug', action='store_true')
  args = parser.parse_args(args)
  args.image_size = tuple(map(int, args.image_size.split(',')))
  return args


def bench_projection(args):
  torch.manual_seed(args.seed)
  with torch.no_grad():
    camera_params = random_camera()
    gaussians = random_3d_gaussians(args.n, camera_params, margin=args.margin)
    config = RasterConfig()
    gaussians, camera_params = gaussians.to(args.device), camera_params.to(args.device)
    _, _, vis_idx = projection.project_to_image(gaussians, camera_params, config)
    print(args)
    print(f"benchmarking {args.n} points ({vis_idx.shape[0]} visible) points")
    benchmarked('forward', partial(projection.project_to_image, gaussians, camera_params, config),
                profile=args.profile, iters=args.iters)

  def project_backward():
    for t in (gaussians.position, gaussians.log_scaling, gaussians.rotation, gaussians.alpha_logit,
              camera_params.T_camera_world, camera_params.projection):
      t.grad = None
    points, depth, _ = projection.project_to_image(gaussians, camera_params, config)
    (points.sum() + depth.sum()).backward()

  gaussians.requires_grad_(True)
  benchmarked('backward (gaussians)', project_backward, profile=args.profile, iters=args.iters)
  gaussians.requires_grad_(False)
  camera_params.T_camera_world.requires_grad_(True)
  benchmarked('backward (extrinsics)', project_backward, profile=args.profile, iters=args.iters)
  camera_params.T_camera_world.requires_grad_(False)
  camera_params.projection.requires_grad_(True)
  benchmarked('backward (intrinsics)', project_backward, profile=args.profile, iters=args.iters)
  gaussians.requires_grad_(True)
  camera_params.T_camera_world.requires_grad_(True)
  benchmarked('backward (everything)', project_backward, profile=args.profile, iters=args.iters)


def main():
  bench_projection(parse_args())


if __name__ == '__main__':
  main()
