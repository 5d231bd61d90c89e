"""Spherical-harmonics colour forward / backward (reference benchmarks/bench_sh.py: 1 M points, degree 3)."""
import argparse

import torch

from ..spherical_harmonics import evaluate_sh_at
from ..testing import random_camera
from .util import benchmarked


def parse_args(args=None):
  parser = argparse.ArgumentParser()
  parser.add_argument('--profile', action='store_true')
  parser.add_argument('--image_size', type=str, default='1024,768')
  parser.add_argument('--device', type=str, default='cuda:0')
  parser.add_argument('--n', type=int, default=1000000)
  parser.add_argument('--seed', type=int, default=0)
  parser.add_argument('--iters', type=int, default=200)
  parser.add_argument('--degree', type=int, default=3)
  parser.add_argument('--debug', action='store_true')
  args = parser.parse_args(args)
  args.image_size = tuple(map(int, args.image_size.split(',')))
  return args


def bench_sh(args):
  torch.manual_seed(args.seed)
  device = torch.device(args.device)
  camera = random_camera().to(device=device)
  camera_pos = camera.camera_position.clone()
  sh_features = torch.rand(args.n, 3, (args.degree + 1) ** 2, device=device)
  points = torch.randn(args.n, 3, device=device)
  indexes = torch.arange(args.n, device=device)
  print(args)

  with torch.no_grad():
    benchmarked('forward', lambda: evaluate_sh_at(sh_features, points, indexes, camera_pos),
                profile=args.profile, iters=args.iters)

  def backward():
    for t in (sh_features, points, camera_pos):
      t.grad = None
    evaluate_sh_at(sh_features, points, indexes, camera_pos).sum().backward()

  sh_features.requires_grad_(True)
  benchmarked('backward (sh_features)', backward, profile=args.profile, iters=args.iters)
  points.requires_grad_(True)
  camera_pos.requires_grad_(True)
  benchmarked('backward (all)', backward, profile=args.profile, iters=args.iters)


def main():
  bench_sh(parse_args())


if __name__ == '__main__':
  main()
