"""Fit an image with 2D gaussians: the README demo of the reference
(``examples/fit_image_gaussians.py``) as a headless script on the MI355X back end.

Exercises the whole differentiable path in a training loop: ``rasterize`` forward / backward with
visibility and split heuristics (N2), the visibility-aware sparse optimiser through ``ParameterClass``
(N3), and split / prune densification with ``uniform_split_gaussians2d`` (N4).

  python -m taichi_splatting_amd.examples.fit_image_gaussians [image.npy|image.pt] --n 2000 --iters 600

Without an image file a procedural test card is fitted.  No display / OpenCV dependency: progress is
printed, ``--write`` saves the final render as a torch tensor.
"""
from __future__ import annotations

import argparse
import math
import time
from typing import Dict, Optional, Tuple

import torch

from ..data_types import Gaussians2D, RasterConfig
from ..misc.renderer2d import point_basis, project_gaussians2d, uniform_split_gaussians2d
from ..optim import ParameterClass, VisibilityAwareLaProp
from ..rasterizer import rasterize
from ..testing import random_2d_gaussians

FIELDS = ('position', 'depths', 'log_scaling', 'rotation', 'alpha_logit', 'feature')


def test_card(w: int, h: int, device) -> torch.Tensor:
  """Smooth colour gradients + discs + a checker corner: something with edges and flat areas."""
  y, x = torch.meshgrid(torch.linspace(0, 1, h, device=device), torch.linspace(0, 1, w, device=device), indexing='ij')
  img = torch.stack([x, y, 0.5 + 0.5 * torch.sin(6.28 * (x + y))], dim=-1)
  for cx, cy, r, col in ((0.3, 0.35, 0.18, (0.9, 0.2, 0.1)), (0.7, 0.6, 0.25, (0.1, 0.3, 0.9)), (0.5, 0.8, 0.1, (1.0, 1.0, 0.2))):
    mask = ((x - cx) ** 2 + (y - cy) ** 2) < r * r
    img[mask] = torch.tensor(col, device=device)
  checker = ((torch.floor(x * 16) + torch.floor(y * 16)) % 2).unsqueeze(-1)
  corner = (x > 0.75) & (y < 0.25)
  img[corner] = checker.expand(-1, -1, 3)[corner]
  return img.contiguous()


def psnr(a: torch.Tensor, b: torch.Tensor) -> float:
  return float(10 * torch.log10(1 / torch.nn.functional.mse_loss(a, b)))


def log_lerp(t: float, a: float, b: float) -> float:
  return math.exp(math.log(a) * (1 - t) + math.log(b) * t)


def as_gaussians(params: ParameterClass) -> Gaussians2D:
  return Gaussians2D(**{k: params.tensors[k] for k in FIELDS}, batch_size=(params.batch_size[0],))


def train_epoch(params: ParameterClass, ref_image: torch.Tensor, config: RasterConfig, epoch_size: int,
                opacity_reg: float = 0.0, scale_reg: float = 0.0):
  """``epoch_size`` optimiser steps; returns (last image, accumulated (prune_cost, split_score), params)."""
  h, w = ref_image.shape[:2]
  n = params.batch_size[0]
  heuristic = torch.zeros((n, 2), device=ref_image.device)
  image = None
  for _ in range(epoch_size):
    params.zero_grad()
    with torch.enable_grad():
      gaussians = as_gaussians(params)
      raster = rasterize(gaussians2d=project_gaussians2d(gaussians), depth=gaussians.depths.clamp(0, 1),
                         features=gaussians.feature, image_size=(w, h), config=config)
      image = raster.image
      scale = torch.exp(gaussians.log_scaling) / min(w, h)
      loss = (torch.nn.functional.mse_loss(image, ref_image) + opacity_reg * gaussians.opacity.mean()
              + scale_reg * scale.pow(2).mean())
      loss.backward()
    visibility = raster.visibility
    visible = (visibility > 1e-8).nonzero().squeeze(1)
    params.step(indexes=visible, visibility=visibility[visible], basis=point_basis(gaussians[visible]).detach())
    with torch.no_grad():
      params.tensors['rotation'].copy_(torch.nn.functional.normalize(params.tensors['rotation']))
      params.tensors['log_scaling'].clamp_(-5, 5)
    heuristic += raster.point_heuristic
  return image.detach(), (heuristic[:, 0], heuristic[:, 1])


def make_epochs(total_iters: int, first_epoch: int, max_epoch: int):
  """Epoch sizes growing linearly from ``first_epoch`` to ``max_epoch``, summing to ``total_iters``."""
  epochs, done = [], 0
  while done < total_iters:
    t = done / total_iters
    size = min(int(round(first_epoch + t * (max_epoch - first_epoch))), total_iters - done)
    epochs.append(max(size, 1))
    done += epochs[-1]
  return epochs


def take_n(score: torch.Tensor, n: int, descending: bool) -> torch.Tensor:
  mask = torch.zeros_like(score, dtype=torch.bool)
  if n > 0:
    mask[torch.argsort(score, descending=descending)[:n]] = True
  return mask


def split_prune(params: ParameterClass, t: float, target: int, prune_rate: float,
                heuristics: Tuple[torch.Tensor, torch.Tensor]) -> Tuple[ParameterClass, Dict[str, int]]:
  """Prune the cheapest points, split the highest-scoring ones towards ``target`` points."""
  prune_cost, split_score = heuristics
  n = params.batch_size[0]
  prune_mask = take_n(prune_cost, int(prune_rate * n * (1 - t)), descending=False)
  split_mask = take_n(split_score, max(0, (target - n) + int(prune_mask.sum())), descending=True)
  both = split_mask & prune_mask
  split_mask, prune_mask = split_mask ^ both, prune_mask ^ both

  to_split = params[split_mask] if bool(split_mask.any()) else None
  kept = params[~(split_mask | prune_mask)]
  if to_split is not None:
    children = uniform_split_gaussians2d(as_gaussians(to_split).detach(), n=2, random_axis=True)
    kept = kept.append_tensors({k: getattr(children, k) for k in FIELDS})      # fresh optimiser state
  return kept, dict(split=int(split_mask.sum()), prune=int(prune_mask.sum()))


def fit(ref_image: torch.Tensor, n: int = 1000, iters: int = 500, target: Optional[int] = None, seed: int = 0,
        tile_size: int = 16, antialias: bool = False, max_lr: float = 0.5, min_lr: float = 0.1, epoch: int = 8,
        max_epoch: int = 32, prune_rate: float = 0.025, opacity_reg: float = 1e-5, scale_reg: float = 0.1,
        verbose: bool = False):
  """Returns (final image, params, history of (iteration, psnr, n))."""
  device = ref_image.device
  h, w = ref_image.shape[:2]
  torch.manual_seed(seed)
  gaussians = random_2d_gaussians(n, (w, h), alpha_range=(0.5, 1.0), scale_factor=0.5).to(device)
  groups = dict(position=dict(lr=max_lr, type='local_vector'), log_scaling=dict(lr=0.1), rotation=dict(lr=1.0),
                alpha_logit=dict(lr=0.1), feature=dict(lr=0.025, type='vector'))
  params = ParameterClass({k: getattr(gaussians, k) for k in FIELDS}, groups, optimizer=VisibilityAwareLaProp,
                          vis_smooth=0.1, vis_beta=0.8, betas=(0.9, 0.9), eps=1e-16, bias_correction=True)
  config = RasterConfig(compute_point_heuristic=True, compute_visibility=True, tile_size=tile_size,
                        blur_cov=0.0 if antialias else 0.3, antialias=antialias,
                        pixel_stride=(1, 1) if tile_size == 8 else (2, 2))
  history, iteration, image = [], 0, None
  for epoch_size in make_epochs(iters, epoch, max_epoch):
    t = (iteration + epoch_size * 0.5) / iters
    params.set_learning_rate(position=log_lerp(t, max_lr, min_lr))
    start = time.time()
    image, heuristics = train_epoch(params, ref_image, config, epoch_size, opacity_reg, scale_reg)
    metrics = dict(psnr=psnr(ref_image, image), n=params.batch_size[0])
    if target and iteration + epoch_size < iters:
      t_points = min(math.sqrt(t * 2), 1.0)
      goal = math.ceil(params.batch_size[0] * (1 - t_points) + t_points * target)
      params, counts = split_prune(params, t, goal, prune_rate, heuristics)
      metrics.update(counts)
    iteration += epoch_size
    history.append((iteration, metrics['psnr'], metrics['n']))
    if verbose:
      torch.cuda.synchronize()
      rate = epoch_size / (time.time() - start)
      print(f"iter {iteration:5d}  " + "  ".join(f"{k}={v:.2f}" if isinstance(v, float) else f"{k}={v}"
                                                   for k, v in metrics.items()) + f"  {rate:.0f} it/s", flush=True)
  return image, params, history


def main():
  p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
  p.add_argument('image_file', nargs='?', default=None, help='.npy / .pt tensor (H, W, 3) in [0, 1] or uint8')
  p.add_argument('--size', type=int, nargs=2, default=(512, 384), help='test card size when no file is given')
  p.add_argument('--n', type=int, default=1000)
  p.add_argument('--target', type=int, default=None)
  p.add_argument('--prune', action='store_true', help='enable pruning (equivalent to --target=n)')
  p.add_argument('--iters', type=int, default=2000)
  p.add_argument('--seed', type=int, default=0)
  p.add_argument('--tile_size', type=int, default=16)
  p.add_argument('--antialias', action='store_true')
  p.add_argument('--write', type=str, default=None)
  args = p.parse_args()
  device = torch.device('cuda:0')
  if args.image_file is None:
    ref = test_card(args.size[0], args.size[1], device)
  else:
    if args.image_file.endswith('.npy'):
      import numpy as np
      ref = torch.from_numpy(np.load(args.image_file))
    else:
      ref = torch.load(args.image_file)
    ref = (ref.float() / 255 if ref.dtype == torch.uint8 else ref.float()).to(device).contiguous()
  print(f"image {ref.shape[1]}x{ref.shape[0]}")
  image, params, history = fit(ref, n=args.n, iters=args.iters, target=args.target or (args.n if args.prune else None),
                               seed=args.seed, tile_size=args.tile_size, antialias=args.antialias, verbose=True)
  print(f"final PSNR {history[-1][1]:.2f} dB with {params.batch_size[0]} gaussians")
  if args.write:
    torch.save(image.cpu(), args.write)


if __name__ == '__main__':
  main()
