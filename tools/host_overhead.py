#!/usr/bin/env python
"""Where does an eager frame's wall time go on THIS box?  Prints, for config D: eager ms/frame, HIP-graph replay
ms/frame, kernel-time sum is left to rocprof; and for a tiny scene (GPU time ~ 0) the host-bound ms/frame = Python +
launch cost of one frame.  python tools/host_overhead.py [--n 6000000 --size 2048]"""
import argparse
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))


def scene(n, size, dev, deg=3):
  from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
  torch.manual_seed(0)
  cam = random_camera(image_size=(size, size))
  g = random_3d_gaussians(n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.0)
  g = g.replace(feature=(torch.rand(n, 3, (deg + 1) ** 2) - 0.5) * 0.5)
  return g.to(dev), cam.to(device=dev)


def main():
  p = argparse.ArgumentParser()
  p.add_argument('--n', type=int, default=6_000_000)
  p.add_argument('--size', type=int, default=2048)
  p.add_argument('--steps', type=int, default=100)
  p.add_argument('--no-gc', action='store_true', help='gc.collect(); gc.freeze(); gc.disable() before the loops')
  args = p.parse_args()
  if args.no_gc:
    import gc
    gc.collect(); gc.freeze(); gc.disable()
  from taichi_splatting_amd import RasterConfig, render_gaussians, frame
  dev = torch.device('cuda', 0)
  cfg = RasterConfig()

  def run(g, cam, steps):
    params = [t.detach().requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
    gg = g.replace(position=params[0], log_scaling=params[1], rotation=params[2], alpha_logit=params[3], feature=params[4])

    def step():
      for t in params:
        t.grad = None
      render_gaussians(gg, cam, cfg, use_sh=True).image.sum().backward()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.6:
      step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
      step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) * 1e3 / steps
    # per-frame host intervals (time between successive returns of step()): stalls show up as outliers
    marks = [time.perf_counter()]
    for _ in range(4 * steps):
      step()
      marks.append(time.perf_counter())
    torch.cuda.synchronize()
    iv = sorted((b - a) * 1e3 for a, b in zip(marks, marks[1:]))
    print(f"  frame intervals ms: min {iv[0]:.2f} median {iv[len(iv) // 2]:.2f} p90 {iv[int(len(iv) * 0.9)]:.2f} "
          f"p99 {iv[int(len(iv) * 0.99)]:.2f} max {iv[-1]:.2f}; frames > 1.3 x median: "
          f"{sum(1 for v in iv if v > 1.3 * iv[len(iv) // 2])} of {len(iv)}")
    # host time of one frame when the GPU is idle at its start (includes the wait for K = the frame's front end)
    host = []
    for _ in range(10):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      step()
      host.append((time.perf_counter() - t0) * 1e3)
      torch.cuda.synchronize()
    return eager, sorted(host)[len(host) // 2], step

  g, cam = scene(args.n, args.size, dev)
  eager, host, step = run(g, cam, args.steps)
  print(f"config: n={args.n} {args.size}^2  eager {eager:.3f} ms/frame; host returns after {host:.3f} ms (GPU idle at start)")
  del step
  g2, cam2 = scene(2000, 64, dev)
  eager2, host2, _ = run(g2, cam2, 300)
  print(f"tiny scene (2000 gaussians, 64^2): eager {eager2:.3f} ms/frame = host + launch bound; host returns after {host2:.3f} ms")


if __name__ == '__main__':
  main()
