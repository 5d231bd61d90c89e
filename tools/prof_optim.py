#!/usr/bin/env python
"""Profiling driver for the optimiser step (VisibilityAwareAdam over the five parameter groups of a 3D gaussian model:
59 floats per gaussian) on config D's size, for rocprofv3 --kernel-trace / --pmc runs.

    python tools/prof_optim.py [n] [steps] [dense]
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taichi_splatting_amd.optim import ParameterClass, VisibilityAwareAdam   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dense = len(sys.argv) > 3 and sys.argv[3] == 'dense'
dev = 'cuda:0'
torch.manual_seed(0)
tensors = dict(position=torch.randn(n, 3, device=dev), log_scaling=torch.randn(n, 3, device=dev), rotation=torch.randn(n, 4, device=dev),
               alpha_logit=torch.randn(n, 1, device=dev), feature=torch.randn(n, 3, 16, device=dev))
groups = dict(position=dict(lr=1e-4), log_scaling=dict(lr=5e-3), rotation=dict(lr=1e-3), alpha_logit=dict(lr=5e-2), feature=dict(lr=2.5e-3))
params = ParameterClass(tensors, groups, optimizer=VisibilityAwareAdam, vis_beta=0.8, vis_smooth=0.1, betas=(0.9, 0.999), eps=1e-16)
for t in params.tensors.values():
  t.grad = torch.randn_like(t)
vis = torch.rand(n, device=dev) + 0.01
idx = torch.arange(n, device=dev)
torch.cuda.synchronize()
for _ in range(steps):
  if dense:
    params.step(indexes=None, visibility=vis)
  else:
    params.step(indexes=idx, visibility=vis)
torch.cuda.synchronize()
alg = n * 59 * 28 + n * (8 + 36)
print(f"n={n} steps={steps} mode={'dense' if dense else 'indexed'} algorithmic_bytes_per_step={alg}", flush=True)
