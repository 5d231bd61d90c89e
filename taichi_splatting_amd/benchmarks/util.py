"""``benchmarked(name, f, iters, warmup, profile)`` under the reference's module path (``benchmarks/util.py``) for
scripts that import it: a thin front of the harness's event timer (``components.time_ms``).  With ``profile=True``
the device time of the calls is broken down per kernel instead."""
from collections import defaultdict

import torch

from .components import time_ms


def benchmarked(name, f, iters=100, warmup=10, profile: bool = False):
  if not profile:
    ms = time_ms(f, iters=iters, warmup=warmup)
    print(f"{name}: {ms:.3f} ms per call over {iters} calls ({1e3 / ms:.1f} calls/s)")
    return ms
  return kernel_breakdown(name, f, iters=iters, warmup=warmup)


def kernel_breakdown(name, f, iters=20, warmup=2, rows=20):
  """microseconds of device time per call and per kernel, largest first (torch.profiler's kernel events)"""
  import torch.profiler as tp
  for _ in range(max(warmup, 1)):
    f()
  torch.cuda.synchronize()
  with tp.profile(activities=[tp.ProfilerActivity.CUDA]) as session:
    for _ in range(iters):
      f()
    torch.cuda.synchronize()
  per_kernel = defaultdict(float)
  for event in session.events():
    if event.device_type == torch.autograd.DeviceType.CUDA:
      per_kernel[event.name] += event.device_time if hasattr(event, 'device_time') else event.cuda_time
  total = sum(per_kernel.values())
  print(f"{name}: {total / iters:.1f} us of kernels per call")
  for kernel, us in sorted(per_kernel.items(), key=lambda kv: -kv[1])[:rows]:
    print(f"  {us / iters:10.1f} us  {100 * us / max(total, 1e-9):5.1f} %  {kernel[:100]}")
  return per_kernel


timed_benchmark = lambda name, f, iters=100, warmup=10: benchmarked(name, f, iters, warmup, False)
profiled_benchmark = lambda name, f, iters=100, warmup=1: benchmarked(name, f, iters, warmup, True)
