"""``benchmarked(name, f, iters, warmup, profile)`` of the reference's ``benchmarks/util.py:23-46`` on top of the
harness's event timer (``components.time_ms``); ``profile=True`` prints a per-kernel table from ``torch.profiler``."""
import torch

from .components import time_ms


def timed_benchmark(name, f, iters=100, warmup=10):
  ms = time_ms(f, iters=iters, warmup=warmup)
  print(f"{name}  {iters} iterations at {ms:.3f} ms each, {1e3 / ms:.1f} iters/sec")
  return ms


def profiled_benchmark(name, f, iters=100, warmup=1):
  from torch.profiler import ProfilerActivity, profile
  for _ in range(warmup):
    f()
  with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(iters):
      f()
    torch.cuda.synchronize()
  print(name)
  print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=100))


def benchmarked(name, f, iters=100, warmup=10, profile: bool = False):
  return (profiled_benchmark if profile else timed_benchmark)(name, f, iters, warmup)
