"""Timing helper of the component benchmarks (reference benchmarks/util.py:23-46)."""
from typing import Callable, Dict

import torch

RESULTS: Dict[str, float] = {}      # name -> ms per iteration of the last run (for tests / scripts)


def timed_benchmark(name: str, f: Callable[[], object], iters: int = 100, warmup: int = 10) -> float:
  for _ in range(min(warmup, max(iters // 4, 1))):
    f()
  torch.cuda.synchronize()
  start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  start.record()
  for _ in range(iters):
    f()
  end.record()
  torch.cuda.synchronize()
  elapsed = start.elapsed_time(end) / 1000.0
  ms = elapsed / iters * 1e3
  RESULTS[name] = ms
  print(f'{name}  {iters} iterations in {elapsed:.3f}s at {iters / elapsed:.1f} iters/sec  ({ms:.3f} ms)')
  return ms


def profiled_benchmark(name: str, f: Callable[[], object], iters: int = 100, warmup: int = 1) -> None:
  from torch.profiler import ProfilerActivity, profile
  for _ in range(warmup):
    f()
  with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(iters):
      f()
    torch.cuda.synchronize()
  print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=100))


def benchmarked(name: str, f: Callable[[], object], iters: int = 100, warmup: int = 10, profile: bool = False):
  if profile:
    return profiled_benchmark(name, f, iters=iters)
  return timed_benchmark(name, f, iters=iters, warmup=warmup)
