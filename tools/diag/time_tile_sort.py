"""Throughput of ms_tile_depth_sort by run length (which path of csrc/tile_sort.hip a run takes):
  python tools/diag/time_tile_sort.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
from taichi_splatting_amd import _lib

DEV = 'cuda:0'
lib = _lib.load()
stream = _lib.current_stream(torch.device(DEV))
for tiles, n, kind in [(16384, 780, 'float01'), (2048, 2000, 'float01'), (1024, 4000, 'float01'), (512, 8000, 'float01'),
                       (64, 20000, 'float01'), (16, 100000, 'float01'), (1, 1000000, 'float01'), (2048, 2000, 'two clusters'),
                       (512, 8000, '16 bit')]:
  k = tiles * n
  torch.manual_seed(0)
  if kind == 'float01':
    keys = torch.rand(k).view(torch.int32).to(torch.int64)
  elif kind == '16 bit':
    keys = torch.randint(0, 65536, (k,), dtype=torch.int64)
  else:
    keys = torch.where(torch.rand(k) < 0.5, 0.2 + 1e-4 * torch.rand(k), 0.9 + 1e-4 * torch.rand(k)).view(torch.int32).to(torch.int64)
  tile = torch.arange(tiles).repeat_interleave(n)
  composite = ((tile << 32) | keys).to(DEV)
  ids = torch.arange(k, dtype=torch.int32, device=DEV)
  ends = torch.arange(1, tiles + 1, dtype=torch.int32) * n
  ranges = torch.stack([ends - n, ends], dim=1).to(DEV)
  scratch = torch.empty(k, dtype=torch.int64, device=DEV)
  times = []
  for _ in range(4):
    srt, o2p = composite.clone(), ids.clone()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _lib.check(lib.ms_tile_depth_sort(ranges.data_ptr(), tiles, srt.data_ptr(), o2p.data_ptr(), scratch.data_ptr(), stream), "sort")
    b.record(); torch.cuda.synchronize()
    times.append(a.elapsed_time(b))
  print(f"{tiles:6d} runs x {n:8d} ({kind:12s}) K = {k / 1e6:6.2f} M: {min(times):8.3f} ms = {min(times) * 1e6 / k:7.2f} ns per entry")
