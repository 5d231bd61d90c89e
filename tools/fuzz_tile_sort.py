#!/usr/bin/env python
"""Randomised cross-check of ms_tile_depth_sort (csrc/tile_sort.hip) against a stable composite sort in torch.

Every round draws a set of tile runs (lengths from a mixture that reaches every size class, the cost fallback and the
global radix path) and 32 bit keys from a mixture of shapes (uniform bits, uniform floats, binade spreads, clusters with
outliers, few distinct values, denormals / inf / NaN bit patterns, 16 bit keys) and asserts identical index lists.

    python tools/fuzz_tile_sort.py [--rounds 200] [--seed 0]
"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from taichi_splatting_amd import _lib      # noqa: E402

DEV = 'cuda:0'


def draw_lengths(gen):
  kind = int(torch.randint(0, 5, (1,), generator=gen))
  if kind == 0:
    n = torch.randint(0, 60, (int(torch.randint(1, 4000, (1,), generator=gen)),), generator=gen)
  elif kind == 1:
    n = torch.randint(500, 1100, (int(torch.randint(1, 300, (1,), generator=gen)),), generator=gen)
  elif kind == 2:
    n = torch.randint(900, 2700, (int(torch.randint(1, 120, (1,), generator=gen)),), generator=gen)
  elif kind == 3:
    n = torch.randint(2400, 5400, (int(torch.randint(1, 60, (1,), generator=gen)),), generator=gen)
  else:
    n = torch.cat([torch.randint(0, 3000, (40,), generator=gen), torch.randint(5000, 60000, (3,), generator=gen)])
  return n[torch.randperm(n.numel(), generator=gen)]


def draw_keys(k, gen):
  kind = int(torch.randint(0, 9, (1,), generator=gen))
  if kind == 0:
    return torch.randint(0, 1 << 32, (k,), dtype=torch.int64, generator=gen), 'u32'
  if kind == 1:
    return torch.rand(k, generator=gen).view(torch.int32).to(torch.int64), 'float01'
  if kind == 2:
    return torch.exp2(-30 * torch.rand(k, generator=gen)).view(torch.int32).to(torch.int64), 'binades'
  if kind == 3:
    x = 0.5 + 1e-5 * torch.rand(k, generator=gen)
    x[torch.rand(k, generator=gen) < 1e-3] = 1e-3
    x[torch.rand(k, generator=gen) < 1e-3] = 0.999
    return x.view(torch.int32).to(torch.int64), 'cluster+outliers'
  if kind == 4:
    vals = torch.rand(int(torch.randint(1, 6, (1,), generator=gen)), generator=gen)
    return vals[torch.randint(0, vals.numel(), (k,), generator=gen)].view(torch.int32).to(torch.int64), 'few values'
  if kind == 5:
    special = torch.tensor([0, 1, 0x007fffff, 0x00800000, 0x7f7fffff, 0x7f800000, 0x7fc00000, 0x80000000, 0xffffffff,
                            0x3f800000, 0xbf800000], dtype=torch.int64)
    return special[torch.randint(0, special.numel(), (k,), generator=gen)], 'special bit patterns'
  if kind == 6:
    return torch.randint(0, 65536, (k,), dtype=torch.int64, generator=gen), '16 bit'
  if kind == 7:
    return (torch.randn(k, generator=gen) * 10).view(torch.int32).to(torch.int64) & 0xffffffff, 'signed floats'
  a = torch.rand(k, generator=gen) * 1e-3 + 0.2
  b = torch.rand(k, generator=gen) * 1e-3 + 0.9
  return torch.where(torch.rand(k, generator=gen) < 0.5, a, b).view(torch.int32).to(torch.int64), 'two clusters'


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--rounds', type=int, default=200)
  ap.add_argument('--seed', type=int, default=0)
  args = ap.parse_args()
  lib = _lib.load()
  gen = torch.Generator().manual_seed(args.seed)
  stream = _lib.current_stream(torch.device(DEV))
  seen = {}
  for r in range(args.rounds):
    lengths = draw_lengths(gen).to(torch.int64)
    k = int(lengths.sum())
    if k == 0:
      continue
    keys, kind = draw_keys(k, gen)
    keys = keys & 0xffffffff
    ends = torch.cumsum(lengths, 0)
    ranges = torch.stack([ends - lengths, ends], dim=1).to(torch.int32)
    ranges[lengths == 0] = 0
    tile = torch.repeat_interleave(torch.arange(lengths.numel()), lengths)
    ids = torch.arange(k, dtype=torch.int32)
    composite = (tile << 32) | keys
    want = ids[torch.sort(composite, stable=True).indices]
    srt, o2p = composite.to(DEV), ids.to(DEV)
    scratch = torch.empty(k, dtype=torch.int64, device=DEV)
    _lib.check(lib.ms_tile_depth_sort(ranges.to(DEV).data_ptr(), lengths.numel(), srt.data_ptr(), o2p.data_ptr(),
                                      scratch.data_ptr(), stream), "ms_tile_depth_sort")
    torch.cuda.synchronize()
    bad = (o2p.cpu() != want).nonzero()
    if bad.numel():
      i = int(bad[0])
      print(f"round {r} ({kind}, {lengths.numel()} tiles, K = {k}): {bad.numel()} entries differ, first at {i} "
            f"(tile {int(tile[i])}, run length {int(lengths[int(tile[i])])})")
      return 1
    seen[kind] = seen.get(kind, 0) + 1
  print(f"{args.rounds} rounds identical; key shapes drawn: {seen}")
  return 0


if __name__ == '__main__':
  sys.exit(main())
