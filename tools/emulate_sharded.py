"""Per-rank cost of the gaussian-sharded / strip-sharded step (distributed.render_sharded_step) at world
size W, emulated on ONE GPU: for each rank r in turn the step runs on r's shard with the all-to-all
replaced by device copies of exactly the buffers rank r would send / receive (pre-computed from all W
shards outside the timed region).  Everything except the xGMI transfers and the RCCL launch latency is
therefore measured: projection / SH on the shard, routing + packing, unpacking, map / sort / raster on
the strip, gradient return, projection / SH backward.  Prints one JSON line.

  python tools/emulate_sharded.py --world 8 [--n 6000000 --size 2048 --steps 5]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class EmulatedExchange:
  """exchange(send, send_counts, recv_counts, group) for rank ``rank``: returns what the real all-to-all
  would deliver, built from the recorded send buffers of all ranks."""

  def __init__(self, world):
    self.world, self.rank = world, 0
    self.sent = {}            # rank -> (rows, counts) recorded in the preparation pass
    self.record = True
    self.recv_rows = None

  def __call__(self, send, send_counts, recv_counts, group):
    w, r = self.world, self.rank
    if send.dtype == torch.int64:                      # split sizes
      if self.record:
        self.sent[r] = [None, send.view(w).clone()]
        return send.clone()                            # placeholder (symmetric)
      return torch.stack([self.sent[s][1][r] for s in range(w)]).view(w, 1)
    if self.record:
      self.sent[r][0] = send
      return send.clone()
    if send.shape[1] == self.recv_rows.shape[1]:       # forward: rows for my strip from every rank
      return self.recv_rows.clone()
    return send.new_zeros((int(sum(recv_counts)), send.shape[1]))   # backward: gradients coming home

  def prepare(self, rank):
    """Assemble the rows rank ``rank`` receives (buckets `rank` of every recorded send buffer)."""
    self.rank, self.record = rank, False
    parts = []
    for s in range(self.world):
      rows, counts = self.sent[s]
      c = counts.tolist()
      b = sum(c[:rank])
      parts.append(rows[b:b + c[rank]])
    self.recv_rows = torch.cat(parts).contiguous()


def main():
  p = argparse.ArgumentParser()
  p.add_argument('--world', type=int, default=8)
  p.add_argument('--n', type=int, default=6_000_000)
  p.add_argument('--size', type=int, default=2048)
  p.add_argument('--height', type=int, default=0)
  p.add_argument('--tile', type=int, default=16)
  p.add_argument('--sh-degree', type=int, default=3)
  p.add_argument('--seed', type=int, default=0)
  p.add_argument('--steps', type=int, default=5)
  p.add_argument('--warmup', type=int, default=2)
  p.add_argument('--cprofile', action='store_true', help='host-side profile of the timed steps (stderr)')
  p.add_argument('--ranks', type=str, default='', help='comma separated ranks to time (default: all)')
  args = p.parse_args()
  import bench
  from taichi_splatting_amd import RasterConfig, render_gaussians
  from taichi_splatting_amd.distributed import render_sharded_step, shard_range
  dev = torch.device('cuda', 0)
  cfg = RasterConfig(tile_size=args.tile, pixel_stride=(1, 1) if args.tile == 8 else (2, 2))
  g, cam = bench.make_scene(args, dev)
  W = args.world

  def timed(fn):
    """Median wall time of one step (each step bracketed by a device synchronisation)."""
    for _ in range(args.warmup):
      fn()
    times = []
    for _ in range(args.steps):
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      fn()
      torch.cuda.synchronize()
      times.append((time.perf_counter() - t0) * 1e3)
    times.sort()
    return times[len(times) // 2]

  full = g.clone().requires_grad_(True)

  def single():
    for t in (full.position, full.log_scaling, full.rotation, full.alpha_logit, full.feature):
      t.grad = None
    render_gaussians(full, cam, cfg, use_sh=True).image.sum().backward()
  t_single = timed(single)

  ex = EmulatedExchange(W)
  shards = []
  for r in range(W):
    b, e = shard_range(args.n, W, r)
    shards.append((b, g[b:e].clone().contiguous().requires_grad_(True)))
  with torch.no_grad():
    for r in range(W):                # preparation pass: record every rank's send buffer
      ex.rank, ex.record = r, True
      render_sharded_step(shards[r][1], cam, cfg, lambda img, rows: img.sum(), use_sh=True, rank=r, world_size=W,
                          backward=False, index_offset=shards[r][0], exchange=ex)
  per_rank, recv = [], []
  for r in ([int(x) for x in args.ranks.split(',')] if args.ranks else range(W)):
    ex.prepare(r)
    b, shard = shards[r]

    def step():
      for t in (shard.position, shard.log_scaling, shard.rotation, shard.alpha_logit, shard.feature):
        t.grad = None
      render_sharded_step(shard, cam, cfg, lambda img, rows: img.sum(), use_sh=True, rank=r, world_size=W,
                          index_offset=b, exchange=ex)
    if args.cprofile:
      import cProfile, pstats
      step(); torch.cuda.synchronize()
      pr = cProfile.Profile()
      pr.enable()
      for _ in range(args.steps):
        step()
      torch.cuda.synchronize()
      pr.disable()
      st = pstats.Stats(pr, stream=sys.stderr)
      st.sort_stats('tottime').print_stats(45)
      st.sort_stats('cumulative').print_stats(60)
    per_rank.append(round(timed(step), 3))
    recv.append(int(ex.recv_rows.shape[0]))
  out = {"world": W, "n": args.n, "image": [args.size, args.height or args.size], "single_gpu_ms": round(t_single, 3),
         "per_rank_ms": per_rank, "max_rank_ms": max(per_rank), "recv_splats": recv,
         "compute_only_speedup": round(t_single / max(per_rank), 2),
         "note": "xGMI transfer time and RCCL latency not included (device copies stand in for the all-to-all)"}
  print(json.dumps(out))


if __name__ == '__main__':
  main()
