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



# ---- link model (a separate line: the emulation itself charges no link time) ------------------------------------------
# Every rank has one xGMI link to each of the 7 other GPUs of the node (MI355X_MICROARCH.md: 7 links x ~153 GB/s
# bidirectional per GPU, i.e. ~64 GB/s usable per link and direction for large messages); LINK_GBS is the sustained
# one-direction rate this model charges, LAUNCH_US what one RCCL collective costs before its first byte moves.
LINK_GBS = 50.0
LAUNCH_US = 20.0


def link_model(world, sent_bytes_per_peer_by_collective, compute_ms, single_ms):
  """sent_bytes_per_peer_by_collective: for each collective of a step, the bytes a rank sends to EACH peer over that
  peer's own link (all-to-all: its bucket; ring-free reduce-scatter / all-gather over direct links: the peer's shard).
  All peers' links run in parallel, so a collective takes bytes_per_peer / LINK_GBS + LAUNCH_US."""
  times = [b / (LINK_GBS * 1e9) * 1e3 + LAUNCH_US * 1e-3 for b in sent_bytes_per_peer_by_collective]
  total = sum(times)
  return {"assumes": f"{LINK_GBS:g} GB/s per link and direction, all {world - 1} links of a rank busy at once, {LAUNCH_US:g} us per "
                     "collective, NOTHING overlapped with compute",
          "collective_ms": [round(t, 4) for t in times], "link_ms_per_step": round(total, 4),
          "rank_step_ms_with_links": round(compute_ms + total, 4),
          "speedup_with_links": round(single_ms / (compute_ms + total), 2)}


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
  p.add_argument('--static', action='store_true',
                 help='the sync-free step of taichi_splatting_amd/sharded.py (ShardedStep: fixed buckets, frame executor), '
                      'timed eagerly and as a HIP-graph replay')
  p.add_argument('--strips', action='store_true',
                 help='the north_star partition (sharded.StripStep: replicated gaussians, tile-row strips, reduce-scatter + '
                      'all-gather of the 2D-boundary gradients), the collective replaced by device copies of its buffers')
  p.add_argument('--out', type=str, default='', help='also write the JSON line to this file (profiles/emul_*.json)')
  args = p.parse_args()
  if args.strips:
    return main_strips(args)
  if args.static:
    return main_static(args)
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
  f_ch = 3
  out["exchanged_bytes_per_rank"] = {"forward": W * cap * (9 + f_ch) * 4, "backward": W * cap * (7 + f_ch) * 4,
                                     "off_chip_fraction": round((W - 1) / W, 3)}
  out["link_model"] = link_model(W, [cap * (9 + f_ch) * 4, cap * (7 + f_ch) * 4], max(per_rank_graph), t_single)
  emit(args, out)


def emit(args, out):
  line = json.dumps(out)
  print(line)
  if args.out:
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, 'w') as f:
      f.write(line + '\n')


def main_strips(args):
  """StripStep per rank on one GPU.  Every rank holds all gaussians; rank r renders strip r.  The reduce-scatter +
  all-gather of the (rows, 7 + f) gradient buffer is replaced by what it costs the GPU itself: the rank's own piece
  copied out and W pieces written back into the buffer (the sum over ranks and the xGMI transfers are what is NOT
  measured: 2 x (W - 1) / W x 40 B x N per rank over the links)."""
  import bench
  from taichi_splatting_amd import RasterConfig, frame, render_gaussians, sharded
  from taichi_splatting_amd.distributed import overlap_balanced_bounds
  from taichi_splatting_amd.perspective.projection import project_to_image
  dev = torch.device('cuda', 0)
  cfg = RasterConfig(tile_size=args.tile, pixel_stride=(1, 1) if args.tile == 8 else (2, 2))
  g, cam = bench.make_scene(args, dev)
  W = args.world
  size = cam.image_size
  loss_fn = lambda img, rows: img.sum()

  def timed(fn, steps=None):
    steps = steps or max(args.steps, 20)
    with frame.parked_gc():                 # parked BEFORE the warm-up: its collection idles the GPU for 30-50 ms (bench.py)
      t_ramp = time.perf_counter()
      while time.perf_counter() - t_ramp < 0.1:
        fn()
      for _ in range(max(args.warmup, 10)):
        fn()
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(steps):
        fn()
      torch.cuda.synchronize()
      return (time.perf_counter() - t0) / steps * 1e3

  full = g.clone().requires_grad_(True)
  leaves = (full.position, full.log_scaling, full.rotation, full.alpha_logit, full.feature)

  def single():
    for t in leaves:
      t.grad = None
    render_gaussians(full, cam, cfg, use_sh=True).image.sum().backward()
  t_single = timed(single)
  with torch.no_grad():
    bounds = overlap_balanced_bounds(project_to_image(g, cam, cfg)[0], size, cfg, W)

  per_rank, per_rank_graph, stages = [], [], []
  for r in ([int(x) for x in args.ranks.split(',')] if args.ranks else range(W)):
    def fake_reduce(buf, shard, r=r):
      piece = shard.shape[0]
      shard.copy_(buf[r * piece:(r + 1) * piece])            # my piece of the reduce-scatter
      buf.view(W, piece, -1).copy_(shard.unsqueeze(0).expand(W, -1, -1))     # the all-gather writes W pieces
    st = sharded.StripStep(size, cfg, cam.depth_range, r, W, bounds, reduce=fake_reduce)
    st.probe(full, cam, True)

    def step():
      for t in leaves:
        t.grad = None
      st.step(full, cam, loss_fn, use_sh=True)
    per_rank.append(round(timed(step), 3))
    st.timer = sharded.StageTimer(True)
    for _ in range(5):
      step(); torch.cuda.synchronize(); st.timer.end_step()
    stages.append(st.timer.mean_ms())
    st.timer = sharded.StageTimer(False)
    graph = frame.FrameGraph(step, warmup=1)
    per_rank_graph.append(round(timed(graph.replay), 3))
    assert not st.check()['overlap_overflow']
    del graph
  es = 4
  rows = (args.n + W - 1) // W * W
  out = {"world": W, "n": args.n, "image": list(size), "step": "sharded.StripStep (north_star: replicated gaussians, strips, "
         "reduce-scatter + all-gather)", "single_gpu_ms": round(t_single, 3), "bounds": bounds,
         "per_rank_ms_eager": per_rank, "per_rank_ms_graph": per_rank_graph,
         "max_rank_ms_eager": max(per_rank), "max_rank_ms_graph": max(per_rank_graph),
         "compute_only_speedup_eager": round(t_single / max(per_rank), 2),
         "compute_only_speedup_graph": round(t_single / max(per_rank_graph), 2),
         "stage_ms_rank0": stages[0],
         "collective_bytes_per_rank": {"buffer": rows * 10 * es, "sent_over_xgmi": int(2 * (W - 1) / W * rows * 10 * es)},
         "note": "the cross-rank sum and the xGMI transfers of the reduce-scatter + all-gather are NOT included (device "
                 "copies of the same buffers stand in)"}
  shard_bytes = rows // W * 10 * es
  out["link_model"] = link_model(W, [shard_bytes, shard_bytes], max(per_rank_graph), t_single)
  emit(args, out)


def main_static(args):
  """ShardedStep per rank on one GPU: the forward all-to-all delivers the rows recorded from all W shards, the
  reverse one a buffer of the right size (its values do not matter for the time)."""
  import bench
  from taichi_splatting_amd import RasterConfig, frame, render_gaussians, sharded
  from taichi_splatting_amd.distributed import overlap_balanced_bounds, shard_range
  from taichi_splatting_amd.perspective.projection import project_to_image
  dev = torch.device('cuda', 0)
  cfg = RasterConfig(tile_size=args.tile, pixel_stride=(1, 1) if args.tile == 8 else (2, 2))
  g, cam = bench.make_scene(args, dev)
  W = args.world
  size = cam.image_size
  loss_fn = lambda img, rows: img.sum()

  def timed(fn, steps=None):
    steps = steps or max(args.steps, 20)
    with frame.parked_gc():                 # parked BEFORE the warm-up: its collection idles the GPU for 30-50 ms (bench.py)
      t_ramp = time.perf_counter()
      while time.perf_counter() - t_ramp < 0.1:
        fn()
      for _ in range(max(args.warmup, 10)):
        fn()
      torch.cuda.synchronize()
      t0 = time.perf_counter()
      for _ in range(steps):
        fn()
      torch.cuda.synchronize()
      return (time.perf_counter() - t0) / steps * 1e3

  full = g.clone().requires_grad_(True)
  full_leaves = (full.position, full.log_scaling, full.rotation, full.alpha_logit, full.feature)

  def single():
    for t in full_leaves:
      t.grad = None
    render_gaussians(full, cam, cfg, use_sh=True).image.sum().backward()
  t_single = timed(single)
  with torch.no_grad():
    bounds = overlap_balanced_bounds(project_to_image(g, cam, cfg)[0], size, cfg, W)
  del full

  shards, steps_ = [], []
  recorded = {}
  for r in range(W):
    b, e = shard_range(args.n, W, r)
    shard = g[b:e].clone().contiguous().requires_grad_(True)
    shards.append(shard)
    st = sharded.ShardedStep(size, cfg, cam.depth_range, r, W, bounds, index_offset=b)
    steps_.append(st)
  # capacities: the probe is a collective in a real job; here: bucket = largest over ranks, overlaps per strip
  from taichi_splatting_amd.distributed import exchange_to_strips
  from taichi_splatting_amd.mapper.tile_mapper import map_to_tiles_strip
  biggest, rows_for = 0, {r: [] for r in range(W)}
  with torch.no_grad():
    for r in range(W):
      g2d, depths, idx = project_to_image(shards[r], cam, cfg)
      feats = torch.zeros((g2d.shape[0], 3), device=dev)
      loop = lambda send, sc, rc, group: send.clone()
      _, _, _, _, plan = exchange_to_strips(g2d, feats, depths, size, cfg, bounds, global_index=idx,
                                            index_offset=steps_[r].index_offset, exchange=loop, return_plan=True)
      biggest = max(biggest, max(plan.send_counts))
  cap = (int(biggest * 1.15) + 255) // 256 * 256
  for st in steps_:
    st.bucket_capacity = cap
    st.k_capacity = 1 << 26             # generous for the recording pass; fixed per rank below
  # recording pass: every rank's send buffers (geometry rows and colour rows: the split exchange)
  rec_col = {}
  for r in range(W):
    def rec(recv, send, r=r):
      if send.shape[1] == 9:
        recorded[r] = send.clone()
      elif send.shape[1] == 3:
        rec_col[r] = send.clone()
      recv.copy_(send)
    steps_[r].exchange = rec
    steps_[r].exchange_async = lambda recv, send, rec=rec: (rec(recv, send), (lambda: None))[1]
    steps_[r].split_exchange = True
    with torch.no_grad():
      steps_[r].step(shards[r], cam, loss_fn, use_sh=True, backward=False)

  # ---- modelled link time as MEASURED time: a collective = a spin kernel of the modelled duration (one thread: it takes
  # no compute from the kernels beside it) + the device copy of its buffers, on a stream of its own like RCCL's.  The
  # overlapped figure is then a measurement of what the GPU does while the "links" are busy, not arithmetic.
  link = torch.cuda.Stream(dev)
  a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda._sleep(1_000_000); torch.cuda.synchronize()
  a.record(); torch.cuda._sleep(50_000_000); b.record(); torch.cuda.synchronize()
  cycles_per_ms = 50_000_000 / a.elapsed_time(b)

  def link_ms(nbytes_per_peer):
    return nbytes_per_peer / (LINK_GBS * 1e9) * 1e3 + LAUNCH_US * 1e-3

  per_rank, per_rank_graph, stages = [], [], []
  serial_ms, overlapped_ms, stages_overlapped = [], [], []
  for r in ([int(x) for x in args.ranks.split(',')] if args.ranks else range(W)):
    st, shard = steps_[r], shards[r]
    recv_geo = torch.cat([recorded[s][r * cap:(r + 1) * cap] for s in range(W)]).contiguous()
    recv_col = torch.cat([rec_col[s][r * cap:(r + 1) * cap] for s in range(W)]).contiguous()
    recv_rows = torch.cat([recv_geo[:, :7], recv_col, recv_geo[:, 7:9]], dim=1).contiguous()

    def deliver(recv, send):
      width = send.shape[1]
      recv.copy_(recv_rows if width == 12 else recv_geo if width == 9 else recv_col if width == 3 else send)

    def make_exchanges(charge_links):
      def ex(recv, send):                       # blocking, on the step's stream: nothing overlaps it
        if charge_links:
          torch.cuda._sleep(int(link_ms(cap * send.shape[1] * 4) * cycles_per_ms))
        deliver(recv, send)

      def ex_async(recv, send):                 # on the link stream, collectives in issue order; wait() = stream wait
        main = torch.cuda.current_stream()
        link.wait_stream(main)
        with torch.cuda.stream(link):
          if charge_links:
            torch.cuda._sleep(int(link_ms(cap * send.shape[1] * 4) * cycles_per_ms))
          deliver(recv, send)
          done = torch.cuda.Event()
          done.record(link)
        return lambda: torch.cuda.current_stream().wait_event(done)
      return ex, ex_async

    leaves = (shard.position, shard.log_scaling, shard.rotation, shard.alpha_logit, shard.feature)

    def step():
      for t in leaves:
        t.grad = None
      st.step(shard, cam, loss_fn, use_sh=True)

    def stage_table():
      st.timer = sharded.StageTimer(True)
      for _ in range(5):
        step(); torch.cuda.synchronize(); st.timer.end_step()
      table = st.timer.mean_ms()
      st.timer = sharded.StageTimer(False)
      return table

    # (1) compute only, one forward collective (round 5's line): device copies, no link time
    st.exchange, st.exchange_async = make_exchanges(False)
    st.split_exchange = False
    with torch.no_grad():
      st.step(shard, cam, loss_fn, use_sh=True, backward=False)
    st.k_capacity = frame._round_capacity(int(st.check()['overlaps']) * 1.15)
    per_rank.append(round(timed(step), 3))
    stages.append(stage_table())
    graph = frame.FrameGraph(step, warmup=1)
    per_rank_graph.append(round(timed(graph.replay), 3))
    assert not st.check()['overlap_overflow'] and not st.check()['bucket_overflow']
    del graph
    # (2) links charged, NOTHING overlapped: one forward collective, every collective blocks the step's stream
    st.exchange, st.exchange_async = make_exchanges(True)
    serial_ms.append(round(timed(step), 3))
    # (3) the forward exchange as two collectives (this round), compute only: the split itself must cost nothing
    st.exchange, st.exchange_async = make_exchanges(False)
    st.split_exchange = True
    overlapped_ms.append(round(timed(step), 3))
    stages_overlapped.append(stage_table())
    for t in leaves:
      t.grad = None
  f_ch = 3
  geo_ms, col_ms = link_ms(cap * 9 * 4), link_ms(cap * f_ch * 4)
  fwd_ms, bwd_ms = link_ms(cap * (9 + f_ch) * 4), link_ms(cap * (7 + f_ch) * 4)
  out = {"world": W, "n": args.n, "image": list(size), "step": "sharded.ShardedStep (sync-free, fixed buckets)",
         "single_gpu_ms": round(t_single, 3), "bounds": bounds, "bucket_capacity_rows": cap,
         "per_rank_ms_eager": per_rank, "per_rank_ms_graph": per_rank_graph,
         "max_rank_ms_eager": max(per_rank), "max_rank_ms_graph": max(per_rank_graph),
         "compute_only_speedup_eager": round(t_single / max(per_rank), 2),
         "compute_only_speedup_graph": round(t_single / max(per_rank_graph), 2),
         "stage_ms_rank0": stages[0],
         "note": "compute_only_*: device copies stand in for the all-to-all (no xGMI time).  with_links: every collective ALSO "
                 "runs a spin kernel of the modelled duration on the stream the collective would occupy, so the figures below "
                 "are measured step times of this GPU under the stated link model, not sums"}
  out["exchanged_bytes_per_rank"] = {"forward": W * cap * (9 + f_ch) * 4, "backward": W * cap * (7 + f_ch) * 4,
                                     "off_chip_fraction": round((W - 1) / W, 3)}
  out["link_model"] = link_model(W, [cap * (9 + f_ch) * 4, cap * (7 + f_ch) * 4], max(per_rank_graph), t_single)
  # Overlap: a spin kernel on a SECOND stream does not model a busy link beside compute on this runtime — the streams share
  # hardware queues and the spin kernel holds up the kernels queued behind it (measured: the "overlapped" step came out
  # SLOWER than the serial one, 1.32 vs 1.24 ms, and 1.9 ms with GPU_MAX_HW_QUEUES=8).  So the serial figure is measured
  # (spin kernels in line on the step's own stream, nothing beside them) and what the split exchange hides is arithmetic on
  # measured stage times: the colour collective (col_ms) runs while the strip's unpack + mapper do (stage
  # unpack_map_raster minus its raster forward ~ 0.17 ms on config E at N = 8) and is fully hidden; the geometry collective
  # and the backward exchange stay exposed.
  hidden = fwd_ms - geo_ms
  # ... and what the split itself costs: two collectives, a side stream and an event in front of the raster forward — measured
  # as the difference of the two compute-only steps (median over the ranks: single ranks scatter by +-0.03 ms)
  import statistics
  split_cost = statistics.median(a - b for a, b in zip(overlapped_ms, per_rank))
  with_overlap = [round(t - hidden + max(split_cost, 0.0), 3) for t in serial_ms]
  out["with_links"] = {
    "assumes": f"{LINK_GBS:g} GB/s per link and direction, the {W - 1} links of a rank busy at once, {LAUNCH_US:g} us per collective",
    "collective_ms": {"forward_one_collective": round(fwd_ms, 4), "forward_geometry": round(geo_ms, 4),
                      "forward_colours": round(col_ms, 4), "backward": round(bwd_ms, 4)},
    "per_rank_ms_serial_measured": serial_ms,
    "speedup_serial": round(t_single / max(serial_ms), 2),
    "per_rank_ms_split_exchange_compute_only": overlapped_ms,
    "hidden_link_ms": round(hidden, 4), "split_cost_ms_measured": round(split_cost, 4),
    "per_rank_ms_with_overlap": with_overlap,
    "speedup_with_links": round(t_single / max(with_overlap), 2),
    "overlap_built": "forward exchange as two collectives: the strip's mapper runs while the colour rows travel "
                     "(ms_frame_inputs.colours_ready_event); the backward exchange is NOT overlapped",
    "stage_ms_rank0_split_exchange": stages_overlapped[0],
    # what hiding the backward exchange behind the raster backward band by band could still gain, at most: all of it
    "bound_if_backward_exchange_were_free": round(t_single / (max(with_overlap) - bwd_ms), 2),
    "bound_if_every_collective_were_free": round(t_single / max(per_rank_graph), 2)}
  emit(args, out)


if __name__ == '__main__':
  main()
