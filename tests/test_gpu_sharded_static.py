"""-m gpu: the sync-free rank steps of taichi_splatting_amd/sharded.py (frame executor, fixed-capacity buckets, no
host round trip) with REAL processes sharing the box's GPU: process group = gloo, collectives staged through host
memory.  Every rank's strip must equal the rows of the single-process frame bit for bit (same kernels, same per-tile
order) and the gradients of its gaussians the full-frame gradients."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _scene(world, n=20000, size=(320, 208), margin=0.4):
  from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
  torch.manual_seed(world)
  cam = random_camera(image_size=size)
  g = random_3d_gaussians(n, cam, scale_factor=1.5, alpha_range=(0.1, 0.9), margin=margin)     # some gaussians culled
  g = g.replace(feature=(torch.rand(n, 3, 16) - 0.5) * 0.5)
  torch.manual_seed(0)
  G = torch.randn(size[1], size[0], 3, device=DEV)
  return g, cam.to(device=DEV), G


def _grads(g):
  return [g.position.grad, g.log_scaling.grad, g.rotation.grad, g.alpha_logit.grad, g.feature.grad]


def _close(got, want, tol=2e-3):
  worst = 0.0
  for a, b in zip(got, want):
    scale = max(1.0, float(b.abs().max()))
    worst = max(worst, float((a - b).abs().max()) / scale)
  return worst < tol, worst


def _host_all_to_all(recv, send):
  r, s = torch.empty(recv.shape, dtype=recv.dtype), send.cpu()
  dist.all_to_all_single(r, s)
  recv.copy_(r)


def _worker(rank, world, port, mode, balance, ret):
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    from taichi_splatting_amd import RasterConfig, render_gaussians, sharded
    from taichi_splatting_amd.distributed import shard_range, strip_bounds
    size = (320, 208)
    g, cam, G = _scene(world, size=size)
    n = g.position.shape[0]
    cfg = RasterConfig()
    tiles_high = (size[1] + 15) // 16
    bounds = strip_bounds(tiles_high, world, [1.0 + (r % 3) for r in range(tiles_high)] if balance else None)
    loss_fn = lambda img, px: (img * G[px[0]:px[1]]).sum()

    full = g.to(DEV).requires_grad_(True)
    r = render_gaussians(full, cam, cfg, use_sh=True)
    (r.image * G).sum().backward()
    y0, y1 = bounds[rank] * 16, min(bounds[rank + 1] * 16, size[1])

    if mode == 'sharded':
      b, e = shard_range(n, world, rank)
      mine = g[b:e].to(DEV).requires_grad_(True)
      step = sharded.ShardedStep(size, cfg, cam.depth_range, rank, world, bounds, index_offset=b, exchange=_host_all_to_all)
      from taichi_splatting_amd.distributed import all_to_all_via_host
      caps = step.probe(mine, cam, True, exchange=all_to_all_via_host)
      want = [t[b:e] for t in _grads(full)]
    else:
      mine = g.to(DEV).requires_grad_(True)
      step = sharded.StripStep(size, cfg, cam.depth_range, rank, world, bounds)
      caps = step.probe(mine, cam, True)
      want = _grads(full)
      # gloo has no device reduce-scatter: stage the two collectives of the strip step through the host
      rs, ag = dist.reduce_scatter_tensor, dist.all_gather_into_tensor

      def rs_host(out, inp, op=None, group=None):
        chunks = list(inp.cpu().chunk(world))
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.reduce_scatter(o, chunks, op=dist.ReduceOp.SUM)
        out.copy_(o)

      def ag_host(out, inp, group=None):
        parts = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(world)]
        dist.all_gather(parts, inp.cpu())
        out.copy_(torch.cat(parts))
      dist.reduce_scatter_tensor, dist.all_gather_into_tensor = rs_host, ag_host

    ok = True
    for _ in range(2):                     # twice: the persistent accumulators must come back clean
      for t in (mine.position, mine.log_scaling, mine.rotation, mine.alpha_logit, mine.feature):
        t.grad = None
      image, loss = step.step(mine, cam, loss_fn, use_sh=True)
      ok = ok and image.shape[0] == y1 - y0 and torch.equal(image, r.image[y0:y1].detach())
      good, worst = _close(_grads(mine), want)
      ok = ok and good
    st = step.check()
    ok = ok and not st['overlap_overflow'] and not st.get('bucket_overflow', False)
    ret[rank] = (bool(ok), worst, caps, st)
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('mode,world,balance', [('sharded', 2, False), ('sharded', 3, True), ('sharded', 8, False),
                                                ('strips', 2, False), ('strips', 3, True)])
def test_static_rank_steps_match_full_frame(mode, world, balance):
  port = _free_port()
  mgr = mp.Manager()
  ret = mgr.dict()
  mp.spawn(_worker, args=(world, port, mode, balance, ret), nprocs=world, join=True)
  assert len(ret) == world and all(v[0] for v in dict(ret).values()), dict(ret)


def test_world_of_one_equals_render_gaussians_and_flags_bucket_overflow():
  from taichi_splatting_amd import RasterConfig, render_gaussians, sharded
  size = (256, 160)
  g, cam, G = _scene(1, n=15000, size=size)
  cfg = RasterConfig()
  loss_fn = lambda img, px: (img * G[px[0]:px[1]]).sum()
  full = g.to(DEV).requires_grad_(True)
  r = render_gaussians(full, cam, cfg, use_sh=True)
  (r.image * G).sum().backward()
  for cls in (sharded.ShardedStep, sharded.StripStep):
    mine = g.to(DEV).requires_grad_(True)
    step = cls(size, cfg, cam.depth_range, 0, 1, [0, 10])
    step.probe(mine, cam, True)
    image, _ = step.step(mine, cam, loss_fn, use_sh=True)
    assert torch.equal(image, r.image.detach())
    good, worst = _close(_grads(mine), _grads(full))
    assert good, (cls.__name__, worst)
  # a bucket that is too small drops splats and says so
  mine = g.to(DEV).requires_grad_(True)
  step = sharded.ShardedStep(size, cfg, cam.depth_range, 0, 1, [0, 10])
  step.probe(mine, cam, True)
  step.bucket_capacity = 1024
  step.step(mine, cam, loss_fn, use_sh=True)
  assert step.check()['bucket_overflow']
  # and an overlap list that is too small renders the background only and says so
  step = sharded.StripStep(size, cfg, cam.depth_range, 0, 1, [0, 10])
  step.probe(mine, cam, True)
  step.k_capacity = 4096
  image, _ = step.step(mine, cam, loss_fn, use_sh=True, backward=False)
  st = step.check()
  assert st['overlap_overflow'] and st['overlaps'] > 4096 and float(image.abs().max()) == 0.0


def test_world_of_one_with_plain_colours():
  """use_sh=False: d(colour) IS the feature gradient, so ShardedStep sums the returned rows into an interleaved home
  array (ms_strip_return_rows) instead of letting the per-gaussian pass gather them"""
  from taichi_splatting_amd import RasterConfig, render_gaussians, sharded
  size = (256, 160)
  g, cam, G = _scene(2, n=12000, size=size)
  g = g.replace(feature=torch.rand(g.position.shape[0], 3))
  cfg = RasterConfig()
  loss_fn = lambda img, px: (img * G[px[0]:px[1]]).sum()
  full = g.to(DEV).requires_grad_(True)
  r = render_gaussians(full, cam, cfg, use_sh=False)
  (r.image * G).sum().backward()
  for cls in (sharded.ShardedStep, sharded.StripStep):
    mine = g.to(DEV).requires_grad_(True)
    step = cls(size, cfg, cam.depth_range, 0, 1, [0, 10])
    step.probe(mine, cam, False)
    image, _ = step.step(mine, cam, loss_fn, use_sh=False)
    assert torch.equal(image, r.image.detach())
    good, worst = _close(_grads(mine), _grads(full))
    assert good, (cls.__name__, worst)


def test_sharded_step_in_a_hip_graph():
  from taichi_splatting_amd import RasterConfig, frame, sharded
  size = (256, 160)
  g, cam, G = _scene(1, n=15000, size=size)
  cfg = RasterConfig()
  loss_fn = lambda img, px: (img * G[px[0]:px[1]]).sum()
  mine = g.to(DEV).requires_grad_(True)
  leaves = (mine.position, mine.log_scaling, mine.rotation, mine.alpha_logit, mine.feature)
  step = sharded.ShardedStep(size, cfg, cam.depth_range, 0, 1, [0, 10])
  step.probe(mine, cam, True)

  def one():
    for t in leaves:
      t.grad = None
    return step.step(mine, cam, loss_fn, use_sh=True)
  image, _ = one()
  want_image, want = image.clone(), [t.grad.clone() for t in leaves]
  graph = frame.FrameGraph(one, warmup=1)
  for _ in range(2):
    image, _ = graph.replay()
  torch.cuda.synchronize()
  assert torch.equal(image, want_image)
  good, worst = _close([t.grad for t in leaves], want, tol=1e-4)
  assert good, worst


@pytest.mark.parametrize('use_sh', [True, False])
def test_split_forward_exchange_equals_the_single_collective(use_sh):
  """ShardedStep(split_exchange=True) (round 6, opt-in): the forward exchange as two collectives — geometry rows, then
  colour rows that land directly in the strip's colour array while its mapper already runs; the raster forward waits for
  them through ms_frame_inputs.colours_ready_event.  Same rows, same kernels: image bit for bit, gradients to the order
  of the float atomics; and the step still captures into a HIP graph (fork / join through the side stream)."""
  from taichi_splatting_amd import RasterConfig, frame, sharded
  size = (256, 160)
  g, cam, G = _scene(3, n=15000, size=size)
  if not use_sh:
    g = g.replace(feature=torch.rand(g.position.shape[0], 3))
  cfg = RasterConfig()
  loss_fn = lambda img, px: (img * G[px[0]:px[1]]).sum()
  results = {}
  for split in (False, True):
    mine = g.to(DEV).requires_grad_(True)
    step = sharded.ShardedStep(size, cfg, cam.depth_range, 0, 1, [0, 10], split_exchange=split)
    step.probe(mine, cam, use_sh)
    image, _ = step.step(mine, cam, loss_fn, use_sh=use_sh)
    torch.cuda.synchronize()
    assert step.comm_bytes['forward_collectives'] == (2 if split else 1)
    assert not step.check()['bucket_overflow'] and not step.check()['overlap_overflow']
    results[split] = (image.clone(), _grads(mine))
  assert torch.equal(results[True][0], results[False][0]) and float(results[False][0].max()) > 0.05
  good, worst = _close(results[True][1], results[False][1])
  assert good, worst
  # captured
  mine = g.to(DEV).requires_grad_(True)
  leaves = (mine.position, mine.log_scaling, mine.rotation, mine.alpha_logit, mine.feature)
  step = sharded.ShardedStep(size, cfg, cam.depth_range, 0, 1, [0, 10], split_exchange=True)
  step.probe(mine, cam, use_sh)

  def one():
    for t in leaves:
      t.grad = None
    return step.step(mine, cam, loss_fn, use_sh=use_sh)
  graph = frame.FrameGraph(one, warmup=1)
  for _ in range(2):
    image, _ = graph.replay()
  torch.cuda.synchronize()
  assert torch.equal(image, results[False][0])
  good, worst = _close([t.grad for t in leaves], results[False][1], tol=1e-4)
  assert good, worst
