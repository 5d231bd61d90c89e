"""Sync-free multi-GPU rank steps on the frame executor (new design; the reference renders on one GPU).

One process per GPU.  Both decompositions of ``distributed.py`` as FIXED launch sequences without a host round trip,
so that a rank enqueues a whole step ahead of its GPU and the step can be captured in a HIP graph:

``StripStep`` (BASELINE.json north_star): gaussians replicated, every rank renders a strip of tile rows; the
  2D-boundary gradients [d gaussians2d (7) | d colour (F)] are summed with ONE reduce-scatter + ONE all-gather
  between the raster backward and the per-gaussian backward pass.

``ShardedStep``: gaussians sharded by index AND pixels by strip.  A rank projects its shard, routes the splats into
  FIXED-capacity per-destination buckets (``ms_strip_route_pack`` with ``bucket_capacity``: the all-to-all has equal
  splits the host knows without reading counts back; unused bucket rows are all-zero = splats with alpha 0, which
  overlap no tile and get no gradient), renders its strip from the received rows with the executor's
  ``projected_input`` mode, and sends the 2D-boundary gradients home through the reverse all-to-all.

What the step of ``distributed.render_sharded_step`` had to wait for — the visible count, the all-to-all split sizes,
the overlap total — stays on the device here: nothing is compacted, bucket and overlap-list capacities are fixed by
``probe()`` (a synchronising dry run outside the hot loop, like the strip bounds) and overflow only raises device-side
flags.  The flags (and each strip's overlap total) are written through PINNED host words: ``step()`` compares them on
entry without synchronising and raises ``frame.FrameOverflow`` one step after an overflow; ``MS_STRICT=1``
synchronises inside the step and raises for that step; ``check()`` reads the device counters (synchronises).
"""
from __future__ import annotations

import ctypes
from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import _lib, frame
from .data_types import Gaussians3D, RasterConfig
from .perspective import CameraParams


def _exchange_all_to_all(recv: torch.Tensor, send: torch.Tensor, group):
  """equal-split all-to-all (RCCL: point-to-point sends over the xGMI links); a copy on one rank"""
  if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
    from . import distributed
    distributed.note_collective('all_to_all_single', recv, send)
    if send.is_cuda and distributed.host_group(group):          # gloo (dry runs, tests): through host memory
      h = torch.empty(recv.shape, dtype=recv.dtype)
      dist.all_to_all_single(h, send.cpu(), group=group)
      recv.copy_(h)
    else:
      dist.all_to_all_single(recv, send, group=group)
  else:
    recv.copy_(send)


def _exchange_all_to_all_async(recv: torch.Tensor, send: torch.Tensor, group):
  """The same all-to-all started WITHOUT making the current stream wait for it: returns ``wait()``, which does.  RCCL
  runs the collective on the process group's own stream, collectives of one group in issue order: two exchanges issued
  back to back travel one after the other, and whoever waits for the first is not held up by the second."""
  if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
    from . import distributed
    if not (send.is_cuda and distributed.host_group(group)):
      distributed.note_collective('all_to_all_single', recv, send)
      work = dist.all_to_all_single(recv, send, group=group, async_op=True)
      return work.wait
  _exchange_all_to_all(recv, send, group)
  return lambda: None


class StageTimer:
  """HIP events at the stage boundaries of a rank step (recorded on the step's stream, read after a synchronise):
  makes an N > 1 bench line interpretable — compute, exchange and host time can be told apart."""

  def __init__(self, enabled: bool):
    self.enabled = enabled
    self.marks = []
    self.totals = {}
    self.steps = 0

  def mark(self, name: str):
    if self.enabled:
      e = torch.cuda.Event(enable_timing=True)
      e.record()
      self.marks.append((name, e))

  def end_step(self):
    """call after a synchronise: folds the marks of the finished step into per-stage totals"""
    if not self.enabled or len(self.marks) < 2:
      self.marks = []
      return
    for (_, a), (name, b) in zip(self.marks[:-1], self.marks[1:]):
      self.totals[name] = self.totals.get(name, 0.0) + a.elapsed_time(b)
    self.marks = []
    self.steps += 1

  def mean_ms(self):
    return {k: round(v / max(self.steps, 1), 4) for k, v in self.totals.items()}


def _desc(n, image_size, dtype, f, sh_degree, config, depth_range, tile_rows=None, projected=False, capacity=0,
          depth16=False):
  w, h = int(image_size[0]), int(image_size[1])
  ts = config.tile_size
  tiles_high = (h + ts - 1) // ts
  rows = (0, tiles_high) if tile_rows is None else (max(0, int(tile_rows[0])), min(tiles_high, int(tile_rows[1])))
  return _lib.FrameDescC(n=int(n), k_capacity=int(capacity), image_w=w, image_h=h, dtype=_lib.dtype_code(dtype), f=int(f),
                         sh_degree=int(sh_degree), depth16=int(depth16), tile_row_begin=rows[0], tile_row_end=rows[1],
                         projected_input=int(projected), mapper=0, near_plane=float(depth_range[0]),
                         far_plane=float(depth_range[1]), blur_cov=float(config.blur_cov),
                         clamp_margin=float(config.clamp_margin), raster=_lib.raster_config_c(config)), rows


def _layout(desc):
  lay = _lib.FrameLayoutC()
  _lib.check(_lib.load().ms_frame_layout_query(ctypes.byref(desc), ctypes.byref(lay)), "frame layout")
  return lay


def _block(nbytes, device):
  return torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=device)


def _feature_shape(feature: torch.Tensor, use_sh: bool):
  if use_sh:
    assert feature.ndim == 3, f"SH features must have 3 dimensions, got {feature.shape}"
    d = feature.shape[2]
    degree = int(round(d ** 0.5)) - 1
    assert (degree + 1) ** 2 == d and 0 <= degree <= 3, f"SH feature count must be 1, 4, 9 or 16, got {d}"
    return feature.shape[1], degree
  assert feature.ndim == 2, f"Features must be (N, C) if use_sh=False, got {feature.shape}"
  return feature.shape[1], -1


def _accumulate(leaf: torch.Tensor, grad: Optional[torch.Tensor]):
  if grad is None or not leaf.requires_grad:
    return
  if leaf.grad is None:
    leaf.grad = grad
  else:
    leaf.grad += grad


class _RankStep:
  """what both decompositions share: geometry, capacities, flags, the strip frame + loss"""

  def __init__(self, image_size, config: RasterConfig, depth_range, rank: int, world: int, bounds: Sequence[int],
               group=None, time_stages: bool = False):
    self.image_size = (int(image_size[0]), int(image_size[1]))
    self.config, self.depth_range = config, (float(depth_range[0]), float(depth_range[1]))
    self.rank, self.world, self.group = int(rank), int(world), group
    ts = config.tile_size
    self.tiles_high = (self.image_size[1] + ts - 1) // ts
    self.bounds = [int(b) for b in bounds]
    assert len(self.bounds) == world + 1 and self.bounds[0] == 0 and self.bounds[-1] == self.tiles_high, \
      f"bounds must run from 0 to tiles_high = {self.tiles_high} over {world} ranks, got {self.bounds}"
    self.rows = (self.bounds[rank], self.bounds[rank + 1])
    h = self.image_size[1]
    self.px_rows = (min(self.rows[0] * ts, h), min(self.rows[1] * ts, h))
    self.k_capacity = 0
    self.timer = StageTimer(time_stages)
    # Overflow indicators live in PINNED HOST memory the kernels write through (like the eager frame's K word): the
    # host compares them without touching the device, so an overflowed step is reported at the next step() — never a
    # silent run of background-only strips.  flags[0] = bucket overflow (sticky: only ever set), k_word = overlap total
    # of the last strip frame.
    self.flags = torch.zeros((2,), dtype=torch.int32).pin_memory()
    self.k_word = torch.zeros((1,), dtype=torch.int32).pin_memory()
    self._flags_np, self._k_np = self.flags.numpy(), self.k_word.numpy()
    # the strip's mapper reports the longest tile run it sorted with a single workgroup; a strip that shows a giant one
    # maps with the pre-sort sequence from the next step on (frame.LONG_RUN_LIMIT, same rule as render_gaussians)
    self.run_word = torch.zeros((1,), dtype=torch.int32).pin_memory()
    self._run_np = self.run_word.numpy()
    self.mapper = _lib.MAPPER_DIRECT
    self.bucket_overflowed = False   # host-side and sticky: poll() clears the device-written word, check() still reports it
    self.strict = frame.STRICT
    self.last_counters = None    # counters view of the last strip frame
    self.comm_bytes = {}

  def _strip_forward(self, desc, inputs, device, dtype, f):
    """mapper + raster forward of this rank's strip: returns (keep_n, keep_k, image (strip rows only), alpha)"""
    lib = _lib.load()
    stream = _lib.current_stream(device)
    if int(self._run_np[0]) > frame.LONG_RUN_LIMIT or desc.depth16:     # 16-bit keys: pre-sort, as frame.py does
      self.mapper = _lib.MAPPER_PRESORT
    desc.mapper = self.mapper
    inputs.longest_run_host = self.run_word.data_ptr()
    lay = _layout(desc)
    keep_n, scratch_n = _block(lay.keep_n_bytes, device), _block(lay.scratch_n_bytes, device)
    keep_k, scratch_k = _block(lay.keep_k_bytes, device), _block(lay.scratch_k_bytes, device)
    _lib.check(lib.ms_frame_project_count(ctypes.byref(desc), ctypes.byref(inputs), keep_n.data_ptr(), scratch_n.data_ptr(),
                                          self.k_word.data_ptr(), None, stream), "rank step (map)")
    w = self.image_size[0]
    y0, y1 = self.px_rows
    image = torch.empty((y1 - y0, w, f), dtype=dtype, device=device)
    alpha = torch.empty((y1 - y0, w), dtype=dtype, device=device)
    es = image.element_size()
    if y1 > y0:
      image_ptr, alpha_ptr = image.data_ptr() - y0 * w * f * es, alpha.data_ptr() - y0 * w * es
    else:
      # an empty strip (bounds may repeat a value): the mapper still runs (capacity, counters), the raster touches no
      # row; zero-row tensors have a null data pointer, which the C entry points reject
      self._dummy = torch.empty((16,), dtype=dtype, device=device)
      image_ptr = alpha_ptr = self._dummy.data_ptr()
    _lib.check(lib.ms_frame_map_raster(ctypes.byref(desc), ctypes.byref(inputs), keep_n.data_ptr(), scratch_n.data_ptr(),
                                       keep_k.data_ptr(), scratch_k.data_ptr(), image_ptr, alpha_ptr, None, stream),
               "rank step (raster)")
    self.last_counters = keep_n[lay.counters:lay.counters + 32].view(torch.int32)
    return lay, keep_n, keep_k, image, alpha

  def _loss_and_image_grad(self, image, loss_fn, backward):
    image.requires_grad_(backward)
    loss = loss_fn(image, self.px_rows)
    if not backward:
      return loss.detach(), None
    g_image = None
    if loss.requires_grad:
      (g_image,) = torch.autograd.grad(loss, image, allow_unused=True)
    if g_image is None:              # a loss that does not depend on this strip (empty strip): zeros join the collective
      g_image = torch.zeros_like(image)
    return loss.detach(), g_image          # possibly an expanded scalar (sum / mean loss): see _image_grad_pointer

  def _image_grad_pointer(self, gr, g_image, moments_path, row_bytes_f):
    """dL/dimage for the strip's raster backward.  A sum / mean loss hands autograd an EXPANDED scalar (strides 0): the
    moments kernel then reads one pixel's values (``grad_image_broadcast``, as render_gaussians does) instead of an
    (rows, W, f) copy that would be written here and read back there.  Returns the tensor that must stay alive."""
    broadcast = (frame.BROADCAST_GRAD and moments_path and g_image.dim() == 3 and g_image.shape[0] * g_image.shape[1] > 1
                 and g_image.stride(0) == 0 and g_image.stride(1) == 0)
    if broadcast:
      keep = g_image[0, 0].contiguous()
      gr.grad_image, gr.grad_image_broadcast = keep.data_ptr(), 1
    else:
      keep = g_image.contiguous()
      gr.grad_image, gr.grad_image_broadcast = keep.data_ptr() - row_bytes_f, 0
    return keep

  def check(self) -> dict:
    """Host read (synchronises the device) of the overflow indicators: the counters of the LAST step, and whether ANY
    step since construction / ``reset_overflow()`` dropped splats in the exchange (``bucket_overflow`` is sticky on the
    host: ``poll()`` clears the word the kernels write so that it can raise once per event, not the record)."""
    torch.cuda.synchronize()       # the pinned words are written by kernels of the step: all of them have landed now
    k, live, over = (self.last_counters[:3].tolist() if self.last_counters is not None else (0, 0, 0))
    self.bucket_overflowed = self.bucket_overflowed or bool(int(self._flags_np[0]))
    out = {"overlaps": k, "overlap_capacity": self.k_capacity, "overlap_overflow": bool(over),
           "bucket_overflow": self.bucket_overflowed}
    return out

  def reset_overflow(self):
    """after a new probe(): forget earlier bucket overflows"""
    self.bucket_overflowed = False
    self._flags_np[0] = 0

  def poll(self):
    """Raise ``frame.FrameOverflow`` if a finished step exceeded a capacity (no synchronisation: pinned words).
    ``step()`` calls it on entry, so a loop of sync-free / graph-replayed steps stops at the step after the overflow;
    with ``MS_STRICT=1`` (``self.strict``) the step synchronises and raises for itself."""
    k = int(self._k_np[0])
    if self._flags_np[0]:
      self.bucket_overflowed = True
      self._flags_np[0] = 0
      raise frame.FrameOverflow(f"rank {self.rank}: a destination bucket of the splat exchange overflowed (capacity "
                                f"{getattr(self, 'bucket_capacity', 0)} rows): splats were dropped from a strip.  probe() again "
                                "(larger slack) before the next step.")
    if self.k_capacity > 0 and (k < 0 or k > self.k_capacity):
      self._k_np[0] = 0
      raise frame.FrameOverflow(f"rank {self.rank}: the strip produced {k} tile overlaps, its buffers hold {self.k_capacity}: "
                                "that step rendered the background only and returned zero gradients.  probe() again "
                                "(larger slack) before the next step.")

  def _enter_step(self, gaussians, camera_params):
    self.poll()
    for t in (camera_params.T_camera_world, camera_params.projection):
      if t.requires_grad:
        raise NotImplementedError("rank steps do not produce camera gradients (T_camera_world / projection require grad): "
                                  "the per-gaussian pass of a rank sees only its shard / strip; use render_gaussians on one GPU "
                                  "for pose optimisation")

  def _leave_step(self):
    if self.strict and not torch.cuda.is_current_stream_capturing():
      torch.cuda.synchronize()
      self.poll()

  def _raster_backward_mode(self, lib, desc, gr, g_image, device, rows_n):
    """moments path + deterministic commits as frame.py does it (``rasterizer.function.DETERMINISTIC_BACKWARD`` /
    MS_DETERMINISTIC is honoured on the rank steps too)"""
    from .rasterizer import function as raster_function
    det = bool(raster_function.DETERMINISTIC_BACKWARD)
    moments_path = bool(lib.ms_frame_uses_moments(ctypes.byref(desc), int(det)))
    if moments_path:
      gr.moments = frame._moments_buffer(device, rows_n, det).data_ptr()
      gr.boundary_form = _lib.BOUNDARY_COVARIANCE      # the strips hand over dL/d(2D covariance): see include/mi355_splat.h
      gr.deterministic = int(det)
      if det:
        self._fixed_exp = _lib.fixed_point_exponents(g_image)
        gr.fixed_exp = self._fixed_exp.data_ptr()
    return moments_path, det


class StripStep(_RankStep):
  """north_star partition: replicated gaussians, tile-row strips, reduce-scatter + all-gather of the 2D-boundary
  gradients.  ``step(gaussians, camera, loss_fn)``: ``loss_fn(strip_image, (y0, y1))`` gets ONLY the strip's pixel
  rows; afterwards ``.grad`` of the gaussians' leaf tensors holds the full gradient (identical on every rank)."""

  def __init__(self, *args, reduce=None, **kw):
    super().__init__(*args, **kw)
    # reduce(buf (rows, 7 + f), shard (rows / world, 7 + f)): sum over the ranks, result back in every rank's buf.
    # Default: ONE reduce-scatter + ONE all-gather (RCCL).  tools/emulate_sharded.py substitutes device copies of the
    # same buffers to time a rank's own work on one GPU.
    self.reduce = reduce or self._reduce_scatter_all_gather

  def _reduce_scatter_all_gather(self, buf, shard):
    from . import distributed
    distributed.note_collective('reduce_scatter_tensor', shard, buf)
    distributed.note_collective('all_gather_into_tensor', buf, shard)
    if buf.is_cuda and distributed.host_group(self.group):         # gloo (dry runs, tests): through host memory
      world = dist.get_world_size(self.group)
      mine = torch.empty(shard.shape, dtype=shard.dtype)
      dist.reduce_scatter(mine, list(buf.cpu().chunk(world)), op=dist.ReduceOp.SUM, group=self.group)
      parts = [torch.empty(shard.shape, dtype=shard.dtype) for _ in range(world)]
      dist.all_gather(parts, mine, group=self.group)
      buf.copy_(torch.cat(parts))
      shard.copy_(mine)
      return
    dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=self.group)
    dist.all_gather_into_tensor(buf, shard, group=self.group)

  def probe(self, gaussians: Gaussians3D, camera_params: CameraParams, use_sh: bool, slack: float = 1.15):
    """one synchronising dry run: fixes the overlap-list capacity of this rank's strip"""
    self.reset_overflow()          # new capacities: earlier overflows no longer describe this step
    from .mapper.tile_mapper import map_to_tiles_strip
    from .perspective.projection import project_to_image
    with torch.no_grad():
      g2d, depths, _ = project_to_image(gaussians, camera_params, self.config)
      o2p, _ = map_to_tiles_strip(g2d, depths, self.image_size, self.config, tile_rows=self.rows,
                                  ndc_range=self.depth_range)
    self.k_capacity = frame._round_capacity(o2p.shape[0] * slack)
    return self.k_capacity

  def step(self, gaussians: Gaussians3D, camera_params: CameraParams, loss_fn: Callable, use_sh: bool = True,
           backward: bool = True):
    assert self.k_capacity > 0, "StripStep.probe() first (fixes the overlap-list capacity)"
    self._enter_step(gaussians, camera_params)
    lib = _lib.load()
    tensors = [t.detach().contiguous() for t in (*gaussians.shape_tensors(), gaussians.feature,
                                                 camera_params.T_camera_world.reshape(4, 4), camera_params.projection.reshape(4))]
    pos, lsc, rot, alog, feat, Tcw, proj = tensors
    device, dtype, n = pos.device, pos.dtype, pos.shape[0]
    f, degree = _feature_shape(feat, use_sh)
    stream = _lib.current_stream(device)
    timer = self.timer
    timer.mark('start')
    desc, _ = _desc(n, self.image_size, dtype, f, degree, self.config, self.depth_range, tile_rows=self.rows,
                    capacity=self.k_capacity)
    inputs = _lib.FrameInputsC(position=pos.data_ptr(), log_scaling=lsc.data_ptr(), rotation=rot.data_ptr(),
                               alpha_logit=alog.data_ptr(), feature=feat.data_ptr(), T_camera_world=Tcw.data_ptr(),
                               projection=proj.data_ptr(), points7=None, depth=None, colours=None)
    lay, keep_n, keep_k, image, alpha = self._strip_forward(desc, inputs, device, dtype, f)
    timer.mark('project_sh_map_raster')
    loss, g_image = self._loss_and_image_grad(image, loss_fn, backward)
    timer.mark('loss')
    if not backward:
      self._leave_step()
      return image.detach(), loss

    # raster backward of the strip -> (n, 7 + f) 2D-boundary gradients -> sum over the strips
    gr = _lib.FrameGradsC()
    moments_path, det = self._raster_backward_mode(lib, desc, gr, g_image, device, n)
    width = 7 + f
    rows = (n + self.world - 1) // self.world * self.world
    es = image.element_size()
    # moments path, N > 1: the finalize pass stores its rows straight into the collective's (rows, 7 + f) buffer and the
    # per-gaussian pass reads them back with a row stride (no packing / unpacking copies: 4 x 240 MB at 6 M gaussians)
    in_place = moments_path and self.world > 1
    buf = None
    if in_place:
      buf = torch.empty((rows, width), dtype=dtype, device=device)
      if rows != n:
        buf[n:].zero_()
      gp = gc = None
    else:
      gp = torch.empty((n, 7), dtype=dtype, device=device) if moments_path else torch.zeros((n, 7), dtype=dtype, device=device)
      gc = torch.empty((n, f), dtype=dtype, device=device) if moments_path else torch.zeros((n, f), dtype=dtype, device=device)
    y0 = self.px_rows[0]
    row_bytes = y0 * self.image_size[0] * es
    gr.image = image.data_ptr() - row_bytes * f
    g_keep = self._image_grad_pointer(gr, g_image, moments_path, row_bytes * f)      # noqa: F841 (alive until the launch)
    gr.stage = _lib.BACKWARD_RASTER
    if in_place:
      gr.grad_points7, gr.grad_colours, gr.boundary_stride = buf.data_ptr(), buf.data_ptr() + 7 * es, width
    else:
      gr.grad_points7, gr.grad_colours = gp.data_ptr(), gc.data_ptr()
    _lib.check(lib.ms_frame_backward(ctypes.byref(desc), ctypes.byref(inputs), keep_n.data_ptr(), keep_k.data_ptr(),
                                     ctypes.byref(gr), stream), "strip step (raster backward)")
    timer.mark('raster_bwd')
    if self.world > 1:
      if not in_place:
        buf = torch.zeros((rows, width), dtype=dtype, device=device) if rows != n else torch.empty((rows, width), dtype=dtype, device=device)
        buf[:n, :7] = gp
        buf[:n, 7:] = gc
      shard = torch.empty((rows // self.world, width), dtype=dtype, device=device)
      self.reduce(buf, shard)
      if not in_place:
        gp, gc = buf[:n, :7].contiguous(), buf[:n, 7:].contiguous()
      self.comm_bytes = {"reduce_scatter_plus_all_gather_buffer_bytes": rows * width * buf.element_size()}
    timer.mark('reduce_scatter_all_gather')

    need = [t.requires_grad for t in (*gaussians.shape_tensors(), gaussians.feature)]
    grads = [torch.empty_like(t) if need[i] else None for i, t in enumerate((pos, lsc, rot, alog))]
    g2 = _lib.FrameGradsC()
    g2.stage = _lib.BACKWARD_GAUSSIANS
    g2.boundary_form = gr.boundary_form
    if in_place:
      g2.grad_points7, g2.grad_colours, g2.boundary_stride = buf.data_ptr(), buf.data_ptr() + 7 * es, width
    else:
      g2.grad_points7, g2.grad_colours = gp.data_ptr(), gc.data_ptr()
    g2.grad_position, g2.grad_log_scaling, g2.grad_rotation, g2.grad_alpha_logit = (_lib.ptr(t) for t in grads)
    grad_feature = None
    if need[4]:
      if degree >= 0:
        grad_feature = torch.empty_like(feat)
        g2.grad_feature = grad_feature.data_ptr()
      else:
        grad_feature = buf[:n, 7:].contiguous() if in_place else gc
    _lib.check(lib.ms_frame_backward(ctypes.byref(desc), ctypes.byref(inputs), keep_n.data_ptr(), keep_k.data_ptr(),
                                     ctypes.byref(g2), stream), "strip step (gaussian backward)")
    timer.mark('gaussian_bwd')
    for leaf, g in zip((*gaussians.shape_tensors(), gaussians.feature), (*grads, grad_feature)):
      _accumulate(leaf, g)
    self._leave_step()
    return image.detach(), loss


class ShardedStep(_RankStep):
  """gaussians sharded by index, pixels by tile-row strip; fixed-capacity all-to-all both ways.

  ``step(shard, camera, loss_fn)``: ``shard`` holds this rank's gaussians (``index_offset`` = global index of its
  first one); afterwards ``.grad`` of its leaf tensors is the complete gradient of the summed loss."""

  def __init__(self, *args, index_offset: int = 0, exchange=None, exchange_async=None, split_exchange: Optional[bool] = None, **kw):
    super().__init__(*args, **kw)
    self.index_offset = int(index_offset)
    self.bucket_capacity = 0
    self.exchange = exchange or (lambda recv, send: _exchange_all_to_all(recv, send, self.group))
    # Forward exchange as TWO collectives (round 6, opt-in: split_exchange=True / MS_SPLIT_EXCHANGE=1): geometry rows
    # [packed 2D | depth | id] (36 bytes) first, colour rows (4 f bytes) behind them.  The strip's mapper reads geometry only,
    # so it runs while the colours are on the links; the raster forward waits for them through an event
    # (ms_frame_inputs.colours_ready_event).  exchange_async(recv, send) starts a collective and returns wait(), which
    # makes the CURRENT stream wait for it.  Off by default: at config E, N = 8 it hides 0.028 ms of modelled link time and
    # its own side stream + event cost 0.03 ms in the emulation (profiles/r06_emul_sharded_8_4096.json) — it pays on slower
    # links or more colour channels than three.
    import os
    self.split_exchange = (os.environ.get('MS_SPLIT_EXCHANGE', '0') not in ('', '0')) if split_exchange is None else bool(split_exchange)
    if exchange is not None and exchange_async is None:
      # a caller that substitutes the blocking exchange (emulation, tests) gets it for both collectives
      def exchange_async(recv, send, _ex=exchange):
        _ex(recv, send)
        return lambda: None
    self.exchange_async = exchange_async or (lambda recv, send: _exchange_all_to_all_async(recv, send, self.group))
    self._side = {}

  def probe(self, shard: Gaussians3D, camera_params: CameraParams, use_sh: bool, slack: float = 1.15, exchange=None):
    """one synchronising dry run (a collective: every rank calls it): the largest per-destination bucket over all
    ranks fixes the bucket capacity, this rank's strip fixes its overlap-list capacity.  ``exchange``: the
    variable-size all-to-all of ``distributed.exchange_to_strips`` (default: RCCL)"""
    self.reset_overflow()          # new capacities: earlier overflows no longer describe this step
    from .distributed import exchange_to_strips, _all_to_all
    from .mapper.tile_mapper import map_to_tiles_strip
    from .perspective.projection import project_to_image
    with torch.no_grad():
      g2d, depths, idx = project_to_image(shard, camera_params, self.config)
      feats = torch.zeros((g2d.shape[0], 3), dtype=g2d.dtype, device=g2d.device)
      g2, f2, d, gid, plan = exchange_to_strips(g2d, feats, depths, self.image_size, self.config, self.bounds,
                                                global_index=idx, index_offset=self.index_offset, group=self.group,
                                                exchange=exchange or _all_to_all, return_plan=True)
      biggest = torch.tensor([max(plan.send_counts) if plan.send_counts else 0], dtype=torch.int64)
      if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
        from . import distributed
        if not distributed.host_group(self.group):
          biggest = biggest.to(g2d.device)
        distributed.all_reduce_any(biggest, op=dist.ReduceOp.MAX, group=self.group)
      o2p, _ = map_to_tiles_strip(g2, d, self.image_size, self.config, tile_rows=self.rows, ndc_range=self.depth_range)
    self.bucket_capacity = (int(int(biggest.item()) * slack) + 255) // 256 * 256
    self.k_capacity = frame._round_capacity(o2p.shape[0] * slack)
    return self.bucket_capacity, self.k_capacity

  def step(self, shard: Gaussians3D, camera_params: CameraParams, loss_fn: Callable, use_sh: bool = True,
           backward: bool = True):
    assert self.k_capacity > 0 and self.bucket_capacity > 0, "ShardedStep.probe() first (fixes the capacities)"
    self._enter_step(shard, camera_params)
    lib = _lib.load()
    tensors = [t.detach().contiguous() for t in (*shard.shape_tensors(), shard.feature,
                                                 camera_params.T_camera_world.reshape(4, 4), camera_params.projection.reshape(4))]
    pos, lsc, rot, alog, feat, Tcw, proj = tensors
    device, dtype, n = pos.device, pos.dtype, pos.shape[0]
    assert dtype == torch.float32, "ShardedStep: float32 (the routing kernels of csrc/strip_route.hip)"
    f, degree = _feature_shape(feat, use_sh)
    stream = _lib.current_stream(device)
    world, cap = self.world, self.bucket_capacity
    timer = self.timer
    timer.mark('start')

    # ---- per-gaussian stage on the shard (no compaction: culled gaussians carry depth 0 and are not routed) --------
    desc_a, _ = _desc(n, self.image_size, dtype, f, degree, self.config, self.depth_range)
    lay_a = _layout(desc_a)
    keep_a = _block(lay_a.keep_n_bytes, device)
    in_a = _lib.FrameInputsC(position=pos.data_ptr(), log_scaling=lsc.data_ptr(), rotation=rot.data_ptr(),
                             alpha_logit=alog.data_ptr(), feature=feat.data_ptr(), T_camera_world=Tcw.data_ptr(),
                             projection=proj.data_ptr(), points7=None, depth=None, colours=None)
    _lib.check(lib.ms_frame_project(ctypes.byref(desc_a), ctypes.byref(in_a), keep_a.data_ptr(), stream), "sharded step (project)")
    points7 = frame._view(keep_a, lay_a.points7, dtype, (n, 7))
    depth = frame._view(keep_a, lay_a.depth, dtype, (n,))
    colours = frame._view(keep_a, lay_a.colours, dtype, (n, f)) if degree >= 0 else feat
    timer.mark('project_sh')

    # ---- route into fixed buckets, exchange ------------------------------------------------------------------------
    nb = lib.ms_strip_route_blocks(n)
    route = torch.empty((max(n, 1),), dtype=torch.int32, device=device)
    block_offsets = torch.empty((world * nb,), dtype=torch.int32, device=device)
    send_counts = torch.empty((world,), dtype=torch.int64, device=device)
    bounds_c = (ctypes.c_int32 * (world + 1))(*self.bounds)
    _lib.check(lib.ms_strip_route_count(points7.data_ptr(), depth.data_ptr(), n, self.image_size[1], self.config.tile_size,
                                        self.config.alpha_threshold, ctypes.cast(bounds_c, ctypes.c_void_p), world,
                                        route.data_ptr(), block_offsets.data_ptr(), send_counts.data_ptr(), stream),
               "sharded step (route)")
    m = world * cap
    split = self.split_exchange and f >= 1
    width = 9 if split else 9 + f
    send = torch.zeros((m, width), dtype=dtype, device=device)
    send_col = torch.zeros((m, f), dtype=dtype, device=device) if split else None
    send_index = torch.full((m,), -1, dtype=torch.int64, device=device)
    # slots[i, c] = row of the send buffer that carries copy c of gaussian i: where its gradient comes back
    slots = torch.empty((max(n, 1), world), dtype=torch.int32, device=device)
    if n > 0 and split:
      _lib.check(lib.ms_strip_route_pack_split(points7.data_ptr(), colours.data_ptr(), depth.data_ptr(), None, f, n, world,
                                               self.index_offset, route.data_ptr(), block_offsets.data_ptr(),
                                               send_counts.data_ptr(), cap, self.flags.data_ptr(), send.data_ptr(),
                                               send_col.data_ptr(), send_index.data_ptr(), slots.data_ptr(), stream),
                 "sharded step (pack)")
    elif n > 0:
      _lib.check(lib.ms_strip_route_pack_slots(points7.data_ptr(), colours.data_ptr(), depth.data_ptr(), None, f, n, world,
                                               self.index_offset, route.data_ptr(), block_offsets.data_ptr(),
                                               send_counts.data_ptr(), cap, self.flags.data_ptr(), send.data_ptr(),
                                               send_index.data_ptr(), slots.data_ptr(), stream), "sharded step (pack)")
    timer.mark('route_pack')
    recv = torch.empty_like(send)
    g2 = torch.empty((m, 7), dtype=dtype, device=device)
    f2 = torch.empty((m, f), dtype=dtype, device=device)
    d2 = torch.empty((m,), dtype=dtype, device=device)
    ids = torch.empty((m,), dtype=torch.int64, device=device)
    colours_ready = None
    if split:
      wait_geometry = self.exchange_async(recv, send)
      wait_colours = self.exchange_async(f2, send_col)          # the received colour rows ARE the strip's colour array
      wait_geometry()
      timer.mark('exchange_forward')
      # a side stream waits for the colours and records the event the raster forward waits for (fork / join: capturable)
      main = torch.cuda.current_stream(device)
      side = self._side.get(device.index)
      if side is None:
        side = self._side[device.index] = torch.cuda.Stream(device)
      side.wait_stream(main)
      with torch.cuda.stream(side):
        wait_colours()
        colours_ready = torch.cuda.Event()
        colours_ready.record(side)
      _lib.check(lib.ms_strip_unpack(recv.data_ptr(), m, 0, g2.data_ptr(), None, d2.data_ptr(), ids.data_ptr(), stream),
                 "sharded step (unpack)")
    else:
      self.exchange(recv, send)
      timer.mark('exchange_forward')
      _lib.check(lib.ms_strip_unpack(recv.data_ptr(), m, f, g2.data_ptr(), f2.data_ptr(), d2.data_ptr(), ids.data_ptr(), stream),
                 "sharded step (unpack)")

    # ---- this rank's strip from the received rows ------------------------------------------------------------------
    desc_b, _ = _desc(m, self.image_size, dtype, f, -1, self.config, self.depth_range, tile_rows=self.rows,
                      projected=True, capacity=self.k_capacity)
    in_b = _lib.FrameInputsC(points7=g2.data_ptr(), depth=d2.data_ptr(), colours=f2.data_ptr())
    if colours_ready is not None:
      in_b.colours_ready_event = int(colours_ready.cuda_event)
    lay_b, keep_b, keep_k, image, alpha = self._strip_forward(desc_b, in_b, device, dtype, f)
    in_b.colours_ready_event = None          # (the backward calls read the same struct: the colours have long arrived)
    timer.mark('unpack_map_raster')
    loss, g_image = self._loss_and_image_grad(image, loss_fn, backward)
    timer.mark('loss')
    es = image.element_size()
    self.comm_bytes = {"all_to_all_forward_bytes": m * (9 + f) * es, "all_to_all_backward_bytes": m * (7 + f) * es if backward else 0,
                       "forward_collectives": 2 if split else 1,
                       "bucket_capacity_rows": cap, "off_chip_fraction": (world - 1) / world}
    if not backward:
      self._leave_step()
      return image.detach(), loss

    # ---- backward: strip raster -> gradients of the received rows -> home -> per-gaussian pass ---------------------
    gr = _lib.FrameGradsC()
    moments_path, det = self._raster_backward_mode(lib, desc_b, gr, g_image, device, m)
    row_bytes = self.px_rows[0] * self.image_size[0] * es
    gr.image = image.data_ptr() - row_bytes * f
    g_keep = self._image_grad_pointer(gr, g_image, moments_path, row_bytes * f)      # noqa: F841 (alive until the launch)
    gr.stage = _lib.BACKWARD_RASTER
    bw = 7 + f
    if moments_path:
      # the finalize pass stores [d packed 2D | d colour] straight into the rows of the return buffer
      back_send = torch.empty((m, bw), dtype=dtype, device=device)
      gr.grad_points7, gr.grad_colours = back_send.data_ptr(), back_send.data_ptr() + 7 * es
      gr.boundary_stride = bw
    else:
      gp = torch.zeros((m, 7), dtype=dtype, device=device)
      gc = torch.zeros((m, f), dtype=dtype, device=device)
      gr.grad_points7, gr.grad_colours = gp.data_ptr(), gc.data_ptr()
    _lib.check(lib.ms_frame_backward(ctypes.byref(desc_b), ctypes.byref(in_b), keep_b.data_ptr(), keep_k.data_ptr(),
                                     ctypes.byref(gr), stream), "sharded step (raster backward)")
    if not moments_path:
      back_send = torch.cat([gp, gc], dim=1)
    timer.mark('raster_bwd')
    back = torch.empty_like(back_send)
    self.exchange(back, back_send)
    timer.mark('exchange_backward')

    need = [t.requires_grad for t in (*shard.shape_tensors(), shard.feature)]
    grads = [torch.empty_like(t) if need[i] else None for i, t in enumerate((pos, lsc, rot, alog))]
    ga = _lib.FrameGradsC()
    ga.stage = _lib.BACKWARD_GAUSSIANS
    ga.boundary_form = gr.boundary_form
    home = None
    if degree >= 0 or not need[4]:
      # the per-gaussian pass reads every splat's returned rows straight from the receive buffer (summed in copy
      # order): no return pass, no home array, no fill
      ga.gather_world, ga.gather_rows = world, back.data_ptr()
      ga.gather_slots, ga.gather_route = slots.data_ptr(), route.data_ptr()
    else:
      # plain colours with a feature gradient wanted: the caller needs d(colour) as an array — rows summed at home
      home = torch.zeros((n, bw), dtype=dtype, device=device)
      if n > 0:
        _lib.check(lib.ms_strip_return_rows(back.data_ptr(), send_index.data_ptr(), route.data_ptr(), f, m, home.data_ptr(),
                                            stream), "sharded step (return)")
      ga.grad_points7, ga.grad_colours = home.data_ptr(), home.data_ptr() + 7 * es
    ga.boundary_stride = bw
    ga.grad_position, ga.grad_log_scaling, ga.grad_rotation, ga.grad_alpha_logit = (_lib.ptr(t) for t in grads)
    grad_feature = None
    if need[4]:
      if degree >= 0:
        grad_feature = torch.empty_like(feat)
        ga.grad_feature = grad_feature.data_ptr()
      else:
        grad_feature = home[:, 7:].contiguous()
    _lib.check(lib.ms_frame_backward(ctypes.byref(desc_a), ctypes.byref(in_a), keep_a.data_ptr(), None,
                                     ctypes.byref(ga), stream), "sharded step (gaussian backward)")
    timer.mark('return_gaussian_bwd')
    for leaf, g in zip((*shard.shape_tensors(), shard.feature), (*grads, grad_feature)):
      _accumulate(leaf, g)
    self._leave_step()
    return image.detach(), loss
