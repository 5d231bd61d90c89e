"""Multi-GPU rendering of ONE frame (new design; the reference is single GPU).

One process per GPU (``torch.distributed``, backend "nccl" = RCCL over xGMI).  Two decompositions:

``render_sharded_step`` (default for N > 1) — gaussians AND pixels are sharded.  Rank r owns a
  contiguous shard of the gaussians (its parameters, optimizer state and gradients never leave the
  rank) and a contiguous strip of tile rows.  The per-gaussian stages (projection, SH colour and their
  backward) run on the shard; the projected splats [packed 2D (7) | colour (F) | depth | global id]
  = 48 B for RGB are routed with ONE variable-size all-to-all to the ranks whose strip they can
  overlap (the mapper's own row span, so the routing is exact up to a 0.01 px slack); each rank maps /
  sorts / rasterizes its strip; the 2D-boundary gradients travel back through the reverse all-to-all
  and are summed into the shard.  Nothing is replicated and nothing is all-reduced: per-rank work and
  per-rank traffic are ~1/N of the frame (plus the splats that straddle a strip boundary).

``render_strip_step`` — gaussians replicated, strips of tile rows, ONE all-reduce of the 2D-boundary
  gradient (40 B x V for RGB) before the replicated projection / SH backward.  Simple, but the
  per-gaussian stages and the all-reduce do not shrink with N; kept for scenes that fit one GPU and
  callers that want the full gradient on every rank.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .data_types import Gaussians3D, RasterConfig
from .perspective import CameraParams


def strip_rows(tiles_high: int, world_size: int, rank: int,
               row_weights: Optional[Sequence[float]] = None) -> Tuple[int, int]:
  """Tile-row window [begin, end) of ``rank``.  With ``row_weights`` (e.g. overlaps per tile row,
  replicated on every rank, so no communication is needed to agree) the boundaries equalise the
  cumulative weight; otherwise rows are split evenly.  Strips are contiguous, disjoint and cover
  [0, tiles_high)."""
  assert 0 <= rank < world_size
  if row_weights is None:
    bounds = [(tiles_high * r) // world_size for r in range(world_size + 1)]
  else:
    w = torch.as_tensor(row_weights, dtype=torch.float64)
    assert w.shape[0] == tiles_high
    cum = torch.cumsum(w, 0)
    total = float(cum[-1]) if tiles_high > 0 else 0.0
    bounds = [0]
    for r in range(1, world_size):
      target = total * r / world_size
      b = int(torch.searchsorted(cum, torch.tensor(target, dtype=torch.float64)).item()) + 1 if total > 0 else (tiles_high * r) // world_size
      bounds.append(min(max(b, bounds[-1]), tiles_high))
    bounds.append(tiles_high)
  return bounds[rank], bounds[rank + 1]


def allreduce_boundary_grads(grad_points: torch.Tensor, grad_features: torch.Tensor,
                             group=None) -> Tuple[torch.Tensor, torch.Tensor]:
  """Sum the per-gaussian 2D-boundary gradients [d gaussians2d | d colour] (40 B per visible gaussian for RGB)
  over all strips: reduce-scatter + all-gather of ONE buffer.  On the fully connected xGMI fabric both halves are
  direct exchanges that keep all 7 links of a GPU busy with 1/N-sized pieces, and a rank holds its
  reduced shard between the two calls.  Rows are padded to a multiple of the world size."""
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
    return grad_points, grad_features
  world = dist.get_world_size(group)
  n, width = grad_points.shape[0], grad_points.shape[1] + grad_features.shape[1]
  rows = (n + world - 1) // world * world
  buf = grad_points.new_zeros((rows, width)) if rows != n else grad_points.new_empty((rows, width))
  buf[:n, :grad_points.shape[1]] = grad_points
  buf[:n, grad_points.shape[1]:] = grad_features
  shard = buf.new_empty((rows // world, width))
  dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=group)
  dist.all_gather_into_tensor(buf, shard, group=group)
  return buf[:n, :grad_points.shape[1]], buf[:n, grad_points.shape[1]:]


def backward_through_exchange(loss: torch.Tensor, g2: torch.Tensor, f2: torch.Tensor, ran: Optional[dict] = None):
  """``loss.backward()`` for a rank step whose gradient collective sits behind ``g2`` / ``f2`` in the graph.

  Every rank must run that collective or the others hang, but a ``loss_fn`` may return a constant for an empty
  strip (more ranks than tile rows, zero-weight rows of a balanced split): such a loss has no path to the strip
  image.  ``ran`` is the marker the exchange's backward sets (``ExchangePlan.ran``); if it is still unset after the
  loss's own backward pass, zero gradients are sent through instead.  (The first version made the dependency
  unconditional with ``loss + 0 * g2[:1].sum() + 0 * f2[:1].sum()``: a dozen tiny launches and two full-size
  gradient additions per step, on every rank, for a case that almost never occurs.)"""
  if loss.requires_grad:
    loss.backward()
  if ran is not None and not ran.get('backward'):
    tensors = [t for t in (g2, f2) if t.requires_grad]
    if tensors:
      torch.autograd.backward(tensors, [torch.zeros_like(t) for t in tensors])


def render_strip_step(gaussians: Gaussians3D, camera_params: CameraParams, config: RasterConfig,
                      loss_fn: Callable[[torch.Tensor, Tuple[int, int]], torch.Tensor],
                      use_sh: bool = False, rank: Optional[int] = None, world_size: Optional[int] = None,
                      group=None, backward: bool = True, comm_stats: Optional[dict] = None,
                      bounds: Optional[Sequence[int]] = None):
  """One forward(+backward) step of the strip-sharded renderer on this rank.

  ``bounds`` = [b_0 = 0, ..., b_world = tiles_high]: rank r renders tile rows [b_r, b_{r+1}) (default: even
  split).  ``overlap_balanced_bounds`` derives them from the per-tile-row overlap histogram; the gaussians are
  replicated, so every rank computes the same bounds without communicating.

  ``comm_stats`` (optional dict) receives the bytes this rank contributes to the collective per step.

  ``loss_fn(image, (row_begin_px, row_end_px))`` must return this rank's share of the loss computed
  from its rows of ``image`` (rows outside the strip are zero).  After the call, ``.grad`` of the
  leaf tensors in ``gaussians`` holds the FULL gradient (identical on every rank).
  Returns (Rendering of the strip, loss value of the strip).
  """
  from .perspective.projection import project_to_image
  from .renderer import render_projected
  from .spherical_harmonics import evaluate_sh_at

  if rank is None:
    rank = dist.get_rank(group) if dist.is_initialized() else 0
  if world_size is None:
    world_size = dist.get_world_size(group) if dist.is_initialized() else 1

  # per-gaussian stages: replicated (the camera position is its own small launch: issued before the visible-count
  # synchronisation of the projection, not after it)
  camera_position = camera_params.camera_position if use_sh else None
  gaussians2d, depths, indexes = project_to_image(gaussians, camera_params, config)
  if use_sh:
    features = evaluate_sh_at(gaussians.feature, gaussians.position.detach(), indexes, camera_position, unique_indexes=True)
  else:
    features = gaussians.feature[indexes]

  # cut the autograd graph at the 2D boundary so the strip gradients can be reduced there
  g2 = gaussians2d.detach().requires_grad_(gaussians2d.requires_grad)
  f2 = features.detach().requires_grad_(features.requires_grad)

  ts = config.tile_size
  tiles_high = (camera_params.image_size[1] + ts - 1) // ts
  rows = strip_rows(tiles_high, world_size, rank) if bounds is None else (int(bounds[rank]), int(bounds[rank + 1]))
  rendering = render_projected(indexes, g2, f2, depths.detach(), camera_params, config, tile_rows=rows)

  px_rows = (rows[0] * ts, min(rows[1] * ts, camera_params.image_size[1]))
  loss = loss_fn(rendering.image, px_rows)
  if backward and (g2.requires_grad or f2.requires_grad):
    # participation in the all-reduce below does not depend on what loss_fn returned on this rank: a constant
    # loss simply contributes zeros
    if loss.requires_grad:
      loss.backward()
    gp = g2.grad if g2.grad is not None else torch.zeros_like(g2)
    gf = f2.grad if f2.grad is not None else torch.zeros_like(f2)
    gp, gf = allreduce_boundary_grads(gp, gf, group)
    if comm_stats is not None:
      comm_stats['reduce_scatter_plus_all_gather_buffer_bytes'] = (gp.shape[1] + gf.shape[1]) * gp.shape[0] * gp.element_size()
    tensors, grads = [], []
    if gaussians2d.requires_grad:
      tensors.append(gaussians2d); grads.append(gp)
    if features.requires_grad:
      tensors.append(features); grads.append(gf)
    torch.autograd.backward(tensors, grads)
  return rendering, loss.detach()


# --------------------------------------------------------------------------------------------------
# gaussian-sharded / strip-sharded rendering: all-to-all of projected splats
# --------------------------------------------------------------------------------------------------

def shard_range(n: int, world_size: int, rank: int) -> Tuple[int, int]:
  """Contiguous index range [begin, end) of the gaussians owned by ``rank``."""
  assert 0 <= rank < world_size
  return (n * rank) // world_size, (n * (rank + 1)) // world_size


def strip_bounds(tiles_high: int, world_size: int, row_weights: Optional[Sequence[float]] = None):
  """[b_0 = 0, b_1, ..., b_world = tiles_high]: rank r renders tile rows [b_r, b_{r+1})."""
  return [strip_rows(tiles_high, world_size, r, row_weights)[0] for r in range(world_size)] + [tiles_high]


def splat_row_span(points7: torch.Tensor, image_size: Tuple[int, int], config: RasterConfig,
                   slack_px: float = 0.01) -> Tuple[torch.Tensor, torch.Tensor]:
  """Tile-row span [lo, hi) each packed 2D gaussian can overlap: the bounds of the tile mapper's grid
  query (csrc/splat_math.h obb_grid_query; reference taichi_lib/grid_query.py:10-40) widened by
  ``slack_px`` so that float rounding differences can only add rows.  lo == hi == 0 for splats the
  mapper culls (alpha below the threshold, NaN extent)."""
  ts = float(config.tile_size)
  tiles_high = (image_size[1] + config.tile_size - 1) // config.tile_size
  my, ax, ay, sx, sy, alpha = (points7[:, i] for i in (1, 2, 3, 4, 5, 6))
  gs = torch.sqrt(2.0 * torch.log(alpha / config.alpha_threshold))       # NaN when alpha < threshold
  ey = torch.sqrt((ay * sx * gs) ** 2 + (ax * sy * gs) ** 2) + slack_px
  ok = torch.isfinite(ey) & torch.isfinite(my)
  lo = torch.floor((my - ey) / ts).clamp(0, tiles_high)
  hi = torch.ceil((my + ey) / ts)
  hi = torch.minimum(torch.maximum(hi, lo + 1), torch.full_like(hi, float(tiles_high)))
  lo = torch.where(ok, lo, torch.zeros_like(lo)).to(torch.int64)
  hi = torch.where(ok, hi, torch.zeros_like(hi)).to(torch.int64)
  hi = torch.maximum(hi, lo)          # lo >= tiles_high: empty span
  return lo, hi


def route_to_strips(row_lo: torch.Tensor, row_hi: torch.Tensor, bounds: Sequence[int]):
  """Routing plan, all on device and without a host synchronisation.

  Returns (first_rank, copies, send_counts): splat i goes to the ``copies[i]`` consecutive ranks
  starting at ``first_rank[i]`` (the ranks whose strip intersects [row_lo, row_hi)); ``send_counts[d]``
  = number of splats sent to rank d."""
  world = len(bounds) - 1
  ends = torch.as_tensor(list(bounds[1:]), dtype=torch.int64, device=row_lo.device)
  nonempty = row_hi > row_lo
  first = torch.searchsorted(ends, row_lo, right=True).clamp_(max=world - 1)
  last = torch.searchsorted(ends, (row_hi - 1).clamp_(min=0), right=True).clamp_(max=world - 1)
  copies = torch.where(nonempty, last - first + 1, torch.zeros_like(first))
  # splats per destination = running sum of (+1 at first, -1 after last)
  plus = torch.bincount(first[nonempty], minlength=world + 1)
  minus = torch.bincount(last[nonempty] + 1, minlength=world + 1)
  send_counts = torch.cumsum(plus - minus, 0)[:world]
  return first, copies, send_counts


def expand_routes(first: torch.Tensor, copies: torch.Tensor, total: int):
  """(send_index, dest) of the ``total`` routed copies, ordered by destination rank and, within a
  destination, by local splat index (so that depth ties keep breaking by gaussian index after the
  exchange, as in the single-GPU sort order: mapper/tile_mapper.py)."""
  n = first.shape[0]
  dev = first.device
  if total == 0:
    e = torch.empty(0, dtype=torch.int64, device=dev)
    return e, e
  offsets = torch.cumsum(copies, 0) - copies
  idx = torch.repeat_interleave(torch.arange(n, device=dev), copies, output_size=total)
  dest = first[idx] + (torch.arange(total, device=dev) - offsets[idx])
  dest, order = torch.sort(dest, stable=True)
  return idx[order], dest


# ---- what the collectives of a step look like, for dry runs ---------------------------------------------------------
# `bench.py --dry-run --gpus N` runs the N-rank code path with N processes of a `gloo` group sharing ONE GPU (RCCL
# itself needs one device per rank): every collective below then stages its device tensors through host memory, and —
# when `collective_log` is a list — notes (operation, shapes, dtypes, split lists) exactly as the RCCL call would get
# them, so that the ranks' call sequences can be compared before a real 8-GPU node ever runs them.
collective_log = None


def host_group(group=None) -> bool:
  """True when the process group has no device collectives (gloo): device tensors are staged through the host."""
  return dist.is_available() and dist.is_initialized() and dist.get_backend(group) == 'gloo'


def note_collective(op: str, *tensors, **extra):
  if collective_log is not None:
    collective_log.append((op, [(tuple(t.shape), str(t.dtype).replace('torch.', '')) for t in tensors],
                           {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in extra.items()}))


def all_reduce_any(t: torch.Tensor, op=None, group=None):
  """dist.all_reduce in place, through the host for a gloo group"""
  op = dist.ReduceOp.SUM if op is None else op
  note_collective('all_reduce', t, reduce_op=str(op))
  if t.is_cuda and host_group(group):
    h = t.cpu()
    dist.all_reduce(h, op=op, group=group)
    t.copy_(h)
  else:
    dist.all_reduce(t, op=op, group=group)


def _all_to_all(send: torch.Tensor, send_counts, recv_counts, group) -> torch.Tensor:
  """``recv`` = rows received from every rank, grouped by source rank (RCCL all-to-all: unequal splits)."""
  recv = send.new_empty((int(sum(recv_counts)),) + tuple(send.shape[1:]))
  if dist.is_available() and dist.is_initialized():
    note_collective('all_to_all_single', recv, send, recv_splits=list(recv_counts), send_splits=list(send_counts))
    if send.is_cuda and host_group(group):
      h = torch.empty(recv.shape, dtype=recv.dtype)
      dist.all_to_all_single(h, send.contiguous().cpu(), list(recv_counts), list(send_counts), group=group)
      recv.copy_(h)
      return recv
    dist.all_to_all_single(recv, send.contiguous(), list(recv_counts), list(send_counts), group=group)
  else:
    assert list(send_counts) == list(recv_counts)
    recv.copy_(send)
  return recv


def all_to_all_via_host(send: torch.Tensor, send_counts, recv_counts, group) -> torch.Tensor:
  """The same exchange staged through host memory (for process groups without device collectives,
  e.g. gloo in tests: several ranks can then share one GPU)."""
  return _all_to_all(send.cpu(), send_counts, recv_counts, group).to(send.device)


class _StripExchange(torch.autograd.Function):
  """(gaussians2d, features) -> the rows received for this rank's strip.  The send buffer ``rows`` was
  packed from the same tensors (no grad); backward = reverse all-to-all of [d gaussians2d | d features]
  + sum of the copies of each local splat."""

  @staticmethod
  def forward(ctx, gaussians2d, features, rows, send_index, route, send_counts, recv_counts, group, exchange, ran):
    ctx.route, ctx.ran = route, ran
    ctx.send_index, ctx.send_counts, ctx.recv_counts = send_index, send_counts, recv_counts
    ctx.group, ctx.exchange, ctx.n, ctx.f = group, exchange, gaussians2d.shape[0], features.shape[1]
    recv = exchange(rows, send_counts, recv_counts, group)
    g2, f2, d, ids = _split_rows(recv, ctx.f)
    ctx.mark_non_differentiable(d, ids)
    return g2, f2, d, ids

  @staticmethod
  def backward(ctx, grad_g2, grad_f2, _grad_d, _grad_ids):
    if ctx.ran is not None:
      ctx.ran['backward'] = True
    m = int(sum(ctx.recv_counts))
    dev_like = grad_g2 if grad_g2 is not None else grad_f2
    if grad_g2 is None:
      grad_g2 = dev_like.new_zeros((m, 7))
    if grad_f2 is None:
      grad_f2 = dev_like.new_zeros((m, ctx.f))
    back = ctx.exchange(torch.cat([grad_g2, grad_f2], dim=1), ctx.recv_counts, ctx.send_counts, ctx.group)
    if ctx.route is not None:
      back = back.contiguous()
      from . import _lib
      lib = _lib.load()
      gp, gf = back.new_zeros((ctx.n, 7)), back.new_zeros((ctx.n, ctx.f))
      _lib.check(lib.ms_strip_return_grads(_lib.ptr(back), _lib.ptr(ctx.send_index), _lib.ptr(ctx.route), ctx.f, back.shape[0],
                                           _lib.ptr(gp), _lib.ptr(gf), _lib.current_stream(back.device)),
                 'ms_strip_return_grads')
      return gp, gf, None, None, None, None, None, None, None, None
    grad = back.new_zeros((ctx.n, 7 + ctx.f))
    grad.index_add_(0, ctx.send_index, back)
    return grad[:, :7].contiguous(), grad[:, 7:].contiguous(), None, None, None, None, None, None, None, None


class ExchangePlan:
  """What one forward exchange decided: reusable to send per-splat rows of the strip back to their owners."""

  def __init__(self, send_index, route, send_counts, recv_counts, group, exchange, n, ran=None):
    self.send_index, self.route, self.send_counts, self.recv_counts = send_index, route, send_counts, recv_counts
    self.group, self.exchange, self.n = group, exchange, n
    self.ran = ran if ran is not None else {}      # 'backward': set once the reverse exchange of this step has run


def return_to_owners(plan: ExchangePlan, rows: torch.Tensor) -> torch.Tensor:
  """Send per-received-splat rows (M, k) (e.g. visibility, split heuristics accumulated on the strip) home
  through the reverse all-to-all and sum the copies of each local splat: (V_local, k).  Not differentiable."""
  with torch.no_grad():
    back = plan.exchange(rows.detach().contiguous(), plan.recv_counts, plan.send_counts, plan.group)
    out = back.new_zeros((plan.n, rows.shape[1]))
    out.index_add_(0, plan.send_index, back)
  return out


FORCE_TORCH_ROUTING = False      # tests: compare the HIP routing kernels with the torch formulation


def _use_kernels(t: torch.Tensor) -> bool:
  return t.is_cuda and t.dtype == torch.float32 and not FORCE_TORCH_ROUTING


def _split_rows(recv: torch.Tensor, f: int):
  """(gaussians2d, features, depths, ids) arrays of received rows [packed 2D | colour | depth | id bits]."""
  m = recv.shape[0]
  if _use_kernels(recv):
    from . import _lib
    lib = _lib.load()
    g2, f2 = recv.new_empty((m, 7)), recv.new_empty((m, f))
    d, ids = recv.new_empty((m,)), torch.empty((m,), dtype=torch.int64, device=recv.device)
    _lib.check(lib.ms_strip_unpack(_lib.ptr(recv), m, f, _lib.ptr(g2), _lib.ptr(f2), _lib.ptr(d), _lib.ptr(ids),
                                   _lib.current_stream(recv.device)), 'ms_strip_unpack')
    return g2, f2, d, ids
  id_type = torch.int32 if recv.dtype == torch.float32 else torch.int64
  return (recv[:, :7].contiguous(), recv[:, 7:7 + f].contiguous(), recv[:, 7 + f].contiguous(),
          recv[:, 8 + f].contiguous().view(id_type).to(torch.int64))


def exchange_to_strips(gaussians2d: torch.Tensor, features: torch.Tensor, depths: torch.Tensor,
                       image_size: Tuple[int, int], config: RasterConfig, bounds: Sequence[int],
                       global_index: Optional[torch.Tensor] = None, index_offset: int = 0, group=None,
                       exchange=_all_to_all, return_plan: bool = False):
  """Route this rank's projected splats to the strips they can overlap.

  ``features`` is the (V, F) colour tensor or a callable returning it (evaluated after the routing kernels
  are queued, so that it overlaps the split-size synchronisation).

  Returns (gaussians2d, features, depths, global_index) of the splats RECEIVED for this rank's strip,
  grouped by source rank and in source order; gaussians2d / features stay attached to the autograd
  graph (the backward pass sends their gradients home and sums them per splat).  Global id of local
  splat i = (global_index[i] if given else i) + index_offset.  ``exchange`` is the collective.

  float32 device tensors take the HIP path (csrc/strip_route.hip: count / offsets / pack / unpack
  kernels); other inputs (float64 gradcheck-style tests, CPU tensors under gloo) the equivalent torch
  formulation below."""
  world = len(bounds) - 1
  n = gaussians2d.shape[0]
  g2d, dep = gaussians2d.detach().contiguous(), depths.detach().reshape(-1).contiguous()
  kernels = _use_kernels(g2d) and dep.dtype == torch.float32
  if kernels:
    from . import _lib
    lib = _lib.load()
    stream = _lib.current_stream(g2d.device)
    nb = lib.ms_strip_route_blocks(n)
    route = torch.empty((max(n, 1),), dtype=torch.int32, device=g2d.device)
    block_offsets = torch.empty((world * nb,), dtype=torch.int32, device=g2d.device)
    send_counts_t = torch.empty((world,), dtype=torch.int64, device=g2d.device)
    import ctypes
    bounds_c = (ctypes.c_int32 * (world + 1))(*[int(b) for b in bounds])
    _lib.check(lib.ms_strip_route_count(_lib.ptr(g2d), None, n, int(image_size[1]), config.tile_size,
                                        config.alpha_threshold, ctypes.cast(bounds_c, ctypes.c_void_p), world,
                                        _lib.ptr(route), _lib.ptr(block_offsets), _lib.ptr(send_counts_t), stream),
               'ms_strip_route_count')
  else:
    lo, hi = splat_row_span(g2d, image_size, config)
    first, copies, send_counts_t = route_to_strips(lo, hi, bounds)

  # split sizes: one tiny all-to-all + ONE host synchronisation.  The colours are not needed for the routing:
  # a callable ``features`` is evaluated now, so that its kernels run while the host waits for the counts
  recv_counts_t = exchange(send_counts_t.view(world, 1), [1] * world, [1] * world, group)
  if callable(features):
    features = features()
  f = features.shape[1]
  feats = features.detach().contiguous()
  assert not kernels or feats.dtype == torch.float32, "features must be float32 like the packed gaussians"
  send_counts, recv_counts = torch.stack([send_counts_t, recv_counts_t.view(world)]).tolist()
  total = int(sum(send_counts))

  if kernels:
    rows = g2d.new_empty((total, 9 + f))
    send_index = torch.empty((total,), dtype=torch.int64, device=g2d.device)
    ids = global_index.to(torch.int64).contiguous() if global_index is not None else None
    if total > 0:         # nothing to pack when no local splat reaches any strip (all of them below the alpha gate)
      _lib.check(lib.ms_strip_route_pack(_lib.ptr(g2d), _lib.ptr(feats), _lib.ptr(dep), _lib.ptr(ids), f, n,
                                         world, int(index_offset), _lib.ptr(route), _lib.ptr(block_offsets),
                                         _lib.ptr(send_counts_t), 0, None, _lib.ptr(rows), _lib.ptr(send_index), stream),
                 'ms_strip_route_pack')
  else:
    send_index, _ = expand_routes(first, copies, total)
    gid = (global_index if global_index is not None else torch.arange(n, device=g2d.device)) + index_offset
    id_type = torch.int32 if g2d.dtype == torch.float32 else torch.int64        # ids travel as payload bits
    rows = torch.cat([g2d, feats, dep.to(g2d.dtype).unsqueeze(1), gid.to(id_type).view(g2d.dtype).unsqueeze(1)],
                     dim=1)[send_index]

  ran = {}
  g2, f2, d, gid = _StripExchange.apply(gaussians2d, features, rows, send_index, route if kernels else None,
                                        send_counts, recv_counts, group, exchange, ran)
  d = d.reshape((-1,) + tuple(depths.shape[1:]))
  if return_plan:
    return g2, f2, d, gid, ExchangePlan(send_index, route if kernels else None, send_counts, recv_counts, group, exchange, n, ran)
  return g2, f2, d, gid


def render_sharded_step(shard: Gaussians3D, camera_params: CameraParams, config: RasterConfig,
                        loss_fn: Callable[[torch.Tensor, Tuple[int, int]], torch.Tensor],
                        use_sh: bool = False, rank: Optional[int] = None, world_size: Optional[int] = None,
                        group=None, backward: bool = True, index_offset: int = 0,
                        bounds: Optional[Sequence[int]] = None, exchange=_all_to_all,
                        point_stats: Optional[dict] = None, comm_stats: Optional[dict] = None):
  """One forward(+backward) step with gaussians sharded by index and pixels sharded by tile-row strip.

  ``shard`` holds only this rank's gaussians (``index_offset`` = global index of its first one).
  ``loss_fn(strip_image, (row_begin_px, row_end_px))`` returns this rank's share of the loss;
  ``strip_image`` (row_end_px - row_begin_px, W, F) holds ONLY the strip's pixel rows (no rank ever
  allocates or touches the full frame).  After the call ``.grad`` of the leaf tensors of ``shard`` is the complete gradient of the
  summed loss for those gaussians.  Returns (Rendering of the strip, loss value of the strip); the
  ``points`` of the rendering are the splats received for the strip, ``points.idx`` their global ids.

  ``point_stats``: a dict to fill with what the densification / visibility-aware optimisers need for the
  OWNED gaussians, summed over all strips and sent home with one more reverse all-to-all: ``'visibility'``
  (N_shard,) when ``config.compute_visibility``, ``'point_heuristic'`` (N_shard, 2) when
  ``config.compute_point_heuristic`` (zeros for gaussians that were not visible).
  """
  from .perspective.projection import project_to_image
  from .renderer import render_projected
  from .spherical_harmonics import evaluate_sh_at

  if rank is None:
    rank = dist.get_rank(group) if dist.is_initialized() else 0
  if world_size is None:
    world_size = dist.get_world_size(group) if dist.is_initialized() else 1

  # the camera position is its own small launch: issued before the synchronisations below, not between them
  camera_position = camera_params.camera_position if use_sh else None
  gaussians2d, depths, indexes = project_to_image(shard, camera_params, config)

  def features():
    if use_sh:
      return evaluate_sh_at(shard.feature, shard.position.detach(), indexes, camera_position, unique_indexes=True)
    return shard.feature[indexes]

  ts = config.tile_size
  tiles_high = (camera_params.image_size[1] + ts - 1) // ts
  if bounds is None:
    bounds = strip_bounds(tiles_high, world_size)
  rows = (bounds[rank], bounds[rank + 1])

  g2, f2, d, gid, plan = exchange_to_strips(gaussians2d, features, depths, camera_params.image_size, config, bounds,
                                            global_index=indexes, index_offset=index_offset, group=group,
                                            exchange=exchange, return_plan=True)
  rendering = render_projected(gid, g2, f2, d, camera_params, config, tile_rows=rows, crop_to_rows=True)
  if comm_stats is not None:
    # forward rows [packed 2D | colour | depth | id], backward rows [d packed 2D | d colour]
    f, es = f2.shape[1], g2.element_size()
    sent, received = int(sum(plan.send_counts)), int(sum(plan.recv_counts))
    comm_stats.update(all_to_all_forward_sent_bytes=sent * (9 + f) * es, all_to_all_forward_received_bytes=received * (9 + f) * es,
                      all_to_all_backward_sent_bytes=received * (7 + f) * es if backward else 0,
                      all_to_all_backward_received_bytes=sent * (7 + f) * es if backward else 0)

  h = camera_params.image_size[1]
  px_rows = (min(rows[0] * ts, h), min(rows[1] * ts, h))
  loss = loss_fn(rendering.image, px_rows)
  if backward and (g2.requires_grad or f2.requires_grad):
    # the reverse all-to-all in _StripExchange.backward is a collective: every rank runs it, also one whose
    # loss_fn returned a constant (empty strip)
    backward_through_exchange(loss, g2, f2, plan.ran)

  if point_stats is not None and (config.compute_visibility or config.compute_point_heuristic):
    # after the backward pass: the split heuristics are accumulated there
    cols = []
    if config.compute_visibility:
      cols.append(rendering.points.visibility.reshape(-1, 1))
    if config.compute_point_heuristic:
      cols.append(torch.stack([rendering.points.prune_cost, rendering.points.split_score], dim=1))
    home = return_to_owners(plan, torch.cat(cols, dim=1).to(g2.dtype))
    full = home.new_zeros((shard.position.shape[0], home.shape[1]))
    full[indexes] = home
    k = 0
    if config.compute_visibility:
      point_stats['visibility'] = full[:, 0]
      k = 1
    if config.compute_point_heuristic:
      point_stats['point_heuristic'] = full[:, k:k + 2]
  return rendering, loss.detach()


def overlap_balanced_bounds(gaussians2d: torch.Tensor, image_size: Tuple[int, int], config: RasterConfig,
                            world_size: int, group=None, all_reduce: bool = False):
  """Strip boundaries that equalise the number of (tile, gaussian) OVERLAPS per strip — the work of the mapper
  and both raster passes — instead of the number of tile rows: per-tile-row histogram of each splat's tile span
  (rows x columns of the mapper's bounding-box query; the oriented-box test removes a near-constant fraction).
  Replicated gaussians need no communication (``all_reduce=False``: every rank computes the same histogram); for
  sharded gaussians the histograms are summed with one small all-reduce.  One host read of ``tiles_high`` floats:
  meant to be called every few frames, not inside the timed step."""
  ts = float(config.tile_size)
  tiles_high = (image_size[1] + config.tile_size - 1) // config.tile_size
  tiles_wide = (image_size[0] + config.tile_size - 1) // config.tile_size
  p = gaussians2d.detach()
  mx, my, ax, ay, sx, sy, alpha = (p[:, i] for i in range(7))
  gs = torch.sqrt(2.0 * torch.log(alpha / config.alpha_threshold))
  ex = torch.sqrt((ax * sx * gs) ** 2 + (ay * sy * gs) ** 2)
  ey = torch.sqrt((ay * sx * gs) ** 2 + (ax * sy * gs) ** 2)
  ok = torch.isfinite(ex) & torch.isfinite(ey) & torch.isfinite(mx) & torch.isfinite(my)
  cols = (torch.ceil((mx + ex) / ts).clamp(0, tiles_wide) - torch.floor((mx - ex) / ts).clamp(0, tiles_wide)).clamp(min=1)
  lo = torch.floor((my - ey) / ts).clamp(0, tiles_high - 1)
  hi = torch.ceil((my + ey) / ts).clamp(1, tiles_high)
  hi = torch.maximum(hi, lo + 1)
  # rows the mapper culls (alpha below the threshold, NaN / inf extents) weigh nothing AND must not index: their
  # lo / hi are NaN, which converts to INT64_MIN
  weight = torch.where(ok, cols, torch.zeros_like(cols))
  lo = torch.where(ok, lo, torch.zeros_like(lo))
  hi = torch.where(ok, hi, torch.ones_like(hi))
  # difference array: +cols at lo, -cols at hi, then a prefix sum gives overlaps per tile row
  diff = torch.zeros((tiles_high + 1,), dtype=torch.float32, device=p.device)
  diff.index_add_(0, lo.to(torch.int64), weight.float())
  diff.index_add_(0, hi.to(torch.int64), -weight.float())
  hist = torch.cumsum(diff, 0)[:tiles_high]
  if all_reduce and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
    all_reduce_any(hist, group=group)
  return strip_bounds(tiles_high, world_size, hist.clamp(min=0).tolist())


def balanced_strip_bounds(gaussians2d: torch.Tensor, image_size: Tuple[int, int], config: RasterConfig,
                          world_size: int, group=None):
  """Strip boundaries that equalise the number of splat centres per strip over ALL ranks' splats
  (one small all-reduce of a per-tile-row histogram + one host sync)."""
  ts = config.tile_size
  tiles_high = (image_size[1] + ts - 1) // ts
  row = torch.floor(gaussians2d[:, 1].detach() / ts).clamp(0, tiles_high - 1).to(torch.int64)
  hist = torch.bincount(row, minlength=tiles_high).to(torch.float32)
  if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
    all_reduce_any(hist, group=group)
  return strip_bounds(tiles_high, world_size, hist.tolist())
