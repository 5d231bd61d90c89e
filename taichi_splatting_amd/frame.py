"""A whole frame as ONE autograd node on a fixed launch sequence (csrc/frame.hip, include/mi355_splat.h
"frame executor").

``render_gaussians`` (reference ``renderer.py:23-108``) runs through here: projection, SH colour, the tile mapper
(overlap count / scan / emission in storage order, stable sort on the tile bits, ranges, per-tile depth sort — or, per
scene shape, the depth pre-sort sequence: ``_choose_mapper``) and the raster forward are enqueued by two C calls, the
backward pass by one (raster backward + ONE pass over the gaussians).  Differences to the reference's orchestration, none of
them visible in results:

* nothing is compacted and the visible count V is never read back: culled gaussians keep their row (depth 0, no
  overlaps, zero gradients).  The compacted ``(V, ...)`` arrays of ``Rendering.points`` are made when a caller
  touches them (``LazyPoints``) — that access is the only place a host synchronisation on V remains;
* the overlap total K stays on the device.  The overlap buffers have a capacity remembered per scene shape; in eager
  mode the host looks at K (pinned word + event) only AFTER the whole forward is enqueued, so the GPU never waits for
  it, and re-runs the emission with larger buffers in the rare case the capacity was exceeded.  Under HIP-graph
  capture (``FrameGraph`` / ``torch.cuda.graph``) nothing is read back: the capacity is fixed at capture time and an
  overflow shows in ``frame_status``.
"""
from __future__ import annotations

import collections
import contextlib
import ctypes
import gc
from dataclasses import dataclass, replace
import os
import threading
import time
from typing import Optional, Tuple
import weakref

import torch

from . import _lib
from .data_types import RasterConfig

# MS_FRAME=legacy: render_gaussians composes the modular operators (project_to_image, evaluate_sh_at, map_to_tiles,
# rasterize_with_tiles) as in rounds 1-2 — kept for A/B measurements and as the cross-check of the frame path
USE_FRAME = os.environ.get('MS_FRAME', 'frame') != 'legacy'

K_SLACK = 1.25            # capacity = K_SLACK x the largest overlap total seen for this scene shape
K_GRANULE = 1 << 16

_k_capacity = {}          # scene-shape key -> overlap-list capacity
_mapper_mode = {}         # scene-shape key -> _lib.MAPPER_DIRECT / MAPPER_PRESORT (see _choose_mapper)
# overlaps per gaussian above which the depth pre-sort sequence is the faster mapper, with some hysteresis.  Same box,
# direct - presort per frame: K/N 2.13 (config D) -0.16 ms, 2.25 (1 M, tile 32) -0.06, 2.59 (3 M) -0.08, 3.34 (1.5 M)
# -0.04, 3.65 (config D at tile 8) +0.06 (tools/diag/ab_mapper.sh)
PRESORT_ABOVE, DIRECT_BELOW = 3.6, 3.4
# The direct sequence sorts a tile run beyond 5120 entries with ONE workgroup (~12 ns per entry: 0.2 ms at 16 384).  The
# kernel writes the length of a run above that size into a pinned word (ms_frame_inputs.longest_run_host); a scene
# shape that shows one maps with the pre-sort from the next frame on, whatever its overlaps per gaussian.
LONG_RUN_LIMIT = 16384
_run_words = {}           # scene-shape key -> (pinned int32[1] tensor, numpy view)
_presort_sticky = set()   # scene shapes that showed a run above LONG_RUN_LIMIT
_k_host = {}              # device index -> KSlots: a ring of pinned int32 words, ONE PER FRAME IN FLIGHT
_moments = collections.OrderedDict()   # (device index, stream, n, deterministic) -> accumulator rows, zero between frames
_moments_pinned = set()   # keys whose buffer address is baked into a captured HIP graph: never evicted
_identity = {}            # (device index, n) -> arange(n) int64
_lock = threading.RLock() # guards the caches above and serialises the enqueue of a backward pass (ctypes drops the GIL)

MOMENTS_LRU = 4           # accumulator buffers kept per device (scene sizes / streams alternating in one process)
K_SLOTS = 64              # pinned K words per device; a frame holds one from its forward until it has looked at K
GRAPH_K_WORDS = 1024      # pinned K words per device for frames captured into HIP graphs
SPLIT_LONG_RUNS = os.environ.get('MS_SPLIT_LONG_RUNS', '1') not in ('', '0')   # A/B switch of the long-run segments
# run-time parameters of the segments (ms_frame_desc.split_long_runs / split_seg_len): a run is cut when it is longer than
# SPLIT_MIN_RUN entries (0: the library's default, 16 384) into segments of >= SPLIT_SEG_LEN entries (0: default);
# SPLIT_ALWAYS: every frame carries the segment launches, not only shapes that showed a long run (tests reach the segment
# kernels on scenes the oracle finishes in seconds this way: set_split_policy)
SPLIT_MIN_RUN = int(os.environ.get('MS_SPLIT_MIN_RUN', '0'))
SPLIT_SEG_LEN = int(os.environ.get('MS_SPLIT_SEG_LEN', '0'))
SPLIT_ALWAYS = os.environ.get('MS_SPLIT_ALWAYS', '0') not in ('', '0')
BROADCAST_GRAD = os.environ.get('MS_BROADCAST_GRAD', '1') not in ('', '0')   # A/B switch of grad_image_broadcast
STRICT = os.environ.get('MS_STRICT', '0') not in ('', '0')   # graph replays synchronise and raise on overflow

# Eager frames that will be differentiated CAN look at their overlap total LATE (round 6; opt-in: MS_LAZY_SETTLE=1 or
# frame.LAZY_SETTLE = True): the K word stays in its pinned slot and is read at the NEXT frame's entry (or at the first host
# access to the frame's lists / status), so the backward of frame i is enqueued while frame i's forward still runs and the
# host can be a whole frame ahead of the GPU — what a HIP-graph replay gets for free.  Only for scene shapes whose capacity
# has not grown for LAZY_AFTER settled frames; an overflow found late means that frame rendered the background and returned
# zero gradients: FrameOverflow is raised at the point it is found (as a graph replay does).
# OFF by default: on config D the frame time is the same either way (3.216 against 3.219 ms on one box — the host's wait
# was never what separates eager from a graph replay), while a trainer whose overlap total jumps
# by more than K_SLACK between two frames of one shape (a new camera) would get an exception instead of round 5's
# transparent re-run.  Worth switching on for loops with a slow host and steady overlap totals.
LAZY_SETTLE = os.environ.get('MS_LAZY_SETTLE', '0') not in ('', '0')
# Visibility of a frame that WILL be differentiated with point heuristics on, taken from its backward pass (round 6; OPT-IN:
# MS_VISIBILITY_FROM_BACKWARD=1 or frame.VISIBILITY_FROM_BACKWARD = True).  The raster backward visits every (pixel, splat)
# pair again with a lane per splat, where the sum of a splat's blend weights is one more addition per pixel step and a
# twelfth column of the heuristics' moment row; the forward, a lane per pixel, pays a transposing wave reduction per four
# hits for it (+0.175 ms on config D).  Such a frame runs its forward WITHOUT visibility and the per-gaussian backward pass
# writes it.  NOT the reference's number exactly, hence off by default: the reference's forward keeps adding the weights of
# pairs behind a pixel's saturation point (forward.py:127-128 has no early exit), its backward — and therefore this sum —
# drops them (backward.py:154); a pixel contributes at most 1 - saturate_threshold = 1e-4 in total to all the splats behind
# that point (config D-like scenes: 3 % of the gaussians differ by more than 1e-4, the largest difference 1.5e-3; a
# gaussian wholly behind saturated pixels reads 0 instead of ~1e-5 and counts as invisible).  oracle/raster.py
# active_visibility() is the quantity, tests/test_gpu_round6.py holds it.  Reading the visibility BEFORE backward()
# (``rendering.points``, ``point_outputs``) runs the forward's visibility kernel on demand (``visibility_passes``) and gives
# the reference's number.
VISIBILITY_FROM_BACKWARD = os.environ.get('MS_VISIBILITY_FROM_BACKWARD', '0') not in ('', '0')
# The SH colours of a CAPTURED frame on a second stream, beside the mapper (round 6): the colours are not needed before the
# raster forward, the SH pass streams 1.2 GB at the HBM rate while the mapper's dozen launches (scan, radix passes, ranges,
# per-tile sort) are short and partly latency bound — side by side they fill each other's gaps.  The frame forks behind
# ms_frame_project_count (the SH kernel reads the depths for its culling) and joins through
# ms_frame_inputs.colours_ready_event in front of the raster forward; the graph holds both as edges.  Measured on config
# D, one box, two runs each: a replayed step 3.012 / 3.026 ms with it against 3.050 / 3.076 without — and an EAGER frame
# 3.115 / 3.119 against 3.074 / 3.079: between two queues the same two dependencies are barrier packets the host submits,
# dearer than what the overlap returns.  Hence under capture only.  MS_SH_SIDE_STREAM=0 switches it off (A/B).
SH_SIDE_STREAM = os.environ.get('MS_SH_SIDE_STREAM', '1') not in ('', '0')
_side_streams = {}        # device index -> the executor's second stream
visibility_passes = 0     # deferred frames whose visibility was read before their backward pass had written it
LAZY_AFTER = 3
_stable_frames = {}       # scene-shape key -> settled frames in a row that fitted the remembered capacity
_unsettled = collections.deque()   # FrameStates whose overlap total the host has not looked at yet (oldest first)

host_syncs = 0            # looks at an overlap total BETWEEN a frame's forward and its backward / return to the caller
entry_waits = 0           # lazily settled frames whose total was not there yet at the next frame's entry (back-pressure:
                          # the GPU then has that frame's forward rest and whole backward still queued — it does not idle)
settles = 0               # looks at an overlap total, of either kind
point_syncs = 0           # LazyPoints materialisations (host read of the visible count)


@dataclass(frozen=True)
class FrameOptions:
  image_size: Tuple[int, int]
  depth_range: Tuple[float, float]
  config: RasterConfig
  use_sh: bool
  use_depth16: bool = False
  tile_rows: Optional[Tuple[int, int]] = None
  crop_to_rows: bool = False
  render_median_depth: bool = False


def frame_supported(feature: torch.Tensor, config: RasterConfig, use_sh: bool) -> bool:
  """The executor instantiates 1..4 colour channels (csrc/raster.hip); wider feature vectors take the modular
  operators, which chunk the channels."""
  f = feature.shape[1]
  return 1 <= f <= 4 and feature.is_cuda


def set_overlap_capacity(n: int, image_size, config: RasterConfig, capacity: int, device=None, tile_rows=None,
                         use_depth16: bool = False):
  """Fix the overlap-list capacity for a scene shape (needed before capturing a frame in a HIP graph when no eager
  frame of that shape has run yet)."""
  dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
  key = _shape_key(dev, n, image_size, config, tile_rows, use_depth16)
  _k_capacity[key] = _round_capacity(capacity)
  _choose_mapper(key, int(capacity / K_SLACK), n)
  if not torch.cuda.is_current_stream_capturing():
    _k_ring(dev)               # pinned words must exist before a capture starts


def set_split_policy(min_run: int = 0, seg_len: int = 0, always: bool = False):
  """Parameters of the long-run segments for the frames enqueued from now on (process-wide; (0, 0, False) = defaults)."""
  global SPLIT_MIN_RUN, SPLIT_SEG_LEN, SPLIT_ALWAYS
  assert min_run >= 0 and seg_len >= 0
  SPLIT_MIN_RUN, SPLIT_SEG_LEN, SPLIT_ALWAYS = int(min_run), int(seg_len), bool(always)


def _choose_mapper(key, k_total: int, n: int):
  """Remember which launch sequence the next frame of this scene shape maps its tiles with (same lists either way).
  16 bit depth keys (the last entry of a shape key) always take the pre-sort: two passes over n pairs instead of four,
  4-byte pairs through the tile sort (1 M gaussians, K / n 2.45: 0.148 ms against 0.185)."""
  ratio = k_total / max(n, 1)
  now = _mapper_mode.get(key, _lib.MAPPER_DIRECT)
  if (key and key[-1] is True) or key in _presort_sticky:
    now = _lib.MAPPER_PRESORT
  elif ratio > PRESORT_ABOVE:
    now = _lib.MAPPER_PRESORT
  elif ratio < DIRECT_BELOW:
    now = _lib.MAPPER_DIRECT
  _mapper_mode[key] = now


def _shape_key(device, n, image_size, config, tile_rows, depth16):
  return (device.index, int(n), int(image_size[0]), int(image_size[1]), config.tile_size,
          None if tile_rows is None else (int(tile_rows[0]), int(tile_rows[1])), bool(depth16))


def _round_capacity(k: int) -> int:
  k = max(int(k), 1)
  return min((k + K_GRANULE - 1) // K_GRANULE * K_GRANULE, (1 << 31) - 1)


K_PENDING = -(1 << 31)      # sentinel the host writes into the pinned word before a frame; K itself is >= 0


class KSlots:
  """Pinned int32 words the frames' K kernels write (``ms_frame_project_count(k_host)``), one per frame in flight.

  Round 3 kept ONE word per device: a second frame enqueued before the first had looked at its overlap total (a
  viewer thread next to a training thread, two frames rendered back to back before either is settled) reset the
  word and the first frame then validated its capacity against the other frame's K."""

  def __init__(self):
    self.words = torch.zeros((K_SLOTS,), dtype=torch.int32).pin_memory()
    self.view = self.words.numpy()
    self.busy = [False] * K_SLOTS
    self.next = 0
    # words for frames captured into HIP graphs (kept for the life of the process: a graph replays into its word).
    # Allocated HERE, in eager mode: pinning host memory inside a stream capture invalidates the capture
    # (hipHostMalloc is not a capturable call — seen at 6 M gaussians, where torch's host cache had no block to reuse)
    self.graph_words = torch.zeros((GRAPH_K_WORDS,), dtype=torch.int32).pin_memory()
    self.graph_view = self.graph_words.numpy()
    self.graph_next = 0
    self.graph_free = []

  def acquire_for_graph(self, state):
    """a word for a frame being captured; it returns to the pool when the frame's state (kept alive by the graph's
    result) is garbage collected"""
    with _lock:
      if self.graph_free:
        i = self.graph_free.pop()
      else:
        i = self.graph_next
        if i >= GRAPH_K_WORDS:
          raise RuntimeError(f"more than {GRAPH_K_WORDS} live frames captured into HIP graphs in this process")
        self.graph_next = i + 1
    weakref.finalize(state, self.graph_free.append, i)
    return self.graph_words[i:i + 1], self.graph_view[i:i + 1]

  def acquire(self):
    """(slot index or -1, one-element pinned tensor, its numpy view)"""
    with _lock:
      for _ in range(K_SLOTS):
        i = self.next
        self.next = (i + 1) % K_SLOTS
        if not self.busy[i]:
          self.busy[i] = True
          return i, self.words[i:i + 1], self.view[i:i + 1]
    word = torch.zeros((1,), dtype=torch.int32).pin_memory()       # more than K_SLOTS unsettled frames: a private word
    return -1, word, word.numpy()

  def release(self, i):
    if i >= 0:
      self.busy[i] = False


def _k_ring(device) -> KSlots:
  with _lock:
    ring = _k_host.get(device.index)
    if ring is None:
      ring = _k_host[device.index] = KSlots()
  return ring


def _pinned_k(device):
  """A pinned int32 word of its own for the frame about to be enqueued: (slot, tensor, numpy view, event, ring)"""
  ring = _k_ring(device)
  slot, word, view = ring.acquire()
  return slot, word, view, torch.cuda.Event(), ring


def _wait_for_k(k_np, k_event, at_entry=False) -> int:
  """The host's one wait per eager frame.  Polling the pinned word the K kernel writes returns within microseconds of
  the write; ``Event.synchronize`` (an interrupt-driven sleep) cost 0.3-0.5 ms of wake-up latency per frame, which
  at 1 M gaussians left the GPU idle for a third of the frame."""
  global host_syncs, entry_waits
  deadline = time.perf_counter() + 2.0
  spins = 0
  while k_np[0] == K_PENDING:
    spins += 1
    if spins & 63 == 0:
      time.sleep(0)                         # let other Python threads run: the poll holds the GIL otherwise
      if time.perf_counter() > deadline:    # never seen; a lost write must not hang the caller
        k_event.synchronize()
        break
  if at_entry:
    entry_waits += 1 if spins else 0
  else:
    host_syncs += 1
  return int(k_np[0])


@contextlib.contextmanager
def parked_gc():
  """Run a training / timing loop with Python's cyclic collector parked.

  An eager frame keeps the GPU about one frame ahead of the host (the one wait per frame is for the overlap total).
  A generation-2 collection walks every tracked object of the process — with torch imported, 35-45 ms, i.e. ten
  config-D frames — about once every couple of hundred frames, and the GPU idles for most of it
  (``tools/host_overhead.py``: frame intervals median 3.40 ms, max 44 ms; 11 ms with the collector parked).
  Reference counting still frees every tensor of a frame when the frame ends; only cycle detection is deferred to
  the end of the block.  (A HIP-graph replay, ``FrameGraph``, needs no host per frame and is not affected.)"""
  was_enabled = gc.isenabled()
  gc.collect()
  gc.freeze()
  gc.disable()
  try:
    yield
  finally:
    gc.unfreeze()
    if was_enabled:
      gc.enable()


def release_caches(force: bool = False):
  settle_all()
  _release_caches(force)


def _release_caches(force: bool = False):
  """Drop what the executor keeps between frames: the persistent moments accumulators (64 B per gaussian), the shared
  identity index lists, the remembered overlap capacities and the pinned K words.  Accumulators whose address a
  captured HIP graph replays into (``FrameGraph`` / ``torch.cuda.graph``) are kept unless ``force`` — freeing them
  would let the next replay accumulate into memory the allocator has handed to someone else; pass ``force=True`` only
  after every such graph is gone."""
  with _lock:
    for key in list(_moments):
      if force or key not in _moments_pinned:
        del _moments[key]
    if force:
      _moments_pinned.clear()
    _identity.clear()
    _stable_frames.clear()
    _k_capacity.clear()
    _mapper_mode.clear()
    _presort_sticky.clear()
    # the run words stay (4 pinned bytes per scene shape): a frame still in flight, or a captured one at every replay,
    # writes its word — handing the memory back to the allocator would let that write land in somebody else's tensor
    for _, view in _run_words.values():
      view[0] = 0
    if force:
      _run_words.clear()
    if force:
      _k_host.clear()          # captured graphs write their overlap totals into these pinned words


def identity_indexes(n: int, device) -> torch.Tensor:
  """arange(n) int64, shared: callers treat index lists as read-only.  Built outside inference mode so that a first
  use under ``torch.inference_mode()`` (evaluation render) does not poison later training frames."""
  key = (device.index, int(n))
  cached = _identity.get(key)
  if cached is None:
    _identity.clear()
    with torch.inference_mode(False):
      cached = _identity[key] = torch.arange(n, dtype=torch.int64, device=device)
  return cached


def _moments_buffer(device, n: int, deterministic: bool) -> torch.Tensor:
  """Persistent (n, MOMENT_ROW) accumulator of the raster backward: zero on entry of every backward pass, and zero
  again when it returns (the per-gaussian pass clears the rows it reads) — no 64 B-per-gaussian fill per frame.

  One buffer per (device, stream, size, mode): two backward passes on DIFFERENT streams never share rows (on one
  stream they run one after the other and the rows are zero in between).  A small LRU per device instead of "one
  size at a time": alternating two scene sizes (a 3D model and ``rasterize()`` of a 2D set) no longer reallocates and
  zero-fills on every backward.  A buffer handed out during HIP-graph capture is pinned: its address is part of the
  graph."""
  stream = int(torch.cuda.current_stream(device).cuda_stream)
  key = (device.index, stream, int(n), bool(deterministic))
  with _lock:
    buf = _moments.get(key)
    if buf is None and torch.cuda.is_current_stream_capturing():
      # a capture runs on a stream of its own: take the accumulator the eager warm-up frames of this scene size used
      # (zero between frames) instead of allocating AND ZERO-FILLING a new one inside the graph — the fill would be
      # replayed with every step (384 MB at 6 M gaussians)
      for other, cand in _moments.items():
        if (other[0], other[2], other[3]) == (key[0], key[2], key[3]):
          buf = _moments[key] = cand
          _moments_pinned.add(other)
          break
    if buf is None:
      mine = [k for k in _moments if k[0] == device.index and k not in _moments_pinned]
      for k in mine[:max(0, len(mine) - (MOMENTS_LRU - 1))]:
        del _moments[k]                     # least recently used first (OrderedDict order)
      buf = _moments[key] = torch.zeros((max(n, 1), _lib.MOMENT_ROW),
                                        dtype=torch.int64 if deterministic else torch.float32, device=device)
    else:
      _moments.move_to_end(key)
    if torch.cuda.is_current_stream_capturing():
      _moments_pinned.add(key)
  return buf


def _drop_moments():
  """After a failed backward enqueue the accumulator rows may be half-written: start from clean ones.  Buffers whose
  address a captured HIP graph replays into are zeroed IN PLACE (freeing them would let the next replay accumulate into
  memory the allocator has handed to someone else — what ``release_caches`` guards against with ``force``); the others
  are dropped and reallocated on demand."""
  with _lock:
    for key in list(_moments):
      if key not in _moments_pinned:
        del _moments[key]
        continue
      buf = _moments[key]
      if torch.cuda.is_current_stream_capturing():
        continue              # (a failure inside a capture: the capture is lost anyway, and a fill must not join it)
      # on the buffer's OWN stream (key[1]): the graph that replays into it runs there, and a fill on whatever stream
      # is current could race with a replay in flight
      own = torch.cuda.ExternalStream(key[1], device=buf.device) if key[1] else torch.cuda.default_stream(buf.device)
      with torch.cuda.stream(own):
        buf.zero_()


class FrameState:
  """What one frame keeps between its forward pass, its backward pass and the lazily built ``points``."""

  def __init__(self):
    self.desc = None
    self.layout = None
    self.inputs = None
    self.keep_n = None
    self.keep_k = None
    self.k = None                 # overlap total (eager mode), None under graph capture
    self.capacity = 0
    self.children = []            # (weakref to a tensor handed out, index list or None, 'points7' | 'colours')
    self.y0 = 0
    self.pending = None           # eager mode: the look at K + capacity check, run once (by the frame's caller, or later)
    self.k_peek = None            # numpy view of the frame's pinned K word while `pending` is set
    self.consumed = False         # a backward pass was enqueued before the frame was settled
    self.key = None               # scene-shape key
    self.lazy = False             # render_frame left the look at K to the next frame's entry (LAZY_SETTLE)
    self.captured = False         # enqueued under HIP-graph capture: k_word / k_view = the pinned word replays write K to
    self.k_word = self.k_view = None
    self.overflowed = 0           # largest overlap total a replay was seen to exceed the capacity with (sticky)
    self.colours_ready = None     # event behind the SH pass on the executor's second stream (SH_SIDE_STREAM)
    self.vis_deferred = False     # the forward ran without visibility: the backward writes it (VISIBILITY_FROM_BACKWARD)
    self.vis_ready = True
    self.vis_args = None          # what the on-demand pass needs (detached views: no cycle through the autograd node)

  def ensure_visibility(self):
    """A deferred visibility (VISIBILITY_FROM_BACKWARD) that is read before the backward pass wrote it: the forward's
    visibility kernel on the frame's kept lists, into a scratch image."""
    global visibility_passes
    if not self.vis_deferred or self.vis_ready:
      return
    visibility, points7, colours, w, h, f, rows, y0, y1, config = self.vis_args
    self.settle()
    self.vis_ready = True
    visibility_passes += 1
    if y1 <= y0 or points7.shape[0] == 0 or self.capacity == 0 or self.keep_k is None:
      return
    lib = _lib.load()
    dtype, device = points7.dtype, points7.device
    cfg_v = _lib.raster_config_c(replace(config, compute_visibility=True, compute_point_heuristic=False))
    es = points7.element_size()
    with torch.no_grad():
      image = torch.empty((y1 - y0, w, f), dtype=dtype, device=device)
      alpha = torch.empty((y1 - y0, w), dtype=dtype, device=device)
      visibility.zero_()
      _lib.check(lib.ms_raster_fwd(points7.data_ptr(), colours.data_ptr(), self.tile_ranges().data_ptr(),
                                   self.overlap_to_point().data_ptr(), w, h, f, cfg_v,
                                   image.data_ptr() - y0 * w * f * es, alpha.data_ptr() - y0 * w * es, visibility.data_ptr(),
                                   rows[0], rows[1], _lib.dtype_code(dtype), _lib.current_stream(device)),
                 "render_gaussians (visibility on demand)")

  def settle(self):
    """Look at the frame's overlap total (waiting for it if the GPU has not produced it yet) and re-run the emission
    with larger buffers when it did not fit.  Raises FrameOverflow when that comes too late: the backward pass of the
    frame was already enqueued on the lists of the overflowed forward."""
    with _lock:
      fn, self.pending = self.pending, None      # (taken atomically: a viewer thread may settle the trainer's frames)
    if fn is not None:
      fn()

  def settle_if_known(self):
    """Backward pass of a lazily settled frame: settle only if that costs no wait.  Returns False when the overlap total
    is not there yet — the backward is then enqueued on the frame as it is (see LAZY_SETTLE)."""
    if self.pending is None:
      return True
    if not self.lazy:
      self.settle()                 # (a caller of the bare Function that has not settled: wait, as round 5 did)
      return True
    if self.k_peek is not None and int(self.k_peek[0]) == K_PENDING:
      self.consumed = True          # from here on a re-run of the forward could not repair this frame's gradients
      return False
    self.settle()
    return True

  def counters(self) -> torch.Tensor:
    self.settle()
    return self.keep_n[self.layout.counters:self.layout.counters + 32].view(torch.int32)

  def overlap_to_point(self) -> torch.Tensor:
    self.settle()
    off = self.layout.overlap_to_point
    return self.keep_k[off:off + 4 * self.capacity].view(torch.int32)

  def tile_ranges(self) -> torch.Tensor:
    self.settle()
    off = self.layout.tile_ranges
    ts = self.desc.raster.tile_size
    th, tw = (self.desc.image_h + ts - 1) // ts, (self.desc.image_w + ts - 1) // ts
    return self.keep_n[off:off + 8 * th * tw].view(torch.int32).view(th, tw, 2)

  def register(self, tensor: torch.Tensor, idx, kind: str):
    self.children.append((weakref.ref(tensor), idx, kind))

  def retained(self):
    out = []
    for ref, idx, kind in self.children:
      t = ref()
      if t is not None and t.requires_grad and t.retains_grad:
        out.append((t, idx, kind))
    return out


def _view(block: torch.Tensor, offset: int, dtype, shape):
  count = 1
  for s in shape:
    count *= s
  nbytes = count * torch.empty((), dtype=dtype).element_size()
  return block[offset:offset + nbytes].view(dtype).view(*shape)


def _strip_pixels(rows, tile_size, h):
  return min(rows[0] * tile_size, h), min(rows[1] * tile_size, h)


def _enqueue_forward(desc, inputs, keep_n, scratch_n, key, image_ptr, alpha_ptr, visibility, device, what, state,
                     settle_now=False):
  """The two forward calls of the executor plus the eager-mode capacity policy: everything is enqueued with the
  remembered capacity BEFORE the host looks at the overlap total (a pinned word the K kernel writes), and the emission
  is re-run with larger buffers in the rare case it did not fit.  Fills ``state`` (layout, keep_k, capacity, k) and
  leaves ``state.pending`` = the settle step (the wait + check) for the caller to run as the LAST thing it does for
  this frame — the later the host looks, the more of its own per-frame work is hidden behind queued GPU work."""
  lib = _lib.load()
  stream = _lib.current_stream(device)
  capturing = torch.cuda.is_current_stream_capturing()
  if not capturing:
    settle_all()                 # earlier frames of this process: their words are (almost always) written long ago
  # the scene-shape caches are shared by every thread that renders (a viewer next to a trainer): looked up and updated
  # under the lock
  with _lock:
    capacity = _k_capacity.get(key, 0)
    # what the per-tile sort of an earlier frame of this shape reported (no synchronisation: the word is a frame or two old)
    run_word = _run_words.get(key)
    if run_word is None and not capturing:
      t = torch.zeros((1,), dtype=torch.int32).pin_memory()
      run_word = _run_words[key] = (t, t.numpy())
    if run_word is not None:
      if int(run_word[1][0]) > LONG_RUN_LIMIT and key not in _presort_sticky:
        _presort_sticky.add(key)
        _mapper_mode[key] = _lib.MAPPER_PRESORT
      inputs.longest_run_host = run_word[0].data_ptr()
    # the same in every call of this frame.  The FIRST frame of a scene shape (nothing known about how its overlaps are
    # spread) takes the sequence whose cost does not depend on that — the pre-sort, 0.1-0.2 ms dearer on a scene like
    # config D — and carries the segment launches: a pile-up then costs its first frame what it costs every frame, not
    # the 7 ms of ONE workgroup sorting a 244 000-entry run plus one workgroup rasterizing it (round 5's first frame of
    # that shape: 11.1 ms against 1.67 steady; tools/sweep_scenes.py guards first <= 3 x steady).  The second frame
    # runs what _choose_mapper settled on.
    first_of_shape = key not in _mapper_mode
    desc.mapper = _mapper_mode.get(key, _lib.MAPPER_PRESORT)
    # a shape that showed a run above LONG_RUN_LIMIT also has its long runs rasterized in segments (the raster forward
    # reports such runs through the same word on either mapper sequence)
    # (field value 1 = the default threshold, > 1 = the threshold itself; the library raises thresholds below 256)
    split_on = SPLIT_LONG_RUNS and (SPLIT_ALWAYS or first_of_shape or key in _presort_sticky)
    desc.split_long_runs = (max(2, SPLIT_MIN_RUN) if SPLIT_MIN_RUN > 0 else 1) if split_on else 0
    desc.split_seg_len = SPLIT_SEG_LEN
  if capturing and capacity == 0:
    raise RuntimeError(f"{what} under HIP-graph capture: the overlap-list capacity of this scene shape is unknown; "
                       "render one eager frame first or call frame.set_overlap_capacity(...)")
  if capturing:
    # a pinned word of the captured frame's own: every replay writes its overlap total there, so the host can tell
    # an overflowed replay (background-only image, zero gradients) WITHOUT touching the device — see check_replays()
    slot, ring, k_event = -1, None, None
    k_word, k_np = _k_ring(device).acquire_for_graph(state)      # pinned in eager mode already (the warm-up frames)
    k_np[0] = 0
    state.k_word, state.k_view, state.captured = k_word, k_np, True
    _captured_frames.append(weakref.ref(state))
  else:
    slot, k_word, k_np, k_event, ring = _pinned_k(device)
    k_np[0] = K_PENDING
  try:
    _lib.check(lib.ms_frame_project_count(ctypes.byref(desc), ctypes.byref(inputs), keep_n.data_ptr(), scratch_n.data_ptr(),
                                          k_word.data_ptr(), None, stream), what)
  except Exception:
    if ring is not None:
      ring.release(slot)
    raise
  if not capturing:
    k_event.record(torch.cuda.current_stream(device))
  if (capturing and SH_SIDE_STREAM and desc.sh_degree >= 0 and not desc.projected_input and desc.n > 0
      and os.environ.get('MS_SPLAT_ROWS', '0') != '1'):
    main = torch.cuda.current_stream(device)
    with _lock:
      side = _side_streams.get(device.index)
      if side is None:
        side = _side_streams[device.index] = torch.cuda.Stream(device=device)
    side.wait_stream(main)                # behind the projection (and the overlap count and scan enqueued with it)
    _lib.check(lib.ms_frame_sh_colours(ctypes.byref(desc), ctypes.byref(inputs), keep_n.data_ptr(), side.cuda_stream), what)
    state.colours_ready = torch.cuda.Event()
    state.colours_ready.record(side)
    inputs.colours_ready_event = int(state.colours_ready.cuda_event)

  def map_raster(cap):
    desc.k_capacity = cap
    lay = _lib.FrameLayoutC()
    _lib.check(lib.ms_frame_layout_query(ctypes.byref(desc), ctypes.byref(lay)), what)
    keep_k = torch.empty((lay.keep_k_bytes,), dtype=torch.uint8, device=device)
    scratch_k = torch.empty((lay.scratch_k_bytes,), dtype=torch.uint8, device=device)
    _lib.check(lib.ms_frame_map_raster(ctypes.byref(desc), ctypes.byref(inputs), keep_n.data_ptr(), scratch_n.data_ptr(),
                                       keep_k.data_ptr(), scratch_k.data_ptr(), image_ptr, alpha_ptr,
                                       _lib.ptr(visibility), stream), what)
    state.layout, state.keep_k, state.capacity = lay, keep_k, cap

  state.k, state.pending = None, None
  state.key = key
  if capturing:
    map_raster(capacity)
    return

  def settle(at_entry=False):
    global settles
    state.pending = None
    state.k_peek = None
    with _lock:
      try:
        _unsettled.remove(state)
      except ValueError:
        pass
    try:
      k_total = _wait_for_k(k_np, k_event, at_entry)
    finally:
      ring.release(slot)        # this frame's word may serve another frame from here on
    settles += 1
    if k_total < 0:
      raise OverflowError(f"{what}: more than 2^31 - 1 tile overlaps (the overlap index is int32 like the "
                          "reference's, tile_mapper.py:150); use a larger tile size or fewer / smaller gaussians")
    overflowed = state.capacity == 0 or k_total > state.capacity
    with _lock:
      _k_capacity[key] = max(_k_capacity.get(key, 0), _round_capacity(k_total * K_SLACK))
      _choose_mapper(key, k_total, desc.n)
      _stable_frames[key] = 0 if overflowed else _stable_frames.get(key, 0) + 1
    if overflowed and state.consumed:
      # found too late: the backward of this frame ran on the empty lists of the overflowed forward
      raise FrameOverflow(
        f"{what}: a frame produced {k_total} tile overlaps but was enqueued with room for {state.capacity}, and its "
        "backward pass had been enqueued before the host looked (frame.LAZY_SETTLE): that frame rendered the background "
        "only and returned zero gradients.  The capacity has been raised for the next frame; set MS_STRICT=1 (or "
        "frame.LAZY_SETTLE = False) to make every frame wait for its overlap total before it returns.")
    if overflowed:
      if visibility is not None and k_total > 0:
        visibility.zero_()
      map_raster(_round_capacity(k_total * K_SLACK))
    state.k = k_total

  state.capacity = 0
  if capacity > 0:
    try:
      map_raster(capacity)        # everything is enqueued before the host looks at K
    except Exception:
      ring.release(slot)          # (an allocation failure here must not leave the frame's K word busy for ever)
      raise
  state.pending, state.k_peek = settle, k_np
  if capacity == 0 or settle_now:
    settle()


def settle_all():
  """Settle every eager frame the host has not looked at yet, oldest first (next-frame entry, graph capture, tests)."""
  while True:
    with _lock:
      if not _unsettled:
        return
      st = _unsettled[0]
    with _lock:
      fn, st.pending = st.pending, None
      if fn is None and _unsettled and _unsettled[0] is st:
        _unsettled.popleft()     # (settled by somebody else in the meantime)
    if fn is not None:
      fn(True)                   # removes itself from the queue first thing


def lazy_settle_allowed(key) -> bool:
  return LAZY_SETTLE and not STRICT and _stable_frames.get(key, 0) >= LAZY_AFTER


_captured_frames = []      # weak references to the FrameStates of frames captured into HIP graphs


class FrameOverflow(OverflowError):
  """A frame replayed from a HIP graph found more tile overlaps than its captured buffers hold: that replay's image is
  the background only and its gradients are zero."""


def check_replays(device=None, synchronize: bool = False):
  """Raise ``FrameOverflow`` if a captured frame's most recent replay (of those that have finished on the GPU; all of
  them with ``synchronize=True``) exceeded its overlap capacity.  Costs a host compare per captured frame: each frame
  owns a pinned word its K kernel rewrites on every replay.  Called by ``FrameGraph.replay`` (before launching the
  next replay; after it, synchronising, with ``MS_STRICT=1``), by ``LazyPoints`` and ``frame_status`` — i.e. at the
  next host touch, so a training loop replaying an overflowed graph stops within one step instead of silently
  training on background frames."""
  if synchronize:
    torch.cuda.synchronize(device)
  alive = []
  worst = None
  for ref in _captured_frames:
    st = ref()
    if st is None:
      continue
    alive.append(ref)
    if device is not None and st.keep_n is not None and st.keep_n.device != torch.device(device):
      continue
    k = int(st.k_view[0])
    if k < 0 or k > st.capacity:
      st.overflowed = max(st.overflowed, k if k >= 0 else (1 << 31) - 1)
      st.k_view[0] = 0             # taken note of; the next replay writes its own total
    if st.overflowed and (worst is None or st.overflowed > worst.overflowed):
      worst = st
  _captured_frames[:] = alive
  if worst is not None:
    need = worst.overflowed
    worst.overflowed = 0           # reported once; the caller re-captures (or not) knowingly
    raise FrameOverflow(
      f"a frame replayed from a HIP graph produced {need} tile overlaps but was captured with room for "
      f"{worst.capacity}: that replay rendered the background only and returned zero gradients.  Call "
      f"frame.set_overlap_capacity(n, image_size, config, {int(need * K_SLACK)}) and capture the step again.")


class _FrameFunction(torch.autograd.Function):
  """project -> SH -> map -> rasterize as one node (reference: perspective/projection.py:123-188,
  indexed_spherical_harmonics.py:138-160, rasterizer/function.py:42-95 chained by renderer.py:23-108)."""

  @staticmethod
  def forward(ctx, position, log_scaling, rotation, alpha_logit, feature, T_camera_world, projection,
              opts: FrameOptions, state: FrameState):
    lib = _lib.load()
    _lib.require_gpu(position, log_scaling, rotation, alpha_logit, feature, T_camera_world, projection)
    tensors = [t.detach().contiguous() for t in
               (position, log_scaling, rotation, alpha_logit, feature, T_camera_world, projection)]
    pos, lsc, rot, alog, feat, Tcw, proj = tensors
    dtype, device = pos.dtype, pos.device
    assert all(t.dtype == dtype for t in tensors), "render_gaussians: all inputs must share one dtype"
    config = opts.config
    n = pos.shape[0]
    w, h = int(opts.image_size[0]), int(opts.image_size[1])
    ts = config.tile_size
    tiles_high = (h + ts - 1) // ts

    if opts.use_sh:
      assert feat.ndim == 3, f"SH features must have 3 dimensions, got {feat.shape}"
      f, d = feat.shape[1], feat.shape[2]
      degree = int(round(d ** 0.5)) - 1
      assert (degree + 1) ** 2 == d, f"SH feature count must be square, got {d} ({feat.shape})"
      assert 0 <= degree <= 3, f"SH degree must be between 0 and 3, got {degree}"
    else:
      assert feat.ndim == 2, f"Features must be (N, C) if use_sh=False, got {feat.shape}"
      f, degree = feat.shape[1], -1

    rows = (0, tiles_high) if opts.tile_rows is None else (max(0, int(opts.tile_rows[0])), min(tiles_high, int(opts.tile_rows[1])))
    key = _shape_key(device, n, (w, h), config, opts.tile_rows, opts.use_depth16)
    desc = _lib.FrameDescC(n=n, k_capacity=_k_capacity.get(key, 0), image_w=w, image_h=h, dtype=_lib.dtype_code(dtype), f=f,
                           sh_degree=degree, depth16=int(opts.use_depth16), tile_row_begin=rows[0], tile_row_end=rows[1],
                           projected_input=0, mapper=0, near_plane=float(opts.depth_range[0]),
                           far_plane=float(opts.depth_range[1]), blur_cov=float(config.blur_cov),
                           clamp_margin=float(config.clamp_margin), raster=_lib.raster_config_c(config))
    layout = _lib.FrameLayoutC()
    _lib.check(lib.ms_frame_layout_query(ctypes.byref(desc), ctypes.byref(layout)), "render_gaussians")
    stream = _lib.current_stream(device)
    keep_n = torch.empty((layout.keep_n_bytes,), dtype=torch.uint8, device=device)
    scratch_n = torch.empty((layout.scratch_n_bytes,), dtype=torch.uint8, device=device)
    inputs = _lib.FrameInputsC(position=pos.data_ptr(), log_scaling=lsc.data_ptr(), rotation=rot.data_ptr(),
                               alpha_logit=alog.data_ptr(), feature=feat.data_ptr(), T_camera_world=Tcw.data_ptr(),
                               projection=proj.data_ptr(), points7=None, depth=None, colours=None)

    # images: with crop_to_rows only the strip's pixel rows exist; the kernels address absolute rows, so they get the
    # address row 0 WOULD have (they touch rows [y0, y1) only)
    y0, y1 = _strip_pixels(rows, ts, h) if opts.crop_to_rows else (0, h)
    image = torch.empty((y1 - y0, w, f), dtype=dtype, device=device)
    alpha = torch.empty((y1 - y0, w), dtype=dtype, device=device)
    if not opts.crop_to_rows and rows != (0, tiles_high):
      image.zero_(); alpha.zero_()
    visibility = torch.zeros((n,), dtype=dtype, device=device) if config.compute_visibility else torch.empty((0,), dtype=dtype, device=device)
    heuristic = torch.zeros((n, 2), dtype=dtype, device=device) if config.compute_point_heuristic else torch.empty((0, 2), dtype=dtype, device=device)
    es = image.element_size()
    state.desc, state.inputs, state.keep_n, state.y0 = desc, inputs, keep_n, y0
    state.tensors = tensors                                  # the pointers in `inputs` stay valid
    if y1 > y0:
      image_ptr, alpha_ptr = image.data_ptr() - y0 * w * f * es, alpha.data_ptr() - y0 * w * es
    else:
      # an empty cropped strip: zero-row tensors have a null data pointer, which the C entry points reject (and
      # `0 - y0 * ...` would wrap); the raster touches no row, any valid address will do
      state.dummy = torch.empty((16,), dtype=dtype, device=device)
      image_ptr = alpha_ptr = state.dummy.data_ptr()
    # a frame that will be differentiated may take its visibility from the backward pass (VISIBILITY_FROM_BACKWARD)
    defer_vis = bool(state.vis_deferred and config.compute_visibility and config.compute_point_heuristic and n > 0
                     and lib.ms_frame_uses_moments(ctypes.byref(desc), 0))
    state.vis_deferred, state.vis_ready = defer_vis, not defer_vis
    _enqueue_forward(desc, inputs, keep_n, scratch_n, key, image_ptr, alpha_ptr,
                     visibility if config.compute_visibility and not defer_vis else None, device,
                     "render_gaussians", state, settle_now=opts.render_median_depth)
    layout = state.layout

    points7 = _view(keep_n, layout.points7, dtype, (n, 7))
    depth = _view(keep_n, layout.depth, dtype, (n,))
    colours = _view(keep_n, layout.colours, dtype, (n, f)) if opts.use_sh else None
    if defer_vis:
      state.vis_args = (visibility.detach(), points7.detach(), (colours if opts.use_sh else feat).detach(), w, h, f, rows,
                        y0, y1, config)

    median = None
    if opts.render_median_depth:
      # renderer.py:77-82: quantile pass over the depths with the same tile lists
      cfg_m = _lib.raster_config_c(replace(config, use_alpha_blending=False, saturate_threshold=config.median_threshold,
                                           compute_visibility=False, compute_point_heuristic=False))
      median = torch.empty((y1 - y0, w, 1), dtype=dtype, device=device)
      median_alpha = torch.empty((y1 - y0, w), dtype=dtype, device=device)
      if not opts.crop_to_rows and rows != (0, tiles_high):
        median.zero_()
      if y1 > y0:
        _lib.check(lib.ms_raster_fwd(points7.data_ptr(), depth.data_ptr(), state.tile_ranges().data_ptr(),
                                     state.overlap_to_point().data_ptr(), w, h, 1, cfg_m,
                                     median.data_ptr() - y0 * w * es, median_alpha.data_ptr() - y0 * w * es, None,
                                     rows[0], rows[1], _lib.dtype_code(dtype), stream), "render_gaussians (median depth)")
      median = median.squeeze(-1)

    ctx.set_materialize_grads(False)
    ctx.state, ctx.opts = state, opts
    ctx.f, ctx.degree, ctx.rows = f, degree, rows
    ctx.save_for_backward(*tensors, image)
    ctx.heuristic = heuristic
    non_diff = [alpha, visibility, heuristic] + ([median] if median is not None else [])
    ctx.mark_non_differentiable(*non_diff)
    return image, alpha, points7, depth, colours, visibility, heuristic, median

  @staticmethod
  def backward(ctx, g_image, g_alpha, g_points7, g_depth, g_colours, g_vis, g_heur, g_median):
    from .rasterizer import function as raster_function
    lib = _lib.load()
    pos, lsc, rot, alog, feat, Tcw, proj, image = ctx.saved_tensors
    state, opts = ctx.state, ctx.opts
    state.settle_if_known()     # (done by render_frame unless the frame settles lazily: then only if it costs no wait)
    desc, config = state.desc, opts.config
    n, f = pos.shape[0], ctx.f
    device, dtype = pos.device, pos.dtype
    none = (None,) * 9
    if g_image is None and g_points7 is None and g_depth is None and g_colours is None:
      return none
    need = ctx.needs_input_grad
    if n == 0 or image.shape[0] == 0:
      zeros = [torch.zeros_like(t) if need[i] else None for i, t in enumerate((pos, lsc, rot, alog, feat, Tcw, proj))]
      return (*zeros, None, None)

    w, h = opts.image_size
    stream = _lib.current_stream(device)
    det = bool(raster_function.DETERMINISTIC_BACKWARD)
    moments_path = bool(lib.ms_frame_uses_moments(ctypes.byref(desc), int(det)))
    # dL/dimage of a sum / mean loss arrives as an EXPANDED scalar (strides 0): the moments kernel reads one pixel's f
    # values instead of an (H, W, f) copy that .contiguous() would write and the kernel read back (50 MB at 2048^2)
    broadcast = (BROADCAST_GRAD and moments_path and g_image is not None and g_image.dim() == 3 and g_image.shape[0] * g_image.shape[1] > 1
                 and g_image.stride(0) == 0 and g_image.stride(1) == 0)
    if broadcast:
      g_image = g_image[0, 0].contiguous()                     # (f,)
    else:
      g_image = g_image.contiguous() if g_image is not None else torch.zeros_like(image)
    retained = state.retained()
    want_points = any(kind == 'points7' for _, _, kind in retained)
    # camera pose optimisation with SH colours: the view direction depends on the camera position =
    # inverse(T_camera_world)[:3, 3] (perspective/params.py:62-65; the renderer detaches the gaussians' positions for
    # the SH evaluation, renderer.py:53, but not the camera's) — needs d(colour) as an array, see below
    sh_camera = bool(need[5] and opts.use_sh)
    want_colours = any(kind == 'colours' for _, _, kind in retained) or sh_camera

    gr = _lib.FrameGradsC()
    row_bytes = state.y0 * w * image.element_size()          # cropped strip: address of the (absent) row 0
    gr.image = image.data_ptr() - row_bytes * f
    gr.grad_image = g_image.data_ptr() - (0 if broadcast else row_bytes * f)
    gr.grad_image_broadcast = int(broadcast)
    extras = []
    for name, g in (('extra_points7', g_points7), ('extra_depth', g_depth), ('extra_colours', g_colours)):
      if g is not None:
        g = g.contiguous()
        extras.append(g)
        setattr(gr, name, g.data_ptr())

    grad_points7 = grad_colours = None
    fixed_exp = None
    if moments_path:
      moments = _moments_buffer(device, n, det)
      gr.moments = moments.data_ptr()
      gr.deterministic = int(det)
      if det:
        fixed_exp = _lib.fixed_point_exponents(g_image)
        gr.fixed_exp = fixed_exp.data_ptr()
      if want_points:
        grad_points7 = torch.empty((n, 7), dtype=dtype, device=device)
      if want_colours:
        grad_colours = torch.empty((n, f), dtype=dtype, device=device)
    else:
      grad_points7 = torch.zeros((n, 7), dtype=dtype, device=device)
      grad_colours = torch.zeros((n, f), dtype=dtype, device=device)
    gr.grad_points7, gr.grad_colours = _lib.ptr(grad_points7), _lib.ptr(grad_colours)

    grads = [torch.empty_like(t) if need[i] else None for i, t in enumerate((pos, lsc, rot, alog))]
    gr.grad_position, gr.grad_log_scaling, gr.grad_rotation, gr.grad_alpha_logit = (_lib.ptr(t) for t in grads)
    grad_feature = None
    if need[4]:
      if opts.use_sh or moments_path:
        grad_feature = torch.empty_like(feat)
        gr.grad_feature = grad_feature.data_ptr()
      else:
        grad_feature = grad_colours                 # plain colours: the raster backward's accumulator IS the gradient
    need_camera = need[5] or need[6]
    grad_camera = torch.zeros((16,), dtype=dtype, device=device) if need_camera else None
    gr.grad_camera = _lib.ptr(grad_camera)
    if config.compute_point_heuristic:
      gr.point_heuristic = ctx.heuristic.data_ptr()
    vis_here = state.vis_deferred and moments_path and state.vis_args is not None
    if vis_here:
      gr.point_visibility = state.vis_args[0].data_ptr()
    elif state.vis_deferred:
      state.ensure_visibility()           # (a backward that cannot deliver it: not on the moments path after all)

    try:
      with _lock:       # the two launches of a backward pass are enqueued back to back (ctypes releases the GIL)
        _lib.check(lib.ms_frame_backward(ctypes.byref(desc), ctypes.byref(state.inputs), state.keep_n.data_ptr(),
                                         state.keep_k.data_ptr(), ctypes.byref(gr), stream), "render_gaussians backward")
    except Exception:
      _drop_moments()         # the accumulator rows may have been left half-written: start from a fresh buffer
      raise
    if vis_here and not state.vis_ready:
      state.vis_ready = True

    if retained:
      # gaussians2d.retain_grad() / features.retain_grad() of the reference's trainers (renderer.py:103-108
      # viewspace_gradient): the 2D-boundary gradients never exist as autograd edges here, so they are handed out
      if not moments_path:
        total_p = grad_points7 if g_points7 is None else grad_points7 + g_points7
        total_c = grad_colours if g_colours is None else grad_colours + g_colours
      else:
        total_p, total_c = grad_points7, grad_colours
      for t, idx, kind in retained:
        total = total_p if kind == 'points7' else total_c
        t.grad = (total if idx is None else total[idx]).reshape(t.shape)

    grad_T = grad_proj = None
    if need_camera:
      if need[5]:
        grad_T = torch.zeros((4, 4), dtype=dtype, device=device)
        grad_T[:3] = grad_camera[:12].view(3, 4)
        if sh_camera:
          # d(colour) -> d(camera position) with the direction-gradient kernel of the modular SH operator (it re-reads
          # the SH rows: a pose-optimisation-only cost), then through the matrix inverse: A = T^-1,
          # dL/dT = -A^T (dL/dA) A^T with dL/dA non-zero in A[:3, 3] only
          total_c = grad_colours if (moments_path or g_colours is None) else grad_colours + g_colours
          g_cam = torch.zeros((3,), dtype=dtype, device=device)
          cam_pos = _view(state.keep_n, state.layout.camera_position, dtype, (3,))
          _lib.check(lib.ms_sh_bwd(feat.data_ptr(), pos.data_ptr(), identity_indexes(n, device).data_ptr(), cam_pos.data_ptr(),
                                   n, f, ctx.degree, None, total_c.data_ptr(), None, None, g_cam.data_ptr(), 0,
                                   _lib.dtype_code(dtype), stream), "render_gaussians backward (camera position)")
          A = torch.inverse(Tcw)
          dA = torch.zeros((4, 4), dtype=dtype, device=device)
          dA[:3, 3] = g_cam
          grad_T = grad_T - A.t() @ dA @ A.t()
      if need[6]:
        grad_proj = grad_camera[12:16].clone()
    return (*grads, grad_feature, grad_T, grad_proj, None, None)


class _RasterizeFrameFunction(torch.autograd.Function):
  """``rasterize(gaussians2d, depth, features, ...)`` (reference rasterizer/function.py:133-165: map_to_tiles +
  rasterize_with_tiles) on the executor's ``projected_input`` mode: one node, no host read of the overlap total
  before the frame is enqueued.  Differentiable w.r.t. gaussians2d and features, like the reference."""

  @staticmethod
  def forward(ctx, gaussians2d, depth, features, image_size, config, use_depth16, state):
    lib = _lib.load()
    _lib.require_gpu(gaussians2d, depth, features)
    assert gaussians2d.ndim == 2 and gaussians2d.shape[1] == 7, f"gaussians2d must be (N, 7), got {gaussians2d.shape}"
    assert features.ndim == 2 and features.shape[0] == gaussians2d.shape[0], \
      f"features must be (N, F), got {features.shape} for {gaussians2d.shape[0]} gaussians"
    assert features.dtype == gaussians2d.dtype, f"dtype mismatch {features.dtype} != {gaussians2d.dtype}"
    p = gaussians2d.detach().contiguous()
    feats = features.detach().contiguous()
    dtype, device = p.dtype, p.device
    d = depth.detach().reshape(-1).to(dtype).contiguous()      # sort keys are float32 bits of the depth either way
    n, f = feats.shape
    w, h = int(image_size[0]), int(image_size[1])
    key = ('2d',) + _shape_key(device, n, (w, h), config, None, use_depth16)
    desc = _lib.FrameDescC(n=n, k_capacity=_k_capacity.get(key, 0), image_w=w, image_h=h, dtype=_lib.dtype_code(dtype), f=f,
                           sh_degree=-1, depth16=int(use_depth16), tile_row_begin=0, tile_row_end=1 << 30,
                           projected_input=1, mapper=0, near_plane=0.0, far_plane=0.0, blur_cov=0.0, clamp_margin=0.0,
                           raster=_lib.raster_config_c(config))
    layout = _lib.FrameLayoutC()
    _lib.check(lib.ms_frame_layout_query(ctypes.byref(desc), ctypes.byref(layout)), "rasterize")
    keep_n = torch.empty((layout.keep_n_bytes,), dtype=torch.uint8, device=device)
    scratch_n = torch.empty((layout.scratch_n_bytes,), dtype=torch.uint8, device=device)
    inputs = _lib.FrameInputsC(points7=p.data_ptr(), depth=d.data_ptr(), colours=feats.data_ptr())
    image = torch.empty((h, w, f), dtype=dtype, device=device)
    alpha = torch.empty((h, w), dtype=dtype, device=device)
    visibility = torch.zeros((n,), dtype=dtype, device=device) if config.compute_visibility else torch.empty((0,), dtype=dtype, device=device)
    heuristic = torch.zeros((n, 2), dtype=dtype, device=device) if config.compute_point_heuristic else torch.empty((0, 2), dtype=dtype, device=device)
    state.desc, state.inputs, state.keep_n = desc, inputs, keep_n
    state.tensors = (p, d, feats)
    _enqueue_forward(desc, inputs, keep_n, scratch_n, key, image.data_ptr(), alpha.data_ptr(),
                     visibility if config.compute_visibility else None, device, "rasterize", state)
    ctx.set_materialize_grads(False)
    ctx.state, ctx.config, ctx.size, ctx.heuristic = state, config, (w, h), heuristic
    ctx.save_for_backward(p, feats, image)
    ctx.mark_non_differentiable(alpha, heuristic, visibility)
    return image, alpha, heuristic, visibility

  @staticmethod
  def backward(ctx, g_image, g_alpha, g_heur, g_vis):
    from .rasterizer import function as raster_function
    lib = _lib.load()
    p, feats, image = ctx.saved_tensors
    state, config = ctx.state, ctx.config
    state.settle_if_known()
    need_points, _, need_features = ctx.needs_input_grad[:3]
    heuristic = ctx.heuristic if config.compute_point_heuristic else None
    if g_image is None or not (need_points or need_features or heuristic is not None):
      return (None,) * 7
    n, f = feats.shape
    device, dtype = p.device, p.dtype
    if n == 0:
      return torch.zeros_like(p), None, torch.zeros_like(feats), None, None, None, None
    det = bool(raster_function.DETERMINISTIC_BACKWARD)
    moments_path = bool(lib.ms_frame_uses_moments(ctypes.byref(state.desc), int(det)))
    make = torch.empty if moments_path else torch.zeros        # the finalize pass stores, the generic kernels accumulate
    gp = make((n, 7), dtype=dtype, device=device)
    gf = make((n, f), dtype=dtype, device=device)
    g_image = g_image.contiguous()
    gr = _lib.FrameGradsC()
    gr.image, gr.grad_image = image.data_ptr(), g_image.data_ptr()
    gr.grad_points7, gr.grad_colours = gp.data_ptr(), gf.data_ptr()
    fixed_exp = None
    if moments_path:
      gr.moments = _moments_buffer(device, n, det).data_ptr()
      gr.deterministic = int(det)
      if det:
        fixed_exp = _lib.fixed_point_exponents(g_image)
        gr.fixed_exp = fixed_exp.data_ptr()
    if heuristic is not None:
      gr.point_heuristic = heuristic.data_ptr()
    try:
      with _lock:
        _lib.check(lib.ms_frame_backward(ctypes.byref(state.desc), ctypes.byref(state.inputs), state.keep_n.data_ptr(),
                                         state.keep_k.data_ptr(), ctypes.byref(gr), _lib.current_stream(device)),
                   "rasterize backward")
    except Exception:
      _drop_moments()
      raise
    return (gp if need_points else None), None, (gf if need_features else None), None, None, None, None


def rasterize_frame(gaussians2d, depth, features, image_size, config: RasterConfig, use_depth16: bool = False):
  """``rasterize`` on the frame executor; returns (image, image_weight, point_heuristic, visibility)."""
  state = FrameState()
  out = _RasterizeFrameFunction.apply(gaussians2d, depth, features, tuple(int(x) for x in image_size), config,
                                      bool(use_depth16), state)
  if out[0].requires_grad and state.pending is not None and state.key is not None and lazy_settle_allowed(state.key):
    state.lazy = True
    with _lock:
      _unsettled.append(state)
  else:
    state.settle()
  return out


class LazyPoints:
  """Builds the ``RenderedPoints`` of a frame on first access (``Rendering.points``): the compacted ``(V, ...)``
  arrays the reference's projection returns (perspective/projection.py:147-150) are made from the full ones here,
  with the reference's one host synchronisation on the visible count."""

  def __init__(self, state: FrameState, gaussians, points7, depth, colours, visibility, heuristic, config, use_sh):
    self.args = (state, gaussians, points7, depth, colours, visibility, heuristic, config, use_sh)

  def materialise(self):
    from .rendering import RenderedPoints
    global point_syncs
    state, gaussians, points7, depth, colours, visibility, heuristic, config, use_sh = self.args
    n = depth.shape[0]
    state.settle()
    state.ensure_visibility()
    with torch.no_grad():
      mask = depth > 0
      v = int(mask.sum().item())
    point_syncs += 1
    if state.captured:
      check_replays(depth.device)      # the host has just synchronised: an overflowed replay is reported here
    if v == n:
      idx, sel = identity_indexes(n, depth.device), None
      g2d, dep = points7, depth.unsqueeze(1)
      feats = colours if use_sh else gaussians.feature[idx]
      vis = visibility if config.compute_visibility else None
      heur = heuristic if config.compute_point_heuristic else None
    else:
      idx = sel = mask.nonzero().squeeze(1)
      g2d, dep = points7[idx], depth[idx].unsqueeze(1)
      feats = colours[idx] if use_sh else gaussians.feature[idx]
      vis = visibility[idx] if config.compute_visibility else None
      heur = heuristic[idx] if config.compute_point_heuristic else None
    if g2d.requires_grad:
      state.register(g2d, sel, 'points7')
    if use_sh and feats.requires_grad:
      state.register(feats, sel, 'colours')
    return RenderedPoints(
      idx=idx, depths=dep, gaussians2d=g2d,
      _visibility=vis,
      _prune_cost=heur[:, 0] if heur is not None else None,
      _split_score=heur[:, 1] if heur is not None else None,
      features=feats, attributes=None, batch_size=(v,))


def render_frame(gaussians, camera_params, config: RasterConfig, use_sh: bool, use_depth16: bool = False,
                 render_median_depth: bool = False, tile_rows=None, crop_to_rows: bool = False):
  """``render_gaussians`` on the frame executor.  Returns a ``Rendering`` whose ``points`` are built lazily."""
  from .rendering import Rendering
  opts = FrameOptions(image_size=tuple(int(x) for x in camera_params.image_size),
                      depth_range=tuple(float(x) for x in camera_params.depth_range), config=config,
                      use_sh=bool(use_sh), use_depth16=bool(use_depth16), tile_rows=tile_rows,
                      crop_to_rows=bool(crop_to_rows), render_median_depth=bool(render_median_depth))
  state = FrameState()
  args = (*gaussians.shape_tensors(), gaussians.feature, camera_params.T_camera_world.reshape(4, 4),
          camera_params.projection.reshape(4))
  # (asked for here: inside Function.forward the grad mode is off whatever the caller's is)
  state.vis_deferred = bool(VISIBILITY_FROM_BACKWARD and config.compute_visibility and config.compute_point_heuristic
                            and torch.is_grad_enabled() and any(t.requires_grad for t in args))
  image, alpha, points7, depth, colours, visibility, heuristic, median = _FrameFunction.apply(*args, opts, state)
  points = LazyPoints(state, gaussians, points7, depth, colours, visibility, heuristic, config, use_sh)
  rendering = Rendering(image=image, image_weight=alpha, depth_image=None, median_depth_image=median, points=points,
                        camera=camera_params, config=config)
  object.__setattr__(rendering, 'frame', state)
  # the host's look at the overlap total: last, behind everything it had to do anyway — or, for a frame that is about to
  # be differentiated on a scene shape with a settled capacity, not before the next frame (LAZY_SETTLE)
  if image.requires_grad and state.pending is not None and state.key is not None and lazy_settle_allowed(state.key):
    state.lazy = True
    with _lock:
      _unsettled.append(state)     # looked at by the next frame's entry (settle_all) or the first host access
  else:
    state.settle()
  return rendering


def point_outputs(rendering) -> dict:
  """The frame's per-gaussian outputs at FULL size (one row per input gaussian, culled ones zero), without the host read of
  the visible count that ``rendering.points`` costs: ``visibility`` (n,), ``point_heuristic`` (n, 2), ``depth`` (n,) — what a
  training loop hands to ``VisibilityAwareAdam.step(indexes=None, visibility=...)`` (dense mode) and accumulates for
  densification."""
  points = object.__getattribute__(rendering, 'points')
  if hasattr(points, 'args'):
    state, _, _, depth, _, visibility, heuristic, config, _ = points.args
    state.ensure_visibility()
    return {"visibility": visibility if config.compute_visibility else None,
            "point_heuristic": heuristic if config.compute_point_heuristic else None, "depth": depth}
  n = rendering.frame.desc.n if hasattr(rendering, 'frame') else int(points.idx.max()) + 1
  return {"visibility": points.full_visibility(n) if points._visibility is not None else None,
          "point_heuristic": None, "depth": None}


def frame_status(rendering) -> dict:
  """Host read of a frame's device-side counters: overlap total, capacity, overflow flag (synchronises)."""
  state = getattr(rendering, 'frame', None)
  assert state is not None, "frame_status: not a rendering of the frame executor"
  state.settle()
  k, live, overflow = state.counters()[:3].tolist()
  if overflow and state.captured:
    state.overflowed = 0               # reported here; check_replays() need not raise for it again
  return {"overlaps": k, "capacity": state.capacity, "overflow": bool(overflow)}


class FrameGraph:
  """A training / rendering step captured in a HIP graph (``torch.cuda.CUDAGraph``) and replayed with one launch.

  ``step()`` is any callable that renders with ``render_gaussians`` (and usually runs ``backward``) on STATIC tensors:
  the gaussians' parameter tensors, the camera tensors and whatever the loss reads keep their storage, and new
  values (a new camera pose, updated parameters) are written into them in place between replays.  The frame executor
  never goes back to the host, so the whole step — about 35 kernel launches for forward + backward — is one graph.
  The overlap-list capacity is the one remembered from the eager warm-up frames.  A replay that exceeds it renders
  the background only and returns zero gradients; it does NOT pass silently: every captured frame owns a pinned word
  its K kernel rewrites on each replay, ``replay()`` compares the words with the capacities before launching the next
  replay and raises ``FrameOverflow`` (so a training loop stops within one step of the overflow), and with
  ``MS_STRICT=1`` (or ``strict=True``) it synchronises after the launch and raises for that very replay.
  ``frame_status(result)`` reads the device-side counters.  Re-capture after ``set_overlap_capacity``.  Do not touch ``result.points`` inside ``step`` (it reads the visible count back), and
  drop every reference to renderings / losses of earlier eager steps first: torch's rule for whole-step capture — an
  autograd graph created on the default stream that is still alive pulls its gradient accumulation onto that stream
  and breaks the capture.
  """

  def __init__(self, step, warmup: int = 2, strict: Optional[bool] = None):
    self.strict = STRICT if strict is None else bool(strict)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      for _ in range(max(1, warmup)):            # eager frames: allocator warm-up and the capacity of this scene shape
        step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    settle_all()                                 # (the warm-up frames may settle lazily)
    self.graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(self.graph):
      self.result = step()

  def replay(self):
    check_replays()                    # earlier replays that have finished: host compare only, no synchronisation
    self.graph.replay()
    if self.strict:
      check_replays(synchronize=True)
    return self.result
