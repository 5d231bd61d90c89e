"""Round 6, CPU: host-side bookkeeping added this round (no kernels run)."""
import ctypes

import pytest
import torch

from taichi_splatting_amd import _lib, frame


def test_split_policy_and_lazy_settle_switches():
  try:
    frame.set_split_policy(min_run=512, seg_len=256, always=True)
    assert (frame.SPLIT_MIN_RUN, frame.SPLIT_SEG_LEN, frame.SPLIT_ALWAYS) == (512, 256, True)
    with pytest.raises(AssertionError):
      frame.set_split_policy(min_run=-1)
  finally:
    frame.set_split_policy()
  assert (frame.SPLIT_MIN_RUN, frame.SPLIT_SEG_LEN, frame.SPLIT_ALWAYS) == (0, 0, False)
  # lazy settle: opt-in, and only for shapes whose capacity has been stable for LAZY_AFTER settled frames
  key = ('shape',)
  default = frame.LAZY_SETTLE
  try:
    frame.LAZY_SETTLE = False
    frame._stable_frames[key] = 10
    assert not frame.lazy_settle_allowed(key)
    frame.LAZY_SETTLE = True
    assert frame.lazy_settle_allowed(key) == (not frame.STRICT)
    frame._stable_frames[key] = frame.LAZY_AFTER - 1
    assert not frame.lazy_settle_allowed(key)
    assert not frame.lazy_settle_allowed(('unknown',))
  finally:
    frame.LAZY_SETTLE = default
    frame._stable_frames.pop(key, None)
  frame.settle_all()                       # nothing queued: returns at once


def test_frame_state_settles_once_and_only_lazy_frames_skip_the_wait():
  calls = []
  st = frame.FrameState()
  st.pending = lambda at_entry=False: calls.append(at_entry)
  st.settle(); st.settle()
  assert calls == [False] and st.pending is None
  # backward of a frame that was NOT queued lazily waits (round 5's behaviour for callers of the bare Function)
  st = frame.FrameState()
  st.pending = lambda at_entry=False: calls.append('waited')
  import numpy as np
  st.k_peek = np.array([frame.K_PENDING], dtype=np.int32)
  assert st.settle_if_known() and calls[-1] == 'waited' and not st.consumed
  # a lazily queued frame whose total is not there yet: the backward goes ahead and the frame is marked
  st = frame.FrameState()
  st.lazy = True
  st.pending = lambda at_entry=False: calls.append('never')
  st.k_peek = np.array([frame.K_PENDING], dtype=np.int32)
  assert st.settle_if_known() is False and st.consumed and calls[-1] != 'never'
  st.k_peek[0] = 1234                       # ... and once it is there, settling costs no wait
  assert st.settle_if_known() and calls[-1] == 'never'


def test_sized_structs_take_keywords_only():
  with pytest.raises(AssertionError):
    _lib.FrameDescC(1000)
  d = _lib.FrameDescC(n=5)
  assert d.n == 5 and d.struct_size == ctypes.sizeof(_lib.FrameDescC) and d.abi_version == _lib.ABI_VERSION
  g = _lib.OptimGroupC(d=3)
  assert ctypes.sizeof(g) % 8 == 0


def test_optimiser_step_refuses_cpu_tensors_and_bad_dense_shapes():
  from taichi_splatting_amd.optim import ParameterClass, VisibilityAwareAdam
  n = 10
  params = ParameterClass(dict(position=torch.randn(n, 3)), dict(position=dict(lr=0.1)), optimizer=VisibilityAwareAdam)
  params.position.grad = torch.randn(n, 3)
  with pytest.raises(AssertionError, match="one visibility per point"):
    params.step(indexes=None, visibility=torch.rand(n - 1))
  with pytest.raises(RuntimeError, match="no CPU fallback"):
    params.step(indexes=None, visibility=torch.rand(n))
  with pytest.raises(AssertionError, match="shape mismatch"):
    params.step(indexes=torch.arange(4), visibility=torch.rand(5))


def test_deferred_visibility_bookkeeping_without_a_device():
  """frame.VISIBILITY_FROM_BACKWARD is opt-in; a frame state that is not deferred, or whose backward pass has written the
  sums, never runs the pass on demand (no library call is made here: the early returns)."""
  assert frame.VISIBILITY_FROM_BACKWARD is False and frame.SH_SIDE_STREAM is True
  st = frame.FrameState()
  assert (st.vis_deferred, st.vis_ready, st.vis_args, st.colours_ready) == (False, True, None, None)
  passes = frame.visibility_passes
  st.ensure_visibility()                                       # not deferred
  st.vis_deferred, st.vis_ready = True, True
  st.ensure_visibility()                                       # deferred and already written by the backward pass
  assert frame.visibility_passes == passes
  # the grads struct carries the pointer the per-gaussian pass writes the sums through, and the header agrees on its place
  # (tests/test_abi.py holds every offset against a C compiler)
  gr = _lib.FrameGradsC()
  assert gr.point_visibility is None and gr.struct_size == ctypes.sizeof(_lib.FrameGradsC)
  names = [f[0] for f in _lib.FrameGradsC._fields_]
  assert names.index('point_visibility') == names.index('point_heuristic') + 1
  assert 'ms_frame_sh_colours' in _lib.SIGNATURES
