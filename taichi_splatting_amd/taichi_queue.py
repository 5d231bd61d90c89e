"""Compatibility shim for the reference's ``TaichiQueue`` (``taichi_queue.py:34-90``).

The reference serialises every Taichi call onto one (optional) worker thread and owns
``ti.init``.  There is no Taichi runtime here: HIP kernels are launched directly on the current
torch stream through the C-ABI library, so the queue degenerates to "run inline".  The class is
kept so that callers written against the reference (``TaichiQueue.init(...)``,
``with taichi_queue(...)``, ``@queued``) work unchanged.
"""
from __future__ import annotations

from concurrent.futures import Future


class TaichiQueueContext:
  def __init__(self, *args, **kwargs):
    self.args, self.kwargs = args, kwargs

  def __enter__(self):
    TaichiQueue.init(*self.args, **self.kwargs)

  def __exit__(self, exc_type, exc_value, traceback):
    TaichiQueue.stop()


def taichi_queue(*args, **kwargs):
  return TaichiQueueContext(*args, **kwargs)


class TaichiQueue:
  initialised = False

  @classmethod
  def init(cls, *args, threaded=False, **kwargs) -> None:
    """Accepts (and ignores) ``ti.init`` arguments: arch, log_level, debug, device_memory_GB ..."""
    from . import _lib
    cls.initialised = True
    _lib.load()   # fail early and loudly if the HIP library is missing

  @staticmethod
  def thread_id():
    return None

  @staticmethod
  def run_async(func, *args, **kwargs) -> Future:
    args = [a.result() if isinstance(a, Future) else a for a in args]
    future = Future()
    future.set_result(func(*args, **kwargs))
    return future

  @staticmethod
  def run_sync(func, *args, **kwargs):
    return TaichiQueue.run_async(func, *args, **kwargs).result()

  @classmethod
  def stop(cls) -> None:
    cls.initialised = False


def queued(kernel):
  def f(*args, **kwargs):
    return TaichiQueue.run_sync(kernel, *args, **kwargs)
  return f
