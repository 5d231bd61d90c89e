"""Integer primitives of the tile mapper, same Python surface as the reference's ``cuda_lib``
(``cuda_lib/__init__.py:16-41``) but backed by hand-written gfx950 kernels (csrc/scan_sort.hip)
instead of CUB.  The module keeps its name so that ``from taichi_splatting import cuda_lib``
call sites keep working; ``hip_lib`` is an alias.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from .. import _lib

_KEY_BYTES = {torch.int32: 4, torch.uint32: 4, torch.int64: 8, torch.uint64: 8,
              torch.int16: 2, torch.uint16: 2}


def check_cuda(name, arg):
  assert arg.is_cuda, f"{name}: device must be a cuda device, got {arg.device}"


def _scratch(nbytes: int, device) -> torch.Tensor:
  return torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=device)


def full_cumsum(x: torch.Tensor, return_total: bool = True) -> Tuple[torch.Tensor, int]:
  """Exclusive scan with the total appended: returns (out (n+1,), total) — full_cumsum.cu:51-67.

  Reading the total back is the one host synchronisation of the mapper (it sizes the key buffers).
  """
  check_cuda("full_cumsum", x)
  assert x.dtype == torch.int32 and x.ndim == 1, "full_cumsum: int32 vector expected"
  if x.shape[0] == 0:
    return x.new_zeros((1,)), 0
  lib = _lib.load()
  x = x.contiguous()
  n = x.shape[0]
  out = x.new_empty((n + 1,))
  nbytes = ctypes.c_size_t(0)
  stream = _lib.current_stream(x.device)
  _lib.check(lib.ms_exclusive_scan_i32(None, n, None, None, None, ctypes.byref(nbytes), stream), "full_cumsum")
  tmp = _scratch(nbytes.value, x.device)
  _lib.check(lib.ms_exclusive_scan_i32(x.data_ptr(), n, out.data_ptr(), None, tmp.data_ptr(),
                                       ctypes.byref(nbytes), stream), "full_cumsum")
  total = int(out[n].item()) if return_total else -1
  return out, total


def radix_sort_pairs(keys: torch.Tensor, values: torch.Tensor, start_bit: int = 0,
                     end_bit: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
  """Stable LSD radix sort of (key, int32 value) pairs on key bits [start_bit, end_bit).

  Keys are ordered as unsigned integers, except full-width sorts of signed dtypes which order as
  signed (cub semantics), implemented by flipping the sign bit.
  """
  check_cuda("keys", keys)
  check_cuda("values", values)
  assert keys.ndim == 1 and values.ndim == 1 and keys.shape[0] == values.shape[0], \
    f"radix_sort_pairs: size mismatch {keys.shape} {values.shape}"
  assert values.dtype == torch.int32, "radix_sort_pairs: values must be int32"
  if keys.dtype not in _KEY_BYTES:
    raise ValueError(f"radix_sort_pairs: unsupported key type {keys.dtype}")

  key_bytes = _KEY_BYTES[keys.dtype]
  if end_bit is None or end_bit < 0:
    end_bit = key_bytes * 8
  signed = keys.dtype in (torch.int16, torch.int32, torch.int64)
  orig_dtype = keys.dtype

  if key_bytes == 2:   # widen 16 bit keys; order preserved
    keys = keys.to(torch.int32) if signed else keys.to(torch.int32) & 0xffff
    key_bytes = 4
    if signed and end_bit == 16:
      end_bit = 32

  flip = signed and end_bit == key_bytes * 8
  work = keys.contiguous()
  if flip:
    sign = torch.tensor(-(1 << (key_bytes * 8 - 1)), dtype=work.dtype, device=work.device)
    work = work ^ sign

  n = work.shape[0]
  keys_out = torch.empty_like(work)
  values_out = torch.empty_like(values)
  if n > 0:
    lib = _lib.load()
    values = values.contiguous()
    nbytes = ctypes.c_size_t(0)
    stream = _lib.current_stream(work.device)
    _lib.check(lib.ms_radix_sort_pairs(None, None, None, None, n, key_bytes, start_bit, end_bit,
                                       None, ctypes.byref(nbytes), stream), "radix_sort_pairs")
    tmp = _scratch(nbytes.value, work.device)
    _lib.check(lib.ms_radix_sort_pairs(work.data_ptr(), values.data_ptr(), keys_out.data_ptr(),
                                       values_out.data_ptr(), n, key_bytes, start_bit, end_bit,
                                       tmp.data_ptr(), ctypes.byref(nbytes), stream), "radix_sort_pairs")
  if flip:
    keys_out = keys_out ^ sign
  if keys_out.dtype != orig_dtype:
    keys_out = keys_out.to(orig_dtype)
  return keys_out, values_out


def radix_argsort(keys: torch.Tensor) -> torch.Tensor:
  idx = torch.arange(keys.shape[0], dtype=torch.int32, device=keys.device)
  _, idx = radix_sort_pairs(keys, idx)
  return idx


def segmented_sort_pairs(keys: torch.Tensor, values: torch.Tensor, start_offsets: torch.Tensor,
                         end_offsets: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
  """Sort every segment [start, end) by key (segmented_sort_pairs.cu:35-73); not on the render path."""
  check_cuda("keys", keys)
  check_cuda("values", values)
  assert keys.dtype in (torch.int32, torch.int16) and values.dtype == torch.int32
  assert start_offsets.dtype == torch.int64 and end_offsets.dtype == torch.int64
  orig = keys.dtype
  k = keys.to(torch.int32).contiguous()
  v = values.contiguous()
  ko, vo = torch.empty_like(k), torch.empty_like(v)
  lib = _lib.load()
  _lib.check(lib.ms_segmented_sort_pairs(k.data_ptr(), v.data_ptr(), ko.data_ptr(), vo.data_ptr(),
                                         k.shape[0], start_offsets.contiguous().data_ptr(),
                                         end_offsets.contiguous().data_ptr(), start_offsets.shape[0],
                                         _lib.current_stream(k.device)), "segmented_sort_pairs")
  return ko.to(orig), vo


__all__ = ["full_cumsum", "radix_sort_pairs", "segmented_sort_pairs", "radix_argsort"]
