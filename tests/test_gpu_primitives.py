"""-m gpu: scan / radix sort / ranges against torch (bit-exact integer work)."""
import numpy as np
import pytest
import torch

from taichi_splatting_amd import cuda_lib

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('n', [0, 1, 63, 64, 4096, 4097, 100003, 3_000_000])
def test_full_cumsum(n):
  torch.manual_seed(n)
  x = torch.randint(0, 9, (n,), dtype=torch.int32, device=DEV)
  out, total = cuda_lib.full_cumsum(x)
  want = torch.cat([torch.zeros(1, dtype=torch.int64, device=DEV), torch.cumsum(x.long(), 0)])
  assert out.shape[0] == n + 1
  assert torch.equal(out.long(), want)
  assert total == int(want[-1])


@pytest.mark.parametrize('n', [1, 100, 4096, 4097, 250_001, 2_000_003])
@pytest.mark.parametrize('dtype,end_bit', [(torch.int64, 48), (torch.int64, 49), (torch.int64, 64),
                                           (torch.int32, 32), (torch.int32, 20)])
def test_radix_sort_pairs_stable(n, dtype, end_bit):
  torch.manual_seed(n + end_bit)
  bits = end_bit if end_bit < 63 else 62
  # few distinct keys in the low range => many ties => stability is exercised
  keys = torch.randint(0, 1 << min(bits, 40), (n,), dtype=torch.int64, device=DEV)
  keys[::3] = keys[::3] % 17
  if dtype == torch.int32:
    keys = (keys % (1 << min(bits, 31))).to(torch.int32)
  values = torch.arange(n, dtype=torch.int32, device=DEV)
  ko, vo = cuda_lib.radix_sort_pairs(keys, values, end_bit=end_bit)
  wk, order = torch.sort(keys.long(), stable=True)
  assert torch.equal(ko.long(), wk)
  assert torch.equal(vo.long(), order)
  # inputs untouched
  assert torch.equal(values, torch.arange(n, dtype=torch.int32, device=DEV))


def test_radix_sort_signed_and_partial_bits():
  torch.manual_seed(0)
  keys = torch.randint(-1000, 1000, (50000,), dtype=torch.int32, device=DEV)
  vals = torch.arange(50000, dtype=torch.int32, device=DEV)
  ko, vo = cuda_lib.radix_sort_pairs(keys, vals)
  wk, order = torch.sort(keys, stable=True)
  assert torch.equal(ko, wk) and torch.equal(vo.long(), order)
  # bit range [8, 16): sort by that byte only, stable
  k2 = torch.randint(0, 1 << 24, (50000,), dtype=torch.int32, device=DEV)
  ko, vo = cuda_lib.radix_sort_pairs(k2, vals, start_bit=8, end_bit=16)
  wk, order = torch.sort((k2 >> 8) & 0xff, stable=True)
  assert torch.equal((ko >> 8) & 0xff, wk) and torch.equal(vo.long(), order)
  assert torch.equal(cuda_lib.radix_argsort(keys).long(), torch.sort(keys, stable=True)[1])


def test_segmented_sort_pairs():
  torch.manual_seed(1)
  k = torch.randint(0, 100, (2000,), dtype=torch.int32, device=DEV)
  v = torch.arange(2000, dtype=torch.int32, device=DEV)
  starts = torch.tensor([0, 700, 700, 1500], dtype=torch.int64, device=DEV)
  ends = torch.tensor([700, 700, 1500, 2000], dtype=torch.int64, device=DEV)
  ko, vo = cuda_lib.segmented_sort_pairs(k, v, starts, ends)
  for s, e in zip(starts.tolist(), ends.tolist()):
    wk, order = torch.sort(k[s:e], stable=True)
    assert torch.equal(ko[s:e], wk)
    assert torch.equal(vo[s:e].long(), order + s)


@pytest.mark.parametrize('n,resolution', [(1, 0.01), (1000, 0.001), (300000, 0.01), (50000, 2.0)])
def test_morton_codes_and_argsort_match_oracle(n, resolution):
  # ms_morton_codes64 + 64-bit radix argsort against oracle/morton.py (bit-exact codes, identical stable order)
  from oracle import morton as omorton
  from taichi_splatting_amd.misc import morton_sort
  torch.manual_seed(n)
  pts = (torch.rand(n, 3) * torch.tensor([8.0, 3.0, 20.0]) - 2.0)
  if n > 10:
    pts[3] = pts[7]                                   # duplicates: stable order decides
  dev = pts.cuda()
  codes = morton_sort.morton_codes(dev, resolution).cpu().numpy().astype(np.uint64)
  want = omorton.morton_codes(pts.numpy(), resolution)
  assert codes.shape == (n,) and np.array_equal(codes, want)
  order = morton_sort.argsort(dev, resolution).cpu().numpy()
  assert np.array_equal(order, omorton.argsort(pts.numpy(), resolution))
  keep = morton_sort.argsort_dedup(dev, resolution).cpu().numpy()
  assert np.array_equal(keep, omorton.argsort_dedup(pts.numpy(), resolution))
  assert torch.equal(morton_sort.sort(dev, resolution).cpu(), pts[torch.from_numpy(order).long()])
  assert torch.equal(morton_sort.sort_dedup(dev, resolution).cpu(), pts[torch.from_numpy(keep).long()])


def test_morton_nan_and_empty():
  from taichi_splatting_amd.misc import morton_sort
  assert morton_sort.argsort(torch.empty(0, 3, device='cuda:0'), 0.1).shape == (0,)
  pts = torch.tensor([[0.0, 0.0, 0.0], [1.0, 1.0, 1.0], [0.5, float('inf'), 0.5]], device='cuda:0')
  codes = morton_sort.morton_codes(pts, 0.25)
  assert codes.shape == (3,) and int(codes[0]) == 0 and int(codes[1]) == 0b111000000   # cell (4, 4, 4)


@pytest.mark.parametrize('n', [1, 63, 4096, 4097, 250_001])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('depth16', [False, True])
@pytest.mark.parametrize('ndc', [False, True])
def test_depth_argsort_equals_keys_then_stable_sort(n, dtype, depth16, ndc):
  """ms_depth_argsort (keys made inside the first radix pass) against ms_depth_sort_keys + a stable torch sort"""
  import ctypes
  from taichi_splatting_amd import _lib
  lib = _lib.load()
  torch.manual_seed(n)
  depth = (torch.rand(n, dtype=dtype) * (9.0 if ndc else 1.0) + (0.5 if ndc else 0.0)).to(DEV)
  depth[::7] = depth[0].clone()                           # ties: broken by index
  near, far = (0.25, 60.0) if ndc else (0.0, 0.0)
  stream = _lib.current_stream(torch.device(DEV))
  keys = torch.empty(n, dtype=torch.int32, device=DEV)
  vals = torch.empty(n, dtype=torch.int32, device=DEV)
  _lib.check(lib.ms_depth_sort_keys(depth.data_ptr(), n, int(depth16), near, far, keys.data_ptr(), vals.data_ptr(),
                                    _lib.dtype_code(dtype), stream), "keys")
  want_keys, want_order = torch.sort(keys.long() & 0xffffffff, stable=True)

  nb = ctypes.c_size_t(0)
  _lib.check(lib.ms_depth_argsort(None, n, int(depth16), near, far, _lib.dtype_code(dtype), None, None, None,
                                  ctypes.byref(nb), stream), "size")
  tmp = torch.empty(max(nb.value, 1), dtype=torch.uint8, device=DEV)
  got_keys = torch.empty(n, dtype=torch.int32, device=DEV)
  got_order = torch.empty(n, dtype=torch.int32, device=DEV)
  _lib.check(lib.ms_depth_argsort(depth.data_ptr(), n, int(depth16), near, far, _lib.dtype_code(dtype),
                                  got_keys.data_ptr(), got_order.data_ptr(), tmp.data_ptr(), ctypes.byref(nb), stream), "argsort")
  assert torch.equal(got_order.long(), want_order)
  assert torch.equal(got_keys.long() & 0xffffffff, want_keys)
