"""-m gpu: scan / radix sort / ranges against torch (bit-exact integer work)."""
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
