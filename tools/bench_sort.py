"""Stand-alone timing of ms_radix_sort_pairs on the two sorts of a config D frame (development tool):
the K = 12.76 M (tile id, point) pairs on 14 bits and the V = 6 M (depth, point) pairs on 32 bits."""
import sys, time, torch, ctypes
sys.path.insert(0, '.')
from taichi_splatting_amd import _lib
lib = _lib.load()
dev = torch.device('cuda', 0)
CASES = ((12_760_302, 14), (6_000_000, 32))
if len(sys.argv) > 1:                      # python tools/bench_sort.py k | v
  CASES = CASES[:1] if sys.argv[1] == 'k' else CASES[1:]
for n, bits in CASES:
  torch.manual_seed(0)
  keys = torch.randint(0, 2 ** min(bits, 31) - 1, (n,), dtype=torch.int32, device=dev)
  vals = torch.arange(n, dtype=torch.int32, device=dev)
  ko, vo = torch.empty_like(keys), torch.empty_like(vals)
  nb = ctypes.c_size_t(0)
  lib.ms_radix_sort_pairs(None, None, None, None, n, 4, 0, bits, None, ctypes.byref(nb), None)
  tmp = torch.empty(nb.value, dtype=torch.uint8, device=dev)
  st = _lib.current_stream(dev)
  def run():
    _lib.check(lib.ms_radix_sort_pairs(keys.data_ptr(), vals.data_ptr(), ko.data_ptr(), vo.data_ptr(), n, 4, 0, bits, tmp.data_ptr(), ctypes.byref(nb), st), "sort")
  for _ in range(5): run()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(50): run()
  torch.cuda.synchronize()
  print(f"n={n} bits={bits}: {(time.perf_counter() - t0) / 50 * 1e3:.4f} ms")
