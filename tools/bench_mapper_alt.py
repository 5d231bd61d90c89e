"""Feasibility timing for a mapper WITHOUT the depth pre-sort (development tool, round 4): the pieces that exist already —
(a) the overlap count in gaussian order (streaming, no gather), (b) the tile sort of K pairs with 64-bit keys
tile << 32 | depth bits over the tile bits only (12-byte items), against today's 8-byte items."""
import sys, time, torch, ctypes
sys.path.insert(0, '.')
from taichi_splatting_amd import _lib
lib = _lib.load()
dev = torch.device('cuda', 0)
st = _lib.current_stream(dev)

def timed(fn, n=50):
  for _ in range(5): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t0) / n * 1e3

K = 12_760_302
torch.manual_seed(0)
tile = torch.randint(0, 16384, (K,), dtype=torch.int64, device=dev)
depth = torch.randint(0, 2 ** 31 - 1, (K,), dtype=torch.int64, device=dev)
vals = torch.arange(K, dtype=torch.int32, device=dev)
for key_bytes, keys, b0, b1 in ((4, tile.to(torch.int32), 0, 14), (8, (tile << 32) | depth, 32, 46)):
  ko, vo = torch.empty_like(keys), torch.empty_like(vals)
  nb = ctypes.c_size_t(0)
  lib.ms_radix_sort_pairs(None, None, None, None, K, key_bytes, b0, b1, None, ctypes.byref(nb), None)
  tmp = torch.empty(nb.value, dtype=torch.uint8, device=dev)
  run = lambda: _lib.check(lib.ms_radix_sort_pairs(keys.data_ptr(), vals.data_ptr(), ko.data_ptr(), vo.data_ptr(), K, key_bytes, b0, b1,
                                                   tmp.data_ptr(), ctypes.byref(nb), st), "sort")
  print(f"tile sort, {key_bytes}-byte keys, bits [{b0}, {b1}): {timed(run):.4f} ms")

# overlap count: gaussian order (no gather) vs through a random order with the ordered copy
import bench, argparse
from taichi_splatting_amd import RasterConfig
from taichi_splatting_amd.perspective.projection import project_to_image
args = argparse.Namespace(n=6_000_000, size=2048, height=None, tile=16, sh_degree=0, seed=0)
g, cam = bench.make_scene(args, dev)
cfg = RasterConfig()
with torch.no_grad():
  p, d, idx = project_to_image(g, cam, cfg)
n = p.shape[0]
counts = torch.empty(n, dtype=torch.int32, device=dev)
order = torch.randperm(n, device=dev).to(torch.int32)
ordered = torch.empty_like(p)
f = lambda o, oc: _lib.check(lib.ms_tile_count(p.data_ptr(), o, n, 2048, 2048, 16, ctypes.c_float(cfg.alpha_threshold), 0, 128,
                                               counts.data_ptr(), oc, st), "count")
print(f"tile_count in gaussian order (streaming): {timed(lambda: f(None, None)):.4f} ms")
print(f"tile_count through a random order + ordered copy: {timed(lambda: f(order.data_ptr(), ordered.data_ptr())):.4f} ms")
