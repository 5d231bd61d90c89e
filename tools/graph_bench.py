"""eager vs HIP-graph replay of the fwd+bwd step: python tools/graph_bench.py n size [tile]"""
import sys, time; sys.path.insert(0, '.')
import torch
from taichi_splatting_amd import RasterConfig, frame, render_gaussians
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
n, size = int(sys.argv[1]), int(sys.argv[2]); tile = int(sys.argv[3]) if len(sys.argv) > 3 else 16
DEV='cuda:0'
torch.manual_seed(0)
cam = random_camera(image_size=(size, size))
g = random_3d_gaussians(n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.0)
g = g.replace(feature=(torch.rand(n, 3, 16) - 0.5) * 0.5).to(DEV); cam = cam.to(device=DEV)
cfg = RasterConfig(tile_size=tile, pixel_stride=(1, 1) if tile == 8 else (2, 2))
gd = g.requires_grad_(True)
leaves = [gd.position, gd.log_scaling, gd.rotation, gd.alpha_logit, gd.feature]
def step():
  for t in leaves: t.grad = None
  r = render_gaussians(gd, cam, cfg, use_sh=True)
  r.image.sum().backward()
  return r
def timeit(fn, k=100):
  for _ in range(10): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(k): fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e3
side = torch.cuda.Stream()
with torch.cuda.stream(side):
  e = timeit(step)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(100): step()
  host = (time.perf_counter() - t0) / 100 * 1e3     # enqueue time without the final sync
  torch.cuda.synchronize()
gr = frame.FrameGraph(step)
q = timeit(gr.replay)
print(f"n={n} size={size} tile={tile}: eager {e:.3f} ms (host enqueue {host:.3f}), graph replay {q:.3f} ms")
