for i in 1 2 3; do python bench.py --n 2000 --size 256 --no-cpu-baseline --no-stages --no-sweep --no-graph --steps 300 2>&1 | grep -o '"ms_per_step": [0-9.]*'; done
python - <<'PY'
import torch, time, cProfile, pstats, io
from taichi_splatting_amd import RasterConfig, render_gaussians
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
torch.manual_seed(0)
cam = random_camera(image_size=(256, 256))
g = random_3d_gaussians(2000, cam, scale_factor=1.0, alpha_range=(0.1, 0.9))
g = g.replace(feature=(torch.rand(2000, 3, 16) - 0.5) * 0.5).to('cuda:0')
cam = cam.to(device='cuda:0')
g.requires_grad_(True)
cfg = RasterConfig()
def step():
  render_gaussians(g, cam, cfg, use_sh=True).image.sum().backward()
for _ in range(50): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 300
pr.disable()
print("host-bound frame: %.3f ms" % (dt * 1e3))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(35); print(s.getvalue()[:6000])
PY
