import sys; sys.path.insert(0, '.')
import torch
from taichi_splatting_amd import RasterConfig, frame, render_gaussians
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
DEV='cuda:0'
mode = sys.argv[1]
torch.manual_seed(7)
cam = random_camera(image_size=(256,256))
g = random_3d_gaussians(20000, cam, scale_factor=1.0, alpha_range=(0.1, 0.9), margin=0.1)
g = g.replace(feature=(torch.rand(20000, 3, 16) - 0.5) * 0.5).to(DEV); cam = cam.to(device=DEV)
cfg = RasterConfig()
gd = g.clone().requires_grad_(mode != 'fwd')
leaves = [gd.position, gd.log_scaling, gd.rotation, gd.alpha_logit, gd.feature]
def step():
  for t in leaves: t.grad = None
  r = render_gaussians(gd, cam, cfg, use_sh=True)
  if mode == 'full':
    r.image.sum().backward()
  return r
if mode == 'fwd':
  with torch.no_grad():
    gr = frame.FrameGraph(step)
    r = gr.replay()
else:
  gr = frame.FrameGraph(step)
  r = gr.replay()
torch.cuda.synchronize()
print(mode, 'ok', float(r.image.sum()))
