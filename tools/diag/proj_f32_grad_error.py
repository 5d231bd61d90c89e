"""Diagnostic: float32 projection gradients — the HIP kernel and the oracle evaluated in float32 (torch CPU), both
against the float64 oracle: distribution of the per-row error relative to the largest gradient."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from oracle import projection as oproj
from taichi_splatting_amd.perspective import projection as hip_proj
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians

DEV = 'cuda:0'
for seed in range(8):
  torch.manual_seed(seed)
  camera = random_camera()
  n = 4000
  g = random_3d_gaussians(n=n, camera_params=camera, margin=0.5, scale_factor=0.1 if seed % 2 else 1.0)
  inputs32 = [t.float() for t in g.shape_tensors()] + [camera.T_camera_world.float(), camera.projection.float()]

  def run(f, args, gp=None, gd=None):
    args = [a.detach().clone().requires_grad_(True) for a in args]
    points, depth, idx = f(*args)
    if gp is None:
      torch.manual_seed(100 + seed)
      gp, gd = torch.randn(points.shape, dtype=torch.float64), torch.randn(depth.shape, dtype=torch.float64)
    torch.autograd.backward([points, depth], [gp.to(points), gd.to(depth)])
    return (points.detach(), depth.detach(), idx), [a.grad for a in args], gp, gd
  f_o = lambda *a: oproj.apply(*a, camera.image_size, camera.depth_range, blur_cov=0.3)
  f_h = lambda *a: hip_proj.apply(*a, camera.image_size, camera.depth_range, blur_cov=0.3)
  o64, g64, gp, gd = run(f_o, [t.double() for t in inputs32])
  o32, g32, _, _ = run(f_o, inputs32, gp, gd)
  oh, gh, _, _ = run(f_h, [t.to(DEV) for t in inputs32], gp.to(DEV), gd.to(DEV))
  print(f"seed {seed}: V={o64[2].shape[0]} sets equal: f32 oracle {torch.equal(o32[2], o64[2])} hip {torch.equal(oh[2].cpu(), o64[2])}")
  if not (torch.equal(o32[2], o64[2]) and torch.equal(oh[2].cpu(), o64[2])):
    continue
  idx = o64[2]
  for name, a, b, c in zip(('position', 'log_scaling', 'rotation', 'alpha_logit', 'T_camera_world', 'projection'), gh, g32, g64):
    a, b = a.cpu().double(), b.double()
    scale = c.abs().max().item()
    if a.dim() == 2 and a.shape[0] == n:
      eh = (a - c).abs().max(dim=1).values[idx] / scale
      eo = (b - c).abs().max(dim=1).values[idx] / scale
      q = torch.tensor([0.5, 0.9, 0.99, 0.999, 1.0], dtype=torch.float64)
      fmt = lambda e: ' '.join(f"{v:.1e}" for v in torch.quantile(e, q).tolist())
      print(f"  {name:12s} hip  q50/90/99/99.9/max {fmt(eh)}  frac<=1e-4 {(eh <= 1e-4).float().mean():.4f}")
      print(f"  {'':12s} f32o q50/90/99/99.9/max {fmt(eo)}  frac<=1e-4 {(eo <= 1e-4).float().mean():.4f}")
    else:
      print(f"  {name:12s} hip {(a - c).abs().max() / scale:.2e}   f32 oracle {(b - c).abs().max() / scale:.2e}")
