"""Diagnostic: float32 projection kernel gradients vs the float64 oracle, error vs conditioning (eigen gap)."""
import sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from oracle import projection as oproj
from taichi_splatting_amd.perspective import projection as hip_proj
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians

DEV = 'cuda:0'
for seed in range(6):
  torch.manual_seed(seed)
  camera = random_camera()
  n = 5000
  g = random_3d_gaussians(n=n, camera_params=camera, margin=0.5, scale_factor=0.1 if seed % 2 else 1.0)
  inputs32 = [t.float() for t in g.shape_tensors()] + [camera.T_camera_world.float(), camera.projection.float()]

  def run(f, inputs):
    args = [a.detach().clone().requires_grad_(True) for a in inputs]
    out = f(*args)
    return out, args
  out_o, a_o = run(lambda *a: oproj.apply(*a, camera.image_size, camera.depth_range, blur_cov=0.3), [t.double() for t in inputs32])
  out_h, a_h = run(lambda *a: hip_proj.apply(*a, camera.image_size, camera.depth_range, blur_cov=0.3), [t.to(DEV) for t in inputs32])
  same = torch.equal(out_o[2], out_h[2].cpu())
  print(f"seed {seed}: V={out_o[2].shape[0]} same visible set: {same}")
  if not same:
    continue
  torch.manual_seed(100 + seed)
  Gp, Gd = torch.randn_like(out_o[0]), torch.randn_like(out_o[1])
  torch.autograd.backward([out_o[0], out_o[1]], [Gp, Gd])
  torch.autograd.backward([out_h[0], out_h[1]], [Gp.float().to(DEV), Gd.float().to(DEV)])
  # conditioning: relative eigen gap of the 2D covariance = (sx^2 - sy^2) / (sx^2 + sy^2)
  s = out_o[0][:, 4:6]
  gap = ((s[:, 0] ** 2 - s[:, 1] ** 2) / (s[:, 0] ** 2 + s[:, 1] ** 2)).abs()
  idx = out_o[2]
  fwd = (out_h[0].cpu().double() - out_o[0]).abs()
  print("  fwd max abs err per column", [f"{v:.1e}" for v in fwd.max(0).values.tolist()], "min gap", f"{gap.min():.2e}")
  for name, x, y in zip(('position', 'log_scaling', 'rotation', 'alpha_logit', 'T_camera_world', 'projection'), a_h, a_o):
    got, want = x.grad.cpu().double(), y.grad
    scale = want.abs().max().item()
    e = (got - want).abs()
    if e.dim() == 2 and e.shape[0] == n:
      ev = e[idx].max(dim=1).values / scale
      line = f"  {name}: max err/scale {ev.max():.2e}"
      for lo in (0.0, 1e-3, 1e-2, 1e-1):
        m = gap >= lo
        line += f" | gap>={lo:g}: {ev[m].max():.1e} ({int(m.sum())})"
      print(line)
    else:
      print(f"  {name}: max err/scale {e.max() / scale:.2e}")
  # the worst rotation-gradient row of this seed in detail
  got, want = a_h[2].grad.cpu().double(), a_o[2].grad
  e = (got - want).abs().max(dim=1).values
  wi = int(e.argmax())
  vis_pos = (out_o[2] == wi).nonzero()
  if vis_pos.numel():
    k = int(vis_pos[0])
    w_, h_ = camera.image_size
    print(f"  worst rotation row {wi}: err {e[wi]:.3e}, |grad| {want[wi].abs().max():.3e}, depth {out_o[1][k].item():.4f} (near {camera.near_plane}), "
          f"mean {out_o[0][k, :2].tolist()} image {w_}x{h_}, sigma {out_o[0][k, 4:6].tolist()}, alpha {out_o[0][k, 6].item():.3f}, "
          f"log_scaling {a_o[1][wi].tolist()}, f32-vs-f64 point row err {(out_h[0][k].cpu().double() - out_o[0][k]).abs().tolist()}")
    print(f"    grad f64 {want[wi].tolist()}\n    grad f32 {got[wi].tolist()}")
