"""Run-length distribution (overlaps per tile) of the bench scene: what the size classes of csrc/tile_sort.hip see.
  python tools/diag/run_lengths.py [bench.py arguments]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
from taichi_splatting_amd import RasterConfig, frame, render_gaussians

args = bench.parse_args()
g, cam = bench.make_scene(args, 'cuda:0')
cfg = RasterConfig(tile_size=args.tile_size) if hasattr(args, 'tile_size') else RasterConfig()
with torch.no_grad():
  r = render_gaussians(g, cam, cfg, use_sh=True)
ranges = r.frame.tile_ranges()
n = (ranges[..., 1] - ranges[..., 0]).flatten().float()
qs = torch.tensor([0.0, 0.01, 0.1, 0.5, 0.9, 0.99, 0.999, 1.0], device=n.device)
print("tiles", n.numel(), "overlaps", int(n.sum()), "mean", float(n.mean()))
print("quantiles", [int(v) for v in torch.quantile(n, qs)])
for cap in (1024, 1280, 1536, 2048, 5120):
  print(f"runs > {cap}: {int((n > cap).sum())}")
