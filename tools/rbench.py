#!/usr/bin/env python
"""Raster kernel bench harness for development variants (tools/build_variant.sh): times the product forward and
backward kernels through the C-ABI on one scene, compares their outputs with a reference file written by an earlier
run, and prints the -DMS_SCAN_STATS / -DMS_SCAN_PHASES counters when the loaded library has them.

    MS_SPLAT_LIB=tools/variants/lib<name>.so python tools/rbench.py [--scene D|dense|C|...] [--save ref.pt | --ref ref.pt]

One JSON line per run on stdout (prefix RBENCH), so that a shell loop over variants gives a table.
"""
import argparse
import ctypes
import json
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

FWD_PHASES = ['barrier_top', 'staging', 'barrier_staged', 'cull', 'hit_walk', 'visibility_flush', 'epilogue', 'before_first_batch']

PHASES = ['barrier_top', 'staging', 'barrier_staged', 'cull', 'chunk_prologue', 'blend', 'chunk_epilogue', 'commit',
          'before_first_batch', 'wave_total', 'chunks', 'waves']


def build_scene(args, dev):
  from taichi_splatting_amd import RasterConfig
  from taichi_splatting_amd.testing import random_camera, random_3d_gaussians, random_2d_gaussians
  from taichi_splatting_amd.perspective.projection import project_to_image
  from taichi_splatting_amd.spherical_harmonics import evaluate_sh_at
  from taichi_splatting_amd.mapper.tile_mapper import map_to_tiles
  from taichi_splatting_amd.rendering import ndc_depth
  from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
  cfg = RasterConfig(tile_size=args.tile, pixel_stride=(1, 1) if args.tile == 8 else (2, 2))
  torch.manual_seed(0)
  with torch.no_grad():
    if args.scene == 'dense':
      size = (1024, 768)
      g = random_2d_gaussians(1_000_000, size, num_channels=3, scale_factor=4.0, alpha_range=(0.75, 1.0),
                              depth_range=(0.1, 100.0)).to(dev)
      g2d, feats, depth = project_gaussians2d(g), g.feature.contiguous(), g.depths.contiguous()
      o2p, ranges = map_to_tiles(g2d, depth, size, cfg)
    else:
      n, side, scale, alpha = {'D': (6_000_000, 2048, 1.0, (0.1, 0.9)), 'C': (1_000_000, 1080, 1.0, (0.1, 0.9)),
                               'E': (6_000_000, 4096, 1.0, (0.1, 0.9)), 'big': (1_500_000, 2048, 2.0, (0.1, 0.9)),
                               'small': (6_000_000, 2048, 0.5, (0.1, 0.9))}[args.scene]
      n = args.n or n
      size = (1920, 1080) if args.scene == 'C' else (side, side)
      cam = random_camera(image_size=size)
      g = random_3d_gaussians(n, cam, scale_factor=scale, alpha_range=alpha, margin=0.0)
      g = g.replace(feature=(torch.rand(n, 3, 16) - 0.5) * 0.5).to(dev)
      cam = cam.to(device=dev)
      g2d, depths, idx = project_to_image(g, cam, cfg)
      feats = evaluate_sh_at(g.feature, g.position, idx, cam.camera_position)
      o2p, ranges = map_to_tiles(g2d, ndc_depth(depths, cam.near_plane, cam.far_plane), size, cfg)
  return cfg, size, g2d.contiguous(), feats.contiguous(), o2p, ranges.view(-1, 2).contiguous()


def main():
  p = argparse.ArgumentParser()
  p.add_argument('--scene', default='D')
  p.add_argument('--n', type=int, default=0)
  p.add_argument('--tile', type=int, default=16)
  p.add_argument('--iters', type=int, default=20)
  p.add_argument('--save', default='')
  p.add_argument('--ref', default='')
  p.add_argument('--tag', default='')
  p.add_argument('--heur', action='store_true', help='backward with the point heuristics (the training configuration)')
  p.add_argument('--vis', action='store_true', help='also time the forward with the per-splat visibility sums')
  p.add_argument('--rows', action='store_true', help='gather from a splat-row table (ms_splat_rows_pack, ms_raster_*_rows)')
  args = p.parse_args()

  from taichi_splatting_amd import _lib
  dev = torch.device('cuda', 0)
  lib = _lib.load()
  cfg, (w, h), g2d, feats, o2p, ranges2 = build_scene(args, dev)
  n, k = g2d.shape[0], o2p.shape[0]
  if args.heur:
    from dataclasses import replace as _replace
    cfg = _replace(cfg, compute_point_heuristic=True)
  cfg_c = _lib.raster_config_c(cfg)
  stream = _lib.current_stream(dev)
  th = (h + cfg.tile_size - 1) // cfg.tile_size
  image = torch.empty((h, w, 3), device=dev)
  alpha = torch.empty((h, w), device=dev)

  rows = None
  if args.rows:
    rows = torch.zeros((n, _lib.SPLAT_ROW), device=dev)
    _lib.check(lib.ms_splat_rows_pack(g2d.data_ptr(), None, feats.data_ptr(), n, rows.data_ptr(), stream), "pack")

  def fwd():
    if rows is not None:
      return _lib.check(lib.ms_raster_fwd_rows(rows.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), w, h, cfg_c,
                                               image.data_ptr(), alpha.data_ptr(), None, 0, th, stream), "fwd rows")
    _lib.check(lib.ms_raster_fwd(g2d.data_ptr(), feats.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), w, h, 3, cfg_c,
                                 image.data_ptr(), alpha.data_ptr(), None, 0, th, _lib.dtype_code(torch.float32), stream),
               "fwd")
  fwd()
  out_fwd = {}
  fwd_phase_fn = getattr(lib, 'ms_debug_fwd_phases', None)
  if fwd_phase_fn is not None:
    # -DMS_FWD_PHASES build: where a forward wave's cycles go (one row of 12 counters per wave)
    fwd_phase_fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    blocks = (ranges2.shape[0] + 63) // 64 * 64                       # xcd_grid pads the launch to a multiple of 64 tiles
    rows_f = torch.zeros((blocks * (cfg.tile_size * cfg.tile_size // 64), 12), dtype=torch.int64, device=dev)
    fwd_phase_fn(ctypes.c_void_p(rows_f.data_ptr()), 0)
    fwd()
    torch.cuda.synchronize()
    fwd_phase_fn(None, 0)
    v = [int(x) for x in rows_f.sum(dim=0).tolist()]
    total = max(v[8], 1)
    out_fwd["fwd_phases_frac"] = {FWD_PHASES[i]: round(v[i] / total, 4) for i in range(8)}
    out_fwd["fwd_phases_frac"]["unaccounted"] = round(1.0 - sum(v[:8]) / total, 4)
    out_fwd["fwd_cycles_per_wave"] = round(total / max(v[11], 1))
    out_fwd["fwd_hits"] = v[9]
    out_fwd["fwd_batches"] = v[10]
    out_fwd["fwd_cycles_per_hit_in_walk"] = round(v[4] / max(v[9], 1), 1)
    out_fwd["fwd_cull_cycles_per_batch"] = round(v[3] / max(v[10], 1), 1)
    out_fwd["fwd_staging_cycles_per_batch"] = round(v[1] / max(v[10], 1), 1)
    out_fwd["fwd_barrier_cycles_per_batch"] = round((v[0] + v[2]) / max(v[10], 1), 1)
    busy = rows_f[rows_f[:, 11] > 0]
    out_fwd["fwd_wave_cycles_quantiles"] = [int(q) for q in busy[:, 8].double().quantile(torch.tensor([0.05, 0.5, 0.95, 1.0], dtype=torch.float64, device=dev)).tolist()]
  torch.manual_seed(1)
  grad_image = torch.rand_like(image) + 0.5
  mom = torch.zeros((n, _lib.MOMENT_ROW), device=dev)

  def bwd():
    if rows is not None and args.tile != 8:
      return _lib.check(lib.ms_raster_bwd_moments_rows(rows.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), image.data_ptr(),
                                                       grad_image.data_ptr(), w, h, cfg_c, mom.data_ptr(), 0, None, 0, th,
                                                       stream), "bwd rows")
    _lib.check(lib.ms_raster_bwd_moments(g2d.data_ptr(), feats.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(),
                                         image.data_ptr(), grad_image.data_ptr(), w, h, cfg_c, mom.data_ptr(), 0, None, 0,
                                         th, stream), "bwd")

  def counters(name, count):
    fn = getattr(lib, name, None)
    if fn is None:
      return None, None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    buf = (ctypes.c_ulonglong * count)()
    return fn, buf
  stats_fn, stats = counters('ms_debug_scan_stats', 12)
  phase_fn, _ = counters('ms_debug_scan_phases', 1)
  if stats_fn is not None:
    stats_fn(None, 1)
  if phase_fn is not None:
    waves = ranges2.shape[0] * (cfg.tile_size * cfg.tile_size // 64)
    phase_rows = torch.zeros((waves, 12), dtype=torch.int64, device=dev)
    phase_fn(ctypes.c_void_p(phase_rows.data_ptr()), 0)
  bwd()
  torch.cuda.synchronize()
  out = {"tag": args.tag or os.path.basename(os.environ.get('MS_SPLAT_LIB', 'default')), "scene": args.scene,
         "tile": args.tile, "V": n, "K": k}
  out.update(out_fwd)
  if stats_fn is not None:
    stats_fn(ctypes.cast(stats, ctypes.c_void_p), 1)
    v = [int(x) for x in stats]
    out["stats"] = {"passes": v[0], "sub_hits": v[1], "patch_hits": v[6], "chunks": v[2], "fill": round(v[3] / max(v[2], 1), 2),
                    "steps": v[4], "pairs": v[5], "pairs_per_step": round(v[5] / max(v[4], 1), 2),
                    "balance": round(v[8] / max(v[9], 1), 3)}
  if phase_fn is not None:
    phase_fn(None, 0)
    v = [int(x) for x in phase_rows.sum(dim=0).tolist()]
    busy = phase_rows[phase_rows[:, 11] > 0]
    out["wave_cycles_quantiles"] = [int(q) for q in busy[:, 9].double().quantile(torch.tensor([0.05, 0.5, 0.95, 1.0], dtype=torch.float64, device=dev)).tolist()]
    total = max(v[9], 1)
    out["phases_frac"] = {PHASES[i]: round(v[i] / total, 4) for i in range(9)}
    out["phases_frac"]["unaccounted"] = round(1.0 - sum(v[:9]) / total, 4)
    out["cycles_per_wave"] = round(total / max(v[11], 1))
    out["chunks"] = v[10]
    out["cycles_per_chunk"] = {PHASES[i]: round(v[i] / max(v[10], 1), 1) for i in (4, 5, 6)}

  # outputs against a reference run
  mom_out = mom.clone()
  if args.save:
    torch.save({"image": image.cpu(), "moments": mom_out.cpu()}, args.save)
  if args.ref and os.path.exists(args.ref):
    ref = torch.load(args.ref)
    di = (image.cpu() - ref["image"]).abs().max().item()
    rm = ref["moments"]
    dm = (mom_out.cpu() - rm).abs().max(dim=0).values / rm.abs().max(dim=0).values.clamp_min(1e-30)
    out["vs_ref"] = {"image_max_abs": di, "moments_max_rel_of_col_max": round(dm.max().item(), 8)}

  def time_ms(fn, pre=None):
    for _ in range(3):
      if pre: pre()
      fn()
    torch.cuda.synchronize()
    tot = []
    for _ in range(args.iters):
      if pre: pre()
      s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      s.record(); fn(); e.record()
      torch.cuda.synchronize()
      tot.append(s.elapsed_time(e))
    tot.sort()
    return round(sum(tot) / len(tot), 4), round(tot[len(tot) // 2], 4), round(tot[0], 4)
  out["fwd_ms_mean_med_min"] = time_ms(fwd)
  out["bwd_ms_mean_med_min"] = time_ms(bwd, pre=lambda: mom.zero_())
  if args.vis:
    from dataclasses import replace
    cfg_v = _lib.raster_config_c(replace(cfg, compute_visibility=True))
    vis = torch.zeros((n,), device=dev)

    def fwd_vis():
      _lib.check(lib.ms_raster_fwd(g2d.data_ptr(), feats.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), w, h, 3, cfg_v,
                                   image.data_ptr(), alpha.data_ptr(), vis.data_ptr(), 0, th, _lib.dtype_code(torch.float32),
                                   stream), "fwd vis")
    out["fwd_vis_ms_mean_med_min"] = time_ms(fwd_vis, pre=lambda: vis.zero_())
    out["vis_sum"] = round(float(vis.double().sum()), 3)
  print("RBENCH " + json.dumps(out), flush=True)


if __name__ == '__main__':
  main()
