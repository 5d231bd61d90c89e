#!/usr/bin/env python
"""Scene-statistics sweep: every tuning constant of the kernels was fitted to
``random_3d_gaussians(scale_factor=1)`` at 6 M gaussians; this runs the stages over a family of scene shapes and reports,
per scene, the overlap statistics, the time per stage, the cost PER OVERLAP of the three stages that scale with overlaps
(tile mapper, raster forward, raster backward) and which mapper sequence the frame executor chose — and flags every stage
whose cost per overlap exceeds GUARD x config D's.

  scenes   random_3d_gaussians at scale_factor 0.5 / 1 / 2 / 4 / 8  x  opacity 0.1-0.9 / 0.75-1.0  x  1024^2 / 2048^2 /
           4096^2 (the reference's three resolutions, BENCHMARK.md:36-40);  a heavy-tailed mix (5 % of the splats at
           20 x scale);  a pile-up (every splat centred inside one 48 x 48 px window: tile runs of ~100 000 entries,
           which the raster kernels cut into segments: ms_raster_fwd_split / ms_raster_bwd_moments_split);
           the reference's dense 2D component shape (benchmarks/bench_rasterizer.py:21-26);
           round 6 — what trained scenes look like and the bench scene does not: HALF-CULLED
           scenes (random_3d_gaussians(margin=0.5): 56 % of the gaussians outside the image, V ~ N / 2), NEEDLE scenes (one
           log-scale axis + ln 10: strongly anisotropic splats, the oriented-box cull levels), and a ZOOM path whose
           overlaps per gaussian cross the direct / pre-sort crossover every few frames.
  first    the FIRST frame of every shape (shape caches cold: capacity unknown, mapper sequence unknown, no long-run
           record; allocator warm) beside the steady frame, guarded at FIRST_GUARD x

    python tools/sweep_scenes.py [--quick] [--out profiles/r05_scene_sweep.txt]
"""
import argparse
import ctypes
import json
import math
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

GUARD = 2.0
FIRST_GUARD = 3.0


def cuda_ms(fn, iters=5, warmup=2):
  """median of ``iters`` individually timed calls (one call that hits the caching allocator's slow path — a 20 ms
  hipMalloc in the first version of this table — does not become the scene's number)"""
  for _ in range(warmup):
    fn()
  torch.cuda.synchronize()
  times = []
  for _ in range(iters):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn()
    b.record()
    torch.cuda.synchronize()
    times.append(a.elapsed_time(b))
  times.sort()
  return times[len(times) // 2]


def make_scene(spec, dev):
  from taichi_splatting_amd.testing import random_camera, random_3d_gaussians, random_2d_gaussians
  torch.manual_seed(spec.get('seed', 0))
  if spec['kind'] == 'dense2d':
    size = (1024, 768)
    g = random_2d_gaussians(1_000_000, size, num_channels=3, scale_factor=4.0, alpha_range=(0.75, 1.0), depth_range=(0.1, 100.0))
    return g.to(dev), None, size
  size = (spec['side'], spec['side'])
  cam = random_camera(image_size=size)
  n = spec['n']
  g = random_3d_gaussians(n, cam, scale_factor=spec['scale'], alpha_range=spec['alpha'], margin=spec.get('margin', 0.0))
  if spec['kind'] == 'needle':
    # one axis ten times longer than the generator's: splats of aspect ~10 at random orientations
    stretch = torch.zeros(n, 3)
    stretch[torch.arange(n), torch.randint(0, 3, (n,))] = math.log(10.0)
    g = g.replace(log_scaling=g.log_scaling + stretch)
  if spec['kind'] == 'heavy_tail':
    big = torch.rand(n) < 0.05
    g = g.replace(log_scaling=g.log_scaling + math.log(20.0) * big[:, None].float())
  if spec['kind'] == 'pile':
    # the generator's own construction with the pixel positions drawn inside one 48 x 48 pixel window: same footprints
    # (sigma ~ w / sqrt(n) px at the splat's own depth), all of them on the same nine tiles
    from taichi_splatting_amd.testing.random_data import _lift_to_world
    from taichi_splatting_amd.rendering import inverse_ndc_depth
    px = torch.rand(n, 2) * 48.0 + torch.tensor([[size[0] * 0.5, size[1] * 0.5]])
    z = inverse_ndc_depth(torch.rand(n), cam.near_plane * 2, cam.far_plane)
    focal = cam.T_image_camera[0, 0]
    world_size = (size[0] / math.sqrt(n)) * (z / focal)
    g = g.replace(position=_lift_to_world(px, z.unsqueeze(1), cam.T_image_world),
                  log_scaling=torch.randn(n, 3) * 0.5 + torch.log(world_size).unsqueeze(1))
  g = g.replace(feature=(torch.rand(n, 3, 16) - 0.5) * 0.5)
  return g.to(dev), cam.to(device=dev), size


def measure(spec, dev):
  from taichi_splatting_amd import RasterConfig, _lib, frame, render_gaussians
  from taichi_splatting_amd.perspective.projection import project_to_image
  from taichi_splatting_amd.spherical_harmonics import evaluate_sh_at
  from taichi_splatting_amd.mapper.tile_mapper import map_to_tiles
  from taichi_splatting_amd.rendering import ndc_depth
  from taichi_splatting_amd.misc.renderer2d import project_gaussians2d
  lib = _lib.load()
  cfg = RasterConfig(tile_size=16)
  g, cam, size = make_scene(spec, dev)
  out = dict(name=spec['name'])
  w, h = size
  with torch.no_grad():
    if cam is None:
      g2d, feats, depth = project_gaussians2d(g), g.feature.contiguous(), g.depths.contiguous()
      out['project_ms'] = out['sh_ms'] = 0.0
    else:
      out['project_ms'] = cuda_ms(lambda: project_to_image(g, cam, cfg))
      g2d, depths, idx = project_to_image(g, cam, cfg)
      out['sh_ms'] = cuda_ms(lambda: evaluate_sh_at(g.feature, g.position, idx, cam.camera_position))
      feats = evaluate_sh_at(g.feature, g.position, idx, cam.camera_position)
      depth = ndc_depth(depths, cam.near_plane, cam.far_plane)
    per_method = {}
    for method in ('direct', 'presort'):
      per_method[method] = cuda_ms(lambda: map_to_tiles(g2d, depth, size, cfg, method=method), iters=3, warmup=1)
    out['map_direct_ms'], out['map_presort_ms'] = per_method['direct'], per_method['presort']
    out['map_auto_ms'] = cuda_ms(lambda: map_to_tiles(g2d, depth, size, cfg), iters=3, warmup=1)
    o2p, ranges = map_to_tiles(g2d, depth, size, cfg)
    o2p_d, ranges_d = map_to_tiles(g2d, depth, size, cfg, method='direct')
    o2p_p, ranges_p = map_to_tiles(g2d, depth, size, cfg, method='presort')
    out['lists_equal'] = bool(torch.equal(o2p_d, o2p_p) and torch.equal(ranges_d, ranges_p))
    del o2p_d, ranges_d, o2p_p, ranges_p
    ranges2 = ranges.view(-1, 2)
    runs = (ranges2[:, 1] - ranges2[:, 0])
    v, k = g2d.shape[0], o2p.shape[0]
    out.update(N=int(g.position.shape[0]), V=v, K=k, K_per_N=k / max(v, 1), tiles=int(ranges2.shape[0]),
               K_per_tile_mean=k / ranges2.shape[0], K_per_tile_max=int(runs.max()))
    cfg_c = _lib.raster_config_c(cfg)
    stream = _lib.current_stream(dev)
    th = (h + 15) // 16
    image = torch.empty((h, w, 3), device=dev)
    alpha = torch.empty((h, w), device=dev)
    g2d_c, feats_c = g2d.contiguous(), feats.contiguous()

    def fwd():
      _lib.check(lib.ms_raster_fwd(g2d_c.data_ptr(), feats_c.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), w, h, 3, cfg_c,
                                   image.data_ptr(), alpha.data_ptr(), None, 0, th, _lib.dtype_code(torch.float32), stream), "fwd")
    out['raster_fwd_ms'] = cuda_ms(fwd)
    grad_image = torch.ones_like(image)
    mom = torch.zeros((v, _lib.MOMENT_ROW), device=dev)
    # a shape with a run above 16 384 entries is rasterized in segments from its second frame on (frame.py sets
    # ms_frame_desc.split_long_runs): those launches are what the guard judges, the per-tile launches are reported
    long_runs = int(runs.max()) > 16384
    out['raster'] = 'segments' if long_runs else 'per tile'
    if long_runs:
      scratch = torch.empty((lib.ms_raster_split_scratch_bytes(k, 16, 0, 0),), dtype=torch.uint8, device=dev)

      def fwd_split():
        _lib.check(lib.ms_raster_fwd_split(g2d_c.data_ptr(), feats_c.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), k, w, h,
                                           cfg_c, image.data_ptr(), alpha.data_ptr(), None, scratch.data_ptr(), 0, 0, 0, th, stream), "fwd split")

      def bwd_split():
        _lib.check(lib.ms_raster_bwd_moments_split(g2d_c.data_ptr(), feats_c.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(), k,
                                                   image.data_ptr(), grad_image.data_ptr(), w, h, cfg_c, mom.data_ptr(), 0, None,
                                                   scratch.data_ptr(), 0, 0, 0, th, stream), "bwd split")
      out['raster_fwd_per_tile_ms'] = out['raster_fwd_ms']
      out['raster_fwd_ms'] = cuda_ms(fwd_split)

    def bwd():
      _lib.check(lib.ms_raster_bwd_moments(g2d_c.data_ptr(), feats_c.data_ptr(), ranges2.data_ptr(), o2p.data_ptr(),
                                           image.data_ptr(), grad_image.data_ptr(), w, h, cfg_c, mom.data_ptr(), 0, None, 0,
                                           th, stream), "bwd")
    out['raster_bwd_ms'] = cuda_ms(bwd)
    if long_runs:
      out['raster_bwd_per_tile_ms'] = out['raster_bwd_ms']
      fwd_split()
      out['raster_bwd_ms'] = cuda_ms(bwd_split)
      del scratch
    del mom, image, alpha, grad_image
  # the frame: forward + backward through the executor, and the mapper sequence it settles on
  if cam is not None:
    g.requires_grad_(True)
    leaves = (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)

    def step():
      for t in leaves:
        t.grad = None
      render_gaussians(g, cam, cfg, use_sh=True).image.sum().backward()
    first = time.perf_counter()
    step(); torch.cuda.synchronize()
    out['first_frame_wall_ms'] = (time.perf_counter() - first) * 1e3       # includes the allocator's hipMallocs
    out['frame_ms'] = cuda_ms(step, iters=5, warmup=3)
    # the first frame of the shape again, with the executor's shape caches dropped and the allocator warm: what the
    # POLICY costs (capacity unknown: wait for K, then map; mapper sequence unknown; no long-run record)
    frame.release_caches()
    torch.cuda.synchronize()
    first = time.perf_counter()
    step(); torch.cuda.synchronize()
    out['first_frame_ms'] = (time.perf_counter() - first) * 1e3
    for _ in range(3):
      step()
    torch.cuda.synchronize()
    key = frame._shape_key(dev, g.position.shape[0], size, cfg, None, False)
    out['frame_mapper'] = {_lib.MAPPER_DIRECT: 'direct', _lib.MAPPER_PRESORT: 'presort'}.get(frame._mapper_mode.get(key), '?')
    g.requires_grad_(False)
    frame.release_caches()
  # the mapper's cost is taken from the sequence the frame executor settles on for this scene shape (the modular
  # map_to_tiles(method=None) counts twice when it picks the pre-sort and cannot see giant tile runs: it is reported, not judged)
  settled = out.get('frame_mapper', 'direct' if out['map_direct_ms'] <= out['map_presort_ms'] else 'presort')
  out['map_ms'] = out['map_presort_ms'] if settled == 'presort' else out['map_direct_ms']
  out['map_best_ms'] = min(out['map_direct_ms'], out['map_presort_ms'])
  for stage, ms in (('map', out['map_ms']), ('fwd', out['raster_fwd_ms']), ('bwd', out['raster_bwd_ms'])):
    out[f'{stage}_ps_per_overlap'] = ms * 1e9 / max(k, 1)         # picoseconds
  return out


def measure_zoom(dev):
  """One scene shape whose footprints alternate between two scales every three frames: overlaps per gaussian 3.0 <-> 4.4,
  either side of the direct / pre-sort crossover (frame.PRESORT_ABOVE / DIRECT_BELOW).  The executor picks the sequence
  from the PREVIOUS frame of the shape, so every switch is mapped once with the other sequence: reported per frame."""
  from taichi_splatting_amd import RasterConfig, _lib, frame, render_gaussians
  from taichi_splatting_amd.testing import random_camera, random_3d_gaussians
  cfg = RasterConfig(tile_size=16)
  n, side = 3_000_000, 2048
  torch.manual_seed(0)
  cam = random_camera(image_size=(side, side))
  base = random_3d_gaussians(n, cam, scale_factor=1.0, alpha_range=(0.1, 0.9))
  base = base.replace(feature=(torch.rand(n, 3, 16) - 0.5) * 0.5).to(dev)
  cam = cam.to(device=dev)
  variants = [base.replace(log_scaling=base.log_scaling + math.log(s)) for s in (1.25, 1.6)]
  for v in variants:
    v.requires_grad_(True)
  frame.release_caches()
  key = frame._shape_key(dev, n, (side, side), cfg, None, False)
  rows = []
  steady = []
  for v in variants:                                   # each scale on its own, settled
    def step(v=v):
      for t in (v.position, v.log_scaling, v.rotation, v.alpha_logit, v.feature):
        t.grad = None
      return render_gaussians(v, cam, cfg, use_sh=True)
    frame.release_caches()
    for _ in range(4):
      r = step(); r.image.sum().backward()
    steady.append(cuda_ms(lambda: step().image.sum().backward(), iters=5, warmup=1))
    rows.append(dict(K=frame.frame_status(r)['overlaps'], mapper=frame._mapper_mode.get(key)))
  frame.release_caches()
  times, modes = [], []
  for i in range(18):
    v = variants[(i // 3) % 2]
    for t in (v.position, v.log_scaling, v.rotation, v.alpha_logit, v.feature):
      t.grad = None
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    r = render_gaussians(v, cam, cfg, use_sh=True)
    modes.append(int(r.frame.desc.mapper))
    r.image.sum().backward()
    b.record()
    torch.cuda.synchronize()
    times.append(a.elapsed_time(b))
  frame.release_caches()
  expect = [steady[(i // 3) % 2] for i in range(18)]
  worst = max(t / e for t, e in zip(times[3:], expect[3:]))
  return dict(name='zoom_3M_2048', K_per_N=[round(r['K'] / n, 2) for r in rows],
              steady_ms=[round(x, 3) for x in steady], settled_mapper=[r['mapper'] for r in rows],
              frame_ms=[round(t, 3) for t in times], frame_mapper=modes,
              mean_over_steady=round(sum(times[3:]) / sum(expect[3:]), 3), worst_over_steady=round(worst, 3))


def scenes(quick):
  specs = [dict(name='configD', kind='plain', n=6_000_000, side=2048, scale=1.0, alpha=(0.1, 0.9))]
  counts = {0.5: 6_000_000, 1.0: 6_000_000, 2.0: 6_000_000, 4.0: 2_000_000, 8.0: 1_000_000}
  sides = (2048,) if quick else (1024, 2048, 4096)
  scales = (0.5, 2.0, 8.0) if quick else (0.5, 1.0, 2.0, 4.0, 8.0)
  for side in sides:
    for scale in scales:
      for alpha in ((0.1, 0.9), (0.75, 1.0)):
        if quick and alpha[0] > 0.5 and scale != 2.0:
          continue
        specs.append(dict(name=f's{scale:g}_a{alpha[0]:g}_{side}', kind='plain', n=counts[scale], side=side, scale=scale, alpha=alpha))
  for side in ((2048,) if quick else (1024, 2048, 4096)):
    specs.append(dict(name=f'culled_s1_{side}', kind='plain', n=6_000_000, side=side, scale=1.0, alpha=(0.1, 0.9), margin=0.5))
    specs.append(dict(name=f'needle_s0.5_{side}', kind='needle', n=6_000_000, side=side, scale=0.5, alpha=(0.1, 0.9)))
  specs.append(dict(name='culled_needle_2048', kind='needle', n=6_000_000, side=2048, scale=0.7, alpha=(0.1, 0.9), margin=0.5))
  specs.append(dict(name='heavy_tail_2048', kind='heavy_tail', n=2_000_000, side=2048, scale=1.0, alpha=(0.1, 0.9)))
  specs.append(dict(name='pile_400k_2048', kind='pile', n=400_000, side=2048, scale=1.0, alpha=(0.1, 0.9)))
  specs.append(dict(name='dense2d_1024x768', kind='dense2d'))
  return specs


def main():
  p = argparse.ArgumentParser()
  p.add_argument('--quick', action='store_true')
  p.add_argument('--out', default='')
  p.add_argument('--only', default='')
  args = p.parse_args()
  dev = torch.device('cuda', 0)
  x = torch.rand(4096, 4096, device=dev)
  t0 = time.perf_counter()
  while time.perf_counter() - t0 < 1.0:          # clocks up before the first scene is timed
    x = (x @ x).clamp_(0, 1)
  torch.cuda.synchronize()
  del x
  rows = []
  for spec in scenes(args.quick):
    if args.only and args.only not in spec['name']:
      continue
    try:
      r = measure(spec, dev)
    except Exception as e:                                  # a scene that does not fit is reported, not fatal
      r = dict(name=spec['name'], error=f"{type(e).__name__}: {e}"[:300])
    rows.append(r)
    print("SWEEP " + json.dumps(r), flush=True)
    torch.cuda.empty_cache()
  zoom = None
  if not args.only or 'zoom' in args.only:
    try:
      zoom = measure_zoom(dev)
    except Exception as e:
      zoom = dict(name='zoom_3M_2048', error=f"{type(e).__name__}: {e}"[:300])
    print("SWEEP " + json.dumps(zoom), flush=True)
    torch.cuda.empty_cache()
  ref = next((r for r in rows if r['name'] == 'configD' and 'error' not in r), None)
  lines = []
  head = f"{'scene':22s} {'N':>8s} {'K':>10s} {'K/N':>6s} {'K/tile':>7s} {'max run':>8s} | {'proj':>5s} {'sh':>5s} {'auto':>6s} {'dir':>6s} {'pre':>6s} {'fwd':>6s} {'bwd':>6s} {'frame':>6s} {'first':>7s} | {'map':>5s} {'fwd':>5s} {'bwd':>5s}  ps/overlap (x config D) | mapper  lists"
  lines.append(head)
  flagged = []
  for r in rows:
    if 'error' in r:
      lines.append(f"{r['name']:22s} ERROR {r['error']}")
      continue
    rel = {}
    for s in ('map', 'fwd', 'bwd'):
      rel[s] = r[f'{s}_ps_per_overlap'] / ref[f'{s}_ps_per_overlap'] if ref else float('nan')
      if ref and rel[s] > GUARD:
        flagged.append((r['name'], s, round(rel[s], 2)))
    if r['map_ms'] > 1.15 * r['map_best_ms'] + 0.02:
      flagged.append((r['name'], 'mapper-choice', round(r['map_ms'] / r['map_best_ms'], 2)))
    if 'first_frame_ms' in r and r['first_frame_ms'] > FIRST_GUARD * r['frame_ms']:
      flagged.append((r['name'], 'first-frame', round(r['first_frame_ms'] / r['frame_ms'], 2)))
    lines.append(f"{r['name']:22s} {r['N']:8d} {r['K']:10d} {r['K_per_N']:6.2f} {r['K_per_tile_mean']:7.0f} {r['K_per_tile_max']:8d} | "
                 f"{r['project_ms']:5.2f} {r['sh_ms']:5.2f} {r['map_auto_ms']:6.3f} {r['map_direct_ms']:6.3f} {r['map_presort_ms']:6.3f} "
                 f"{r['raster_fwd_ms']:6.3f} {r['raster_bwd_ms']:6.3f} {r.get('frame_ms', float('nan')):6.2f} {r.get('first_frame_ms', float('nan')):7.1f} | "
                 f"{r['map_ps_per_overlap']:5.1f} {r['fwd_ps_per_overlap']:5.1f} {r['bwd_ps_per_overlap']:5.1f}  "
                 f"({rel['map']:.2f} {rel['fwd']:.2f} {rel['bwd']:.2f}) | {r.get('frame_mapper', '-'):7s} {'same' if r['lists_equal'] else 'DIFFER'}"
                 + (f"   long runs in segments (one workgroup per tile: fwd {r['raster_fwd_per_tile_ms']:.3f} bwd {r['raster_bwd_per_tile_ms']:.3f} ms)"
                    if 'raster_fwd_per_tile_ms' in r else ""))
  lines.append("")
  lines.append("first = the first frame of the shape with the executor's shape caches cold and the allocator warm, wall time incl. "
               "the final synchronise (ms); the very first frame of a shape in a fresh allocator: first_frame_wall_ms in the SWEEP lines")
  if zoom is not None:
    lines.append("")
    lines.append(f"zoom path ({zoom['name']}): K/N {zoom.get('K_per_N')} settled on mapper {zoom.get('settled_mapper')} at {zoom.get('steady_ms')} ms; "
                 f"alternating every 3 frames: frames {zoom.get('frame_ms')} mapper {zoom.get('frame_mapper')} -> mean {zoom.get('mean_over_steady')} x, "
                 f"worst frame {zoom.get('worst_over_steady')} x its scale's steady frame" if 'error' not in zoom else f"zoom path ERROR {zoom['error']}")
    if 'error' not in zoom and zoom['worst_over_steady'] > 1.3:
      flagged.append((zoom['name'], 'zoom-worst-frame', zoom['worst_over_steady']))
  lines.append(f"guard: cost per overlap of mapper / raster forward / raster backward <= {GUARD} x config D's; first frame <= {FIRST_GUARD} x steady; "
               "zoom path worst frame <= 1.3 x steady")
  lines.append("flagged: " + (", ".join(f"{n}:{s} x{x}" for n, s, x in flagged) if flagged else "none"))
  text = "\n".join(lines)
  print(text)
  if args.out:
    Path(args.out).write_text(text + "\n\n" + "\n".join("SWEEP " + json.dumps(r) for r in rows + ([zoom] if zoom else [])) + "\n")


if __name__ == '__main__':
  main()
