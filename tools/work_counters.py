#!/usr/bin/env python
"""What one launch of the raster backward computes, and what its phases cost alone — the inputs of the VALU-roofline
statement in bench.py's `roofline.compute`.  Condenses, from one GPU session
(tools/gpu/refresh_r05.sh), the RBENCH lines of tools/rbench.py for

  * the -DMS_SCAN_STATS build   (sub-patch hits, chunks, fill, executed pixel steps, CONTRIBUTING (pixel, splat) pairs),
  * the -DMS_SCAN_PHASES build  (share of a wave's cycles per phase),
  * the ablation builds         (-DMS_SCAN_ABLATE=2: staging only; =1: no blend; -DMS_COMMIT_ABLATE=3: no commit traffic),
  * the product build,

and the table of tools/ubench_blend.bin (the blend phase alone, 1..8 waves per SIMD):

    python tools/work_counters.py gpurun_out/r05 > gpurun_out/r05/work.json
"""
import json
import re
import sys
from pathlib import Path

d = Path(sys.argv[1])


def rbench(name):
  f = d / f'rbench_{name}.txt'
  if not f.exists():
    return None
  for line in f.read_text().splitlines():
    if line.startswith('RBENCH '):
      return json.loads(line[7:])
  return None


out = {}
prod, stats, phases = rbench('product'), rbench('stats'), rbench('phases')
if stats:
  out['counts'] = dict(stats['stats'], K=stats['K'], V=stats['V'])
if phases:
  out['wave_cycle_share'] = phases['phases_frac']
  out['cycles_per_chunk'] = phases['cycles_per_chunk']
  out['phases_build_ms'] = phases['bwd_ms_mean_med_min'][1]
alone = {}
for key, name in (('product', 'product'), ('staging_only', 'abl2'), ('no_blend', 'abl1'), ('no_commit_traffic', 'commit3'),
                  ('commit_plain_stores', 'commit2'), ('commit_one_lane_per_row', 'commit1')):
  r = rbench(name)
  if r:
    alone[key + '_ms'] = r['bwd_ms_mean_med_min'][1]
out['kernel_ms_same_box'] = alone
ub = d / 'ubench_blend.txt'
if ub.exists():
  table = {}
  for line in ub.read_text().splitlines():
    m = re.match(r'^(\d+)\s+([\d.]+) ns\s+([\d.]+) ns', line)
    if m:
      w = int(m.group(1))
      table[str(w)] = {"ns_per_chunk_per_wave_slot": float(m.group(2)), "ns_per_chunk_per_simd": round(float(m.group(2)) / w, 1),
                       "without_dpp_scans_ns": float(m.group(3))}
  out['blend_alone'] = table
  if stats and '4' in table:
    chunks = stats['stats']['chunks']
    out['blend_alone_floor_ms'] = round(chunks / 1024 * table['4']['ns_per_chunk_per_simd'] * 1e-6, 4)
    out['blend_alone_floor_is'] = ("chunks of the launch / 1024 SIMDs x time per chunk of tools/ubench_blend.bin at 4 waves per SIMD "
                                   "(all 16 pixel steps, every lane slot; no staging, cull, commit, barrier, global memory)")
print(json.dumps(out, indent=1))
