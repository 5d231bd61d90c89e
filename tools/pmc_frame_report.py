#!/usr/bin/env python
"""Condense tools/pmc_frame.sh's passes into one JSON (profiles/r04_frame_counters.json):

* calibration: FETCH_SIZE x 1024 and WRITE_SIZE x 1024 of the known-byte kernels of tools/ubench_fetch.hip divided by
  the bytes they really read / write — which correction applies to which access pattern on this chip;
* per kernel of the frame (every kernel taking >= 1 % of the frame): duration under PMC, FETCH_SIZE + WRITE_SIZE bytes
  per launch (raw and with the calibration factor of its access pattern), the algorithmic bytes of SURVEY 8(d) and
  the ratio, VALU / SALU / LDS instruction counts.

  python tools/pmc_frame_report.py gpurun_out/pmc_frame > profiles/r04_frame_counters.json"""
import json
import re
import sys
from pathlib import Path

root = Path(sys.argv[1])
summ = json.load(open(root / 'summary.json'))
wl = dict(re.findall(r'(\w+)=(\d+)', open(root / 'workload.txt').read()))
N, V, K, size, tile = (int(wl[k]) for k in ('N', 'V', 'K', 'size', 'tile'))
P = size * size
T = ((size + tile - 1) // tile) ** 2
F, D = 3, 16
known = json.load(open(root / 'calibration_bytes.json'))

calib = {}
for name, v in summ.items():
  for cname, b in known.items():
    if cname.split('<')[0] in name and (('<' not in cname) or cname.split('<')[1].rstrip('>') in name):
      e = calib.setdefault(cname, dict(b))
      if 'FETCH_SIZE' in v:
        e['FETCH_SIZE_bytes'] = int(v['FETCH_SIZE'] * 1024)
      if 'WRITE_SIZE' in v:
        e['WRITE_SIZE_bytes'] = int(v['WRITE_SIZE'] * 1024)
for cname, e in calib.items():
  if 'read_bytes' in e and e.get('FETCH_SIZE_bytes'):
    e['FETCH_SIZE_over_read_bytes'] = round(e['FETCH_SIZE_bytes'] / e['read_bytes'], 3)
    if 'row_bytes_as_64B_sectors' in e:
      e['FETCH_SIZE_over_64B_sector_bytes'] = round(e['FETCH_SIZE_bytes'] / (e['row_bytes_as_64B_sectors'] + e['index_bytes']), 3)
      e['FETCH_SIZE_over_128B_line_bytes'] = round(e['FETCH_SIZE_bytes'] / (e['row_bytes_as_128B_lines'] + e['index_bytes']), 3)
  if 'write_bytes' in e and e.get('WRITE_SIZE_bytes'):
    e['WRITE_SIZE_over_write_bytes'] = round(e['WRITE_SIZE_bytes'] / e['write_bytes'], 3)

# algorithmic bytes per launch (SURVEY 8(d) figures, this design's streams), by kernel-name substring
passes_k = (max(1, (T - 1).bit_length()) + 7) // 8
alg = [
  ('project_fwd', 44 * N + 36 * N),
  ('sh_fwd_rows_deg3', (4 * F * D + 16) * N + 4 * F * N),
  ('tile_count', 60 * V),
  ('tile_emit', 36 * V + 8 * K),
  ('find_ranges', 4 * K + 8 * T),
  ('raster_fwd_f32x3', (4 + 28 + 4 * F) * K + 4 * (F + 1) * P),
  ('raster_bwd_scan', (4 + 28 + 4 * F) * K + 8 * F * P + (28 + 4 * F) * K),
  ('gaussian_bwd', 392 * N),
  # direct-order mapper (later entries win over the shorter names above)
  ('tile_count_direct', 36 * N),
  ('tile_emit_direct', 36 * N + 12 * K),
  ('radix_upsweep_kernel<unsigned long', 8 * K),
  ('radix_downsweep_kernel<unsigned long', 24 * K),
  ('find_ranges2_u64', 8 * K + 8 * T),
  ('tile_depth_sort_kernel<4>', 16 * K),
]
# the factor FETCH_SIZE under-reports by, measured on the calibration kernels of the same run
stream = calib.get('calib_stream128', {}).get('FETCH_SIZE_over_read_bytes')
fetch_correction = round(1.0 / stream, 3) if stream else None
frame_us = 0.0
kern = {}
for name, v in summ.items():
  if 'calib_' in name or 'mean_duration_us_under_pmc' not in v:
    continue
  kern[name] = v
  frame_us += v['mean_duration_us_under_pmc'] * max(1, v.get('launches_seen', 1)) / 3.0
out = {"workload": {"n": N, "V": V, "K": K, "width": size, "height": size, "tile": tile},
       "calibration": calib,
       "calibration_note": "FETCH_SIZE / WRITE_SIZE are in KB (x 1024); the ratios above say what the counter reports for "
                           "a byte count that is KNOWN: wide coalesced streams (stream128), narrow coalesced streams "
                           "(stream32), random 28 B-row gathers against the bytes, the 64 B sectors and the 128 B lines "
                           "they touch, and 32 B-aligned rows",
       "fetch_correction": fetch_correction,
       "fetch_correction_note": "FETCH_SIZE x 1024 x fetch_correction = bytes read: the counter reports half of a coalesced "
                                "stream (both widths) and half of the 128 B lines a random row gather touches; WRITE_SIZE x 1024 "
                                "is exact.  `traffic_over_algorithmic` below uses the corrected fetch",
       "kernels": {}}
for name, v in sorted(kern.items(), key=lambda kv: -kv[1]['mean_duration_us_under_pmc'] * kv[1].get('launches_seen', 1)):
  dur = v['mean_duration_us_under_pmc']
  launches = v.get('launches_seen', 1)
  e = {"mean_duration_us_under_pmc": round(dur, 1), "launches_in_3_frames": launches}
  if 'FETCH_SIZE' in v:
    e['fetch_bytes_raw'] = int(v['FETCH_SIZE'] * 1024)
  if 'WRITE_SIZE' in v:
    e['write_bytes_raw'] = int(v['WRITE_SIZE'] * 1024)
  for key in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_WAVES'):
    if key in v:
      e[key] = int(v[key])
  for sub, b in alg:
    if sub in name:
      e['algorithmic_bytes'] = b
      if 'fetch_bytes_raw' in e and 'write_bytes_raw' in e:
        e['traffic_over_algorithmic_raw'] = round((e['fetch_bytes_raw'] + e['write_bytes_raw']) / b, 2)
        if fetch_correction:
          e['traffic_bytes'] = int(e['fetch_bytes_raw'] * fetch_correction + e['write_bytes_raw'])
          e['traffic_over_algorithmic'] = round(e['traffic_bytes'] / b, 2)
          e['traffic_GBps_under_pmc'] = round(e['traffic_bytes'] / (dur * 1e-6) / 1e9, 1)
      e['algorithmic_GBps_under_pmc'] = round(b / (dur * 1e-6) / 1e9, 1)
  short = name.replace('void ', '').replace('ms::', '')
  out["kernels"][short[:120]] = e
print(json.dumps(out, indent=1))
