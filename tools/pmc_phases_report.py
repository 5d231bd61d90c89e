#!/usr/bin/env python
"""Table of tools/pmc_phases.sh: SQ counters per launch of raster_bwd_scan_kernel for the product build and the builds
that leave phases out, the split of the waves' cycles each shows, and what the removed phase accounts for by difference.

    python tools/pmc_phases_report.py gpurun_out/pmc_phases > phases.txt
"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
NAMES = [('default', 'product kernel'), ('commit3', 'no commit traffic (-DMS_COMMIT_ABLATE=3)'),
         ('abl1', 'no blend (-DMS_SCAN_ABLATE=1): staging + cull + commit of zeros'), ('abl2', 'staging only (-DMS_SCAN_ABLATE=2)')]


def load(name):
  acc, cnt, dur = collections.defaultdict(float), collections.Counter(), []
  for f in glob.glob(f'{root}/{name}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
      if 'raster_bwd_scan' not in r['Kernel_Name']:
        continue
      acc[r['Counter_Name']] += float(r['Counter_Value'])
      cnt[r['Counter_Name']] += 1
  for f in glob.glob(f'{root}/{name}/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
      if 'raster_bwd_scan' in r['Kernel_Name']:
        dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
  out = {k: v / cnt[k] for k, v in acc.items()}
  if dur:
    out['duration_us_under_pmc'] = sum(dur) / len(dur)
  return out


rows = {n: load(n) for n, _ in NAMES}
cols = ['duration_us_under_pmc', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS',
        'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS',
        'SQ_ACTIVE_INST_VMEM', 'SQ_INST_CYCLES_VMEM', 'SQ_ACTIVE_INST_SCA', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS',
        'SQ_LDS_IDX_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'GRBM_GUI_ACTIVE']
print("# raster_bwd_scan_kernel<16> on config D, per launch (M = 1e6; SQ_* cycle counters are summed over all waves / SIMDs as rocprofv3 reports them)")
print(f"{'counter':26s}" + "".join(f"{n:>14s}" for n, _ in NAMES))
for c in cols:
  if not any(c in rows[n] for n, _ in NAMES):
    continue
  unit = 1.0 if c.startswith('duration') else 1e6
  print(f"{c:26s}" + "".join(f"{rows[n].get(c, float('nan')) / unit:14.1f}" for n, _ in NAMES))
print()
for n, what in NAMES:
  print(f"  {n:8s} = {what}")
print()
print("# split of the waves' cycles (SQ_WAVE_CYCLES = 100 %): issuing an instruction / stalled although an instruction is ready / waiting (s_waitcnt, barrier, sleep); of the waiting: on LDS")
for n, _ in NAMES:
  r = rows[n]
  w = r.get('SQ_WAVE_CYCLES')
  if not w or 'SQ_WAIT_ANY' not in r:
    continue
  print(f"  {n:8s} issuing {100 * r['SQ_ACTIVE_INST_ANY'] / w:5.1f} %   issue-stalled {100 * r['SQ_WAIT_INST_ANY'] / w:5.1f} %   waiting {100 * r['SQ_WAIT_ANY'] / w:5.1f} %"
        f"   (LDS-instruction wait {100 * r.get('SQ_WAIT_INST_LDS', float('nan')) / w:5.1f} %;  VALU issue {100 * r.get('SQ_ACTIVE_INST_VALU', float('nan')) / w:5.1f} %)")
print()
d = rows['default']
if d and rows['abl1'] and rows['commit3'] and rows['abl2']:
  def diff(a, b, c):
    return (rows[a].get(c, float('nan')) - rows[b].get(c, float('nan'))) / 1e6
  print("# by difference (M per launch): what the named phase adds to the kernel")
  print(f"{'phase':44s}{'VALU instr':>12s}{'LDS instr':>12s}{'wave cycles':>14s}{'waiting':>12s}{'time us':>10s}")
  for label, a, b in (('blend (product - no blend)', 'default', 'abl1'), ('commit traffic (product - no commit traffic)', 'default', 'commit3'),
                      ('cull + commit of zeros (no blend - staging only)', 'abl1', 'abl2')):
    print(f"{label:44s}{diff(a, b, 'SQ_INSTS_VALU'):12.1f}{diff(a, b, 'SQ_INSTS_LDS'):12.1f}{diff(a, b, 'SQ_WAVE_CYCLES'):14.1f}{diff(a, b, 'SQ_WAIT_ANY'):12.1f}"
          f"{rows[a].get('duration_us_under_pmc', float('nan')) - rows[b].get('duration_us_under_pmc', float('nan')):10.1f}")
  r = rows['abl2']
  print(f"{'staging only (gathers, records, LDS write)':44s}{r.get('SQ_INSTS_VALU', float('nan')) / 1e6:12.1f}{r.get('SQ_INSTS_LDS', float('nan')) / 1e6:12.1f}"
        f"{r.get('SQ_WAVE_CYCLES', float('nan')) / 1e6:14.1f}{r.get('SQ_WAIT_ANY', float('nan')) / 1e6:12.1f}{r.get('duration_us_under_pmc', float('nan')):10.1f}")
