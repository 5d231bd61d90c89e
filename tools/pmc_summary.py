#!/usr/bin/env python
"""Per-kernel means of the counters collected by tools/pmc_collect.sh (rocprofv3 CSV output) as JSON."""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])
wanted = sys.argv[2] if len(sys.argv) > 2 else 'raster'      # substring of the kernel names to keep
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in sorted(root.glob('*/**/*counter_collection.csv')):
  for row in csv.DictReader(open(f)):
    name = row['Kernel_Name'].split('(')[0]
    if wanted not in name:
      continue
    acc[name][row['Counter_Name']].append(float(row['Counter_Value']))
for f in sorted(root.glob('*/**/*kernel_trace.csv')):
  for row in csv.DictReader(open(f)):
    name = row['Kernel_Name'].split('(')[0]
    if wanted in name:
      dur[name].append((int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1e3)
out = {}
for name, counters in acc.items():
  out[name] = {c: sum(v) / len(v) for c, v in counters.items()}
  out[name]['launches_seen'] = max(len(v) for v in counters.values())
  if dur[name]:
    out[name]['mean_duration_us_under_pmc'] = sum(dur[name]) / len(dur[name])
print(json.dumps(out, indent=1, sort_keys=True))
