#!/usr/bin/env python
"""Per-step span, kernel-busy time and idle gaps of the last S steps in a rocprofv3 rocpd database: where an eager frame
and a HIP-graph replay of the same frame differ (kernel durations or the gaps between them).

    python tools/span_busy.py x_results.db [S] [anchor]
"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
last = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 15
anchor = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith('--') else 'project_fwd_kernel'
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name_col = 'name' if 'name' in cols else 'kernel_name'
rows = con.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if anchor in r[0]]
marks = marks[-(last + 1):]
spans, busys, gaps, counts = [], [], [], []
for a, b in zip(marks[:-1], marks[1:]):
  step = rows[a:b]
  spans.append((rows[b][1] - step[0][1]) / 1e3)
  busys.append(sum(e - s for _, s, e in step) / 1e3)
  counts.append(len(step))
  gaps.append(sum(max(0, step[i + 1][1] - step[i][2]) for i in range(len(step) - 1)) / 1e3 + max(0, rows[b][1] - step[-1][2]) / 1e3)
n = len(spans)
if '--each' in sys.argv:
  t_first = rows[marks[0]][1]
  for i, (a, b) in enumerate(zip(marks[:-1], marks[1:])):
    print(f"  step {i:3d}  starts at {(rows[a][1] - t_first) / 1e3:10.1f} us  span {spans[i]:8.1f}  busy {busys[i]:8.1f}  idle {gaps[i]:7.1f}")
print(f"steps {n}  kernels/step {sum(counts) / n:.1f}  span {sum(spans) / n:.1f} us  busy {sum(busys) / n:.1f} us  "
      f"idle between kernels {sum(gaps) / n:.1f} us ({sum(gaps) / sum(counts):.2f} us per launch)")
