"""Timeline of the LAST step in a rocprofv3 rocpd database: start, duration and idle gap of every kernel
after the last dispatch whose name contains ``anchor`` (default: project_fwd_kernel).
    python tools/trace_timeline.py gpurun_out/prof/x_results.db [anchor]"""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else 'project_fwd_kernel'
rows = con.execute("select name,start,end from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if anchor in r[0]]
sub = rows[idx[-1]:]
t0, prev, busy = sub[0][1], None, 0
for n, s, e in sub:
  gap = (s - prev) / 1e3 if prev else 0
  busy += e - s
  n = re.sub(r'\(.*', '', n).replace('void ', '')[:70]
  print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} gap {gap:7.1f}  {n}")
  prev = e
print('span_us', (sub[-1][2] - t0) / 1e3, 'busy_us', busy / 1e3, 'kernels', len(sub))
