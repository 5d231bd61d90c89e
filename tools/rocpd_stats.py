#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (``*_results.db``) into a per-kernel table
(calls, total / average / min / max duration, share) — the equivalent of ``--stats`` CSV output.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--steps K] [--last-steps S [--anchor kernel]] > profiles/r01_xxx.txt
"""
import re
import sqlite3
import sys


def short(name: str) -> str:
  name = re.sub(r'\(.*\)$', '', name)
  name = name.replace('void ', '')
  return name if len(name) <= 110 else name[:107] + '...'


def main():
  path = sys.argv[1]
  steps = None
  if '--steps' in sys.argv:
    steps = int(sys.argv[sys.argv.index('--steps') + 1])
  con = sqlite3.connect(path)
  cur = con.cursor()
  cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
  name_col = 'name' if 'name' in cols else 'kernel_name'
  rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
  if '--last-steps' in sys.argv:
    # steady state only: keep the dispatches from the S-th last launch of the anchor kernel (first kernel of a step) on
    last = int(sys.argv[sys.argv.index('--last-steps') + 1])
    anchor = sys.argv[sys.argv.index('--anchor') + 1] if '--anchor' in sys.argv else 'project_fwd_kernel'
    marks = [i for i, r in enumerate(rows) if anchor in r[0]]
    if len(marks) >= last:
      rows = rows[marks[-last]:]
      steps = last
  agg = {}
  for name, s, e in rows:
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
  total = sum(a[1] for a in agg.values())
  print(f"# {path}: {len(rows)} kernel dispatches, total GPU kernel time {total / 1e6:.3f} ms"
        + (f" ({total / 1e6 / steps:.3f} ms per step over {steps} steps)" if steps else ""))
  print(f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
  for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{a[0]:>7} {a[1] / 1e6:>10.3f} {a[1] / a[0] / 1e3:>10.2f} {a[2] / 1e3:>10.2f} {a[3] / 1e3:>10.2f} "
          f"{100 * a[1] / total:>6.2f}  {short(name)}")


if __name__ == '__main__':
  main()
