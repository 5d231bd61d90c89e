#!/usr/bin/env python
"""Registers, scratch and LDS of every kernel of the hot files, from the compiler's own metadata (hipcc -S for gfx950; no
GPU needed).  Occupancy on CDNA4 is decided here: 512 VGPRs per SIMD lane in steps of 8 (<= 128: four waves per SIMD,
<= 168: three, <= 256: two), and a kernel that spills (scratch > 0) pays a vmcnt(0) per reload.  Round 5 lost 20 us of
the SH forward and 190 us of the fixed-point per-gaussian backward to register counts that had drifted across such a
step unnoticed; tests/test_kernel_budgets.py now holds the kernels below to their budgets.

    python tools/kernel_resources.py [file.hip ...]        # table on stdout
"""
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / 'taichi_splatting_amd' / 'csrc'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-fast-math', '-fno-slp-vectorize', '-S', '--cuda-device-only']
DEFAULT = ['raster_bwd_scan.hip', 'raster_fast.hip', 'sh.hip', 'gaussian_bwd.hip', 'projection.hip', 'mapper.hip']


def demangle(names):
  try:
    out = subprocess.run(['c++filt'], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return out.splitlines()
  except Exception:
    return names


def resources(path):
  """{demangled kernel name: dict(vgpr, sgpr, scratch, lds)} of one .hip file"""
  asm = subprocess.run(['/opt/rocm/bin/hipcc', *FLAGS, str(path), '-o', '-'], capture_output=True, text=True)
  if asm.returncode != 0:
    raise RuntimeError(f"hipcc -S {path}: {asm.stderr[-2000:]}")
  out, cur = {}, None
  for line in asm.stdout.splitlines():
    m = re.match(r'\s*\.amdhsa_kernel\s+(\S+)', line)
    if m:
      cur = dict(name=m.group(1))
      continue
    if cur is None:
      continue
    for key, field in (('vgpr', 'next_free_vgpr'), ('sgpr', 'next_free_sgpr'), ('scratch', 'private_segment_fixed_size'),
                       ('lds', 'group_segment_fixed_size')):
      m = re.match(rf'\s*\.amdhsa_{field}\s+(\d+)', line)
      if m:
        cur[key] = int(m.group(1))
    if '.end_amdhsa_kernel' in line:
      out[cur.pop('name')] = cur
      cur = None
  names = list(out)
  return {d: out[n] for n, d in zip(names, demangle(names))}


def waves_per_simd(vgpr):
  return min(8, 512 // (((vgpr + 7) // 8) * 8))


def main():
  files = [SRC / f for f in (sys.argv[1:] or DEFAULT)]
  with ThreadPoolExecutor(len(files)) as pool:
    tables = list(pool.map(resources, files))
  for f, table in zip(files, tables):
    print(f"== {f.name}")
    for name, r in sorted(table.items()):
      short = re.sub(r'\(.*', '', name).replace('void ', '')
      print(f"  {short[:86]:86s} vgpr {r.get('vgpr', 0):4d} ({waves_per_simd(r.get('vgpr', 1))} waves/SIMD)  sgpr {r.get('sgpr', 0):3d}  scratch {r.get('scratch', 0):4d}  lds {r.get('lds', 0):6d}")


if __name__ == '__main__':
  main()
