#!/usr/bin/env python
"""Static VALU instruction mix of one kernel from hipcc -S output, by issue-cost class (tools/ubench_valu.hip,
tools/ubench_scan.hip, measured on MI355X): wave64 FP32 FMA / MUL / ADD issue in 2 cycles; DPP forms, selects,
compares, min / max, moves, integer and conversion instructions in 4; transcendentals (exp, rcp, sqrt, rsq) in 8.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -fno-slp-vectorize -S --cuda-device-only -o k.s x.hip
    python tools/valu_mix.py k.s _ZN2ms22raster_bwd_scan_kernelILi16ELb0EEE

Instructions are weighted by 10^(loop depth of their basic block) (the depth LLVM prints next to each block label),
so the innermost loops — the hit loop of the forward, the 16 pixel steps of the backward — dominate the way they do
at run time; it prices an instruction stream, it is not a counter.
"""
import json
import re
import sys

TWO = re.compile(r'^v_(mul|add|sub|subrev|fma|fmac|fmamk|fmaak|mac|mad)_f32(_e32|_e64)?$')
TRANS = re.compile(r'^v_(exp|log|rcp|rsq|sqrt|sin|cos)_')


DEPTH = re.compile(r'Depth=(\d+)')


def mix(path, kernel):
  inside, counts = False, {"two_cycle": 0.0, "four_cycle": 0.0, "transcendental": 0.0}
  weight, n_static = 1.0, 0
  for line in open(path):
    if line.startswith(kernel) and ':' in line and not line.startswith((' ', '\t', '.')) and '.' not in line.split(':')[0]:
      inside = True
      continue
    if inside and 's_endpgm' in line:
      break
    if not inside:
      continue
    if line.startswith('.LBB') or line.lstrip().startswith('; %bb.'):
      # a new basic block: depth 0 unless its label line says otherwise (the comment may sit on the next lines)
      m = DEPTH.search(line)
      weight = 10.0 ** int(m.group(1)) if m else 1.0
      continue
    if line.lstrip().startswith(';') and 'Depth=' in line and 'Loop' in line:
      weight = max(weight, 10.0 ** int(DEPTH.search(line).group(1)))
      continue
    tok = line.split()
    if not tok or not tok[0].startswith('v_'):
      continue
    op = tok[0]
    n_static += 1
    if TRANS.match(op):
      counts["transcendental"] += weight
    elif TWO.match(op):
      counts["two_cycle"] += weight
    else:
      counts["four_cycle"] += weight      # includes *_dpp / *_sdwa forms of the arithmetic instructions
  n = sum(counts.values())
  cycles = 2 * counts["two_cycle"] + 4 * counts["four_cycle"] + 8 * counts["transcendental"]
  shares = {k: round(v / max(n, 1e-30), 3) for k, v in counts.items()}
  return {"static_valu_instructions": n_static, **shares, "issue_cycles_per_instr": round(cycles / max(n, 1e-30), 3)}


if __name__ == '__main__':
  print(json.dumps(mix(sys.argv[1], sys.argv[2])))
