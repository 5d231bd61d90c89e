#!/usr/bin/env python
"""Static VALU instruction mix of one kernel from hipcc -S output, by issue-cost class (tools/ubench_valu.hip,
tools/ubench_scan.hip, measured on MI355X): wave64 FP32 FMA / MUL / ADD issue in 2 cycles; DPP forms, selects,
compares, min / max, moves, integer and conversion instructions in 4; transcendentals (exp, rcp, sqrt, rsq) in 8.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-fast-math -fno-slp-vectorize -S --cuda-device-only -o k.s x.hip
    python tools/valu_mix.py k.s _ZN2ms22raster_bwd_scan_kernelILi16ELb0EEE

The kernels' hot loops are fully unrolled and make up most of the static code, so the static mix is a usable
stand-in for the dynamic one; it prices an instruction stream, it is not a counter.
"""
import json
import re
import sys

TWO = re.compile(r'^v_(mul|add|sub|subrev|fma|fmac|fmamk|fmaak|mac|mad)_f32(_e32|_e64)?$')
TRANS = re.compile(r'^v_(exp|log|rcp|rsq|sqrt|sin|cos)_')


def mix(path, kernel):
  inside, counts = False, {"two_cycle": 0, "four_cycle": 0, "transcendental": 0}
  for line in open(path):
    if line.startswith(kernel) and ':' in line and not line.startswith((' ', '\t', '.')) and '.' not in line.split(':')[0]:
      inside = True
      continue
    if inside and 's_endpgm' in line:
      break
    if not inside:
      continue
    tok = line.split()
    if not tok or not tok[0].startswith('v_'):
      continue
    op = tok[0]
    if TRANS.match(op):
      counts["transcendental"] += 1
    elif TWO.match(op):
      counts["two_cycle"] += 1
    else:
      counts["four_cycle"] += 1          # includes *_dpp / *_sdwa forms of the arithmetic instructions
  n = sum(counts.values())
  cycles = 2 * counts["two_cycle"] + 4 * counts["four_cycle"] + 8 * counts["transcendental"]
  return {"static_valu_instructions": n, **counts, "issue_cycles_per_instr": round(cycles / max(n, 1), 3)}


if __name__ == '__main__':
  print(json.dumps(mix(sys.argv[1], sys.argv[2])))
