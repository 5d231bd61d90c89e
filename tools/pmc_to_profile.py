#!/usr/bin/env python
"""Condense the rocprofv3 PMC passes of tools/pmc_collect.sh (summary.json) into the committed counter file that
bench.py reads for `roofline.traffic` and `roofline.compute`:

    python tools/pmc_to_profile.py gpurun_out/pmc_xxx/summary.json N SIZE TILE K > profiles/r02_raster_bwd_counters.json

Derivations (MI355X_MICROARCH.md, "rocprofv3 PMC slots" / "HBM"):
  * FETCH_SIZE / WRITE_SIZE are reported in KB (x 1024 = bytes per launch); FETCH_SIZE under-reports wide coalesced
    streams by 2x on gfx950 — these kernels gather 4..16 B words, so the raw value is kept and the x2 value given
    as an upper bound;
  * SQ_ACTIVE_INST_VALU counts quad-cycles (x4 = SIMD cycles a VALU instruction occupies its SIMD);
  * GRBM_GUI_ACTIVE is summed over the 8 XCDs: clock = GRBM_GUI_ACTIVE / 8 / duration;
  * VALU issue peak = 1024 SIMDs x clock / 2 cycles: a wave64 FP32 FMA / MUL issues in 2 cycles (the chip's
    157 TFLOP/s vector FP32 figure); DPP, select, compare, min/max and integer VALU instructions take 4, so a
    kernel made of those tops out at 0.5 of this peak (tools/ubench_valu.hip, tools/ubench_scan.hip).
"""
import json
import sys

summary, n, size, tile, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
d = json.load(open(summary))


def kernel(prefix):
  for name, v in d.items():
    if prefix in name:
      return name, v
  raise SystemExit(f"no kernel matching {prefix} in {summary}")


def condense(prefix, algorithmic_bytes):
  name, c = kernel(prefix)
  dur = c['mean_duration_us_under_pmc'] * 1e-6
  clock = c['GRBM_GUI_ACTIVE'] / 8 / dur
  valu_per_s = c['SQ_INSTS_VALU'] / dur
  peak = 1024 * clock / 2
  out = {
    "kernel": name.replace('void ', ''),
    "duration_us_under_pmc": round(dur * 1e6, 1),
    "fetch_bytes_raw": int(c['FETCH_SIZE'] * 1024), "fetch_bytes_x2_upper": int(c['FETCH_SIZE'] * 2048),
    "write_bytes": int(c['WRITE_SIZE'] * 1024),
    "traffic_bytes": int((c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024),
    "algorithmic_bytes": algorithmic_bytes,
    "compute": {
      "valu_instr_per_launch": int(c['SQ_INSTS_VALU']), "salu_instr_per_launch": int(c['SQ_INSTS_SALU']),
      "lds_instr_per_launch": int(c['SQ_INSTS_LDS']),
      "clock_ghz": round(clock / 1e9, 3),
      "valu_instr_per_s": round(valu_per_s / 1e9, 1), "peak": round(peak / 1e9, 1), "unit": "G wave64 VALU instr/s",
      "frac": round(valu_per_s / peak, 3),
      "valu_active_cycles_per_instr": round(4 * c['SQ_ACTIVE_INST_VALU'] / c['SQ_INSTS_VALU'], 2),
      "valu_busy_frac_of_wall": round(4 * c['SQ_ACTIVE_INST_VALU'] / 1024 / (clock * dur), 3),
      "exec_lane_util": round(c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_ACTIVE_INST_VALU']), 3),
      "wave_cycles_split": {"issuing": round(c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES'], 3),
                            "issue_stalled": round(c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'], 3),
                            "waiting_waitcnt_or_barrier": round(c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 3)},
      "lds_busy_cycles_per_launch": int(c['SQ_LDS_IDX_ACTIVE']), "lds_bank_conflict_cycles": int(c['SQ_LDS_BANK_CONFLICT']),
    },
    "raw_counters": {key: c[key] for key in sorted(c) if key.isupper()},
  }
  return out


p = size * size
f = 3
bwd = condense('raster_bwd_scan_kernel', (4 + 28 + 4 * f) * k + 8 * f * p + (28 + 4 * f) * k)
fwd = condense('raster_fwd_f32x3_kernel', (4 + 28 + 4 * f) * k + 4 * (f + 1) * p)
result = {
  "_comment": "rocprofv3 --pmc passes (tools/pmc_collect.sh: one counter group per run, kernel trace only) of "
              f"tools/prof_raster.py {n} {size} {tile} on one MI355X, condensed by tools/pmc_to_profile.py. Values per launch.",
  "workload": {"n": n, "width": size, "height": size, "tile": tile, "K": k},
  "traffic_bytes": bwd["traffic_bytes"],
  "compute": {key: bwd["compute"][key] for key in ("valu_instr_per_s", "peak", "unit", "frac", "valu_active_cycles_per_instr",
                                                     "valu_busy_frac_of_wall", "exec_lane_util", "clock_ghz", "valu_instr_per_launch")},
  "raster_bwd": bwd, "raster_fwd": fwd,
}
print(json.dumps(result, indent=1))
