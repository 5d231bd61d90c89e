#!/usr/bin/env python
"""Condense the rocprofv3 PMC passes of tools/pmc_collect.sh (summary.json) into the committed counter file that
bench.py reads for `roofline.traffic` and `roofline.compute`:

    python tools/pmc_to_profile.py gpurun_out/pmc_xxx/summary.json N SIZE TILE K [bwd.s fwd.s] [--work work.json] > profiles/raster_bwd_counters.json

The file carries `kernel_source_sha16` (bench.py::kernel_source_sha16 of the tree it is generated in — run it on the
tree the counters were collected on) and the collection date: bench.py attaches the figures only to a binary built
from byte-identical kernel sources.

(bwd.s / fwd.s: hipcc -S output of raster_bwd_scan.hip / raster_fast.hip for the static instruction mix, tools/valu_mix.py)

Derivations (MI355X_MICROARCH.md, "rocprofv3 PMC slots" / "HBM"):
  * FETCH_SIZE / WRITE_SIZE are reported in KB (x 1024 = bytes per launch); FETCH_SIZE under-reports wide coalesced
    streams by 2x on gfx950 — these kernels gather 4..16 B words, so the raw value is kept and the x2 value given
    as an upper bound;
  * SQ_ACTIVE_INST_VALU comes out at 1.04-1.06 per VALU instruction on both kernels whatever their mix, and x4 it
    exceeds the wall clock on the forward kernel: it counts issue events, not the cycles an instruction occupies its
    SIMD, so it is NOT used for a busy figure (it was in the first version of this file).  The busy estimate prices
    the counted instructions with the per-class issue costs measured by tools/ubench_valu.hip / ubench_scan.hip and
    the kernel's static instruction mix (tools/valu_mix.py) against the cycles of the same PMC run;
  * GRBM_GUI_ACTIVE is summed over the 8 XCDs: clock = GRBM_GUI_ACTIVE / 8 / duration;
  * VALU issue peak = 1024 SIMDs x clock / 2 cycles: a wave64 FP32 FMA / MUL issues in 2 cycles (the chip's
    157 TFLOP/s vector FP32 figure); DPP, select, compare, min/max and integer VALU instructions take 4, so a
    kernel made of those tops out at 0.5 of this peak (tools/ubench_valu.hip, tools/ubench_scan.hip).
"""
import json
import sys

# Round 4 calibration (tools/ubench_fetch.hip under the same counters, profiles/r04_frame_counters.json): FETCH_SIZE x 1024
# is EXACTLY half of the bytes read for coalesced 128-bit and 32-bit streams, and 0.49 of the 128-byte lines (0.83-0.96
# of the 64-byte sectors) touched by random 28- / 32-byte row gathers; WRITE_SIZE x 1024 equals the bytes written.  The
# x2 therefore applies to every access pattern of these kernels and `traffic_bytes` is 2 x FETCH + WRITE.
FETCH_CORRECTION = 2.0
work_file = None
if '--work' in sys.argv:          # tools/work_counters.py output: what the launch computes, and the phase floors
  i = sys.argv.index('--work')
  work_file = sys.argv[i + 1]
  del sys.argv[i:i + 2]
summary, n, size, tile, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
asm = sys.argv[6:8] if len(sys.argv) >= 8 else None
d = json.load(open(summary))


def kernel(prefix):
  for name, v in d.items():
    if prefix in name:
      return name, v
  raise SystemExit(f"no kernel matching {prefix} in {summary}")


def condense(prefix, algorithmic_bytes, mix=None):
  name, c = kernel(prefix)
  dur = c['mean_duration_us_under_pmc'] * 1e-6
  clock = c['GRBM_GUI_ACTIVE'] / 8 / dur
  valu_per_s = c['SQ_INSTS_VALU'] / dur
  peak = 1024 * clock / 2
  out = {
    "kernel": name.replace('void ', ''),
    "duration_us_under_pmc": round(dur * 1e6, 1),
    "fetch_bytes_raw": int(c['FETCH_SIZE'] * 1024), "fetch_bytes": int(c['FETCH_SIZE'] * 1024 * FETCH_CORRECTION),
    "write_bytes": int(c['WRITE_SIZE'] * 1024),
    "traffic_bytes": int((c['FETCH_SIZE'] * FETCH_CORRECTION + c['WRITE_SIZE']) * 1024),
    "traffic_bytes_raw": int((c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024),
    "algorithmic_bytes": algorithmic_bytes,
    "compute": {
      "valu_instr_per_launch": int(c['SQ_INSTS_VALU']), "salu_instr_per_launch": int(c['SQ_INSTS_SALU']),
      "lds_instr_per_launch": int(c['SQ_INSTS_LDS']),
      "clock_ghz": round(clock / 1e9, 3),
      "valu_instr_per_s": round(valu_per_s / 1e9, 1), "peak": round(peak / 1e9, 1), "unit": "G wave64 VALU instr/s",
      "frac": round(valu_per_s / peak, 3),
      "static_mix": mix,
      "valu_busy_model": round(c['SQ_INSTS_VALU'] * mix["issue_cycles_per_instr"] / (1024 * c['GRBM_GUI_ACTIVE'] / 8), 3) if mix else None,
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
mix_bwd = mix_fwd = None
if asm:
  sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent))
  from valu_mix import mix as valu_mix
  mix_bwd = valu_mix(asm[0], f"_ZN2ms22raster_bwd_scan_kernelILi{tile}ELb0E")
  mix_fwd = valu_mix(asm[1], f"_ZN2ms23raster_fwd_f32x3_kernelILi{tile}ELb0E")
bwd = condense('raster_bwd_scan_kernel', (4 + 28 + 4 * f) * k + 8 * f * p + (28 + 4 * f) * k, mix_bwd)
fwd = condense('raster_fwd_f32x3_kernel', (4 + 28 + 4 * f) * k + 4 * (f + 1) * p, mix_fwd)
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
import bench as _bench   # noqa: E402  (fingerprint of the kernel sources)
import datetime          # noqa: E402
result = {
  "kernel_source_sha16": _bench.kernel_source_sha16(),
  "collected": datetime.date.today().isoformat(),
  "_comment": "rocprofv3 --pmc passes (tools/pmc_collect.sh: one counter group per run, kernel trace only) of "
              f"tools/prof_raster.py {n} {size} {tile} on one MI355X, condensed by tools/pmc_to_profile.py. Values per launch.",
  "workload": {"n": n, "width": size, "height": size, "tile": tile, "K": k},
  "traffic_bytes": bwd["traffic_bytes"],
  "traffic_correction": "2 x FETCH_SIZE + WRITE_SIZE (KB x 1024): FETCH_SIZE under-reports by exactly 2 on gfx950 for streams AND "
                        "for 28-byte row gathers (calibrated: tools/ubench_fetch.hip, profiles/r04_frame_counters.json)",
  "compute": {key: bwd["compute"][key] for key in ("valu_instr_per_s", "peak", "unit", "frac", "static_mix", "valu_busy_model",
                                                     "exec_lane_util", "clock_ghz", "valu_instr_per_launch")},
  "raster_bwd": bwd, "raster_fwd": fwd,
}
if work_file:
  result["work"] = json.load(open(work_file))
print(json.dumps(result, indent=1))
