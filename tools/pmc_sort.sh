#!/bin/bash
# PMC passes for the radix kernels on tools/bench_sort.py (development tool): tools/pmc_sort.sh <outdir> [k|v]
out=${1:-gpurun_out/pmc_sort}; which=${2:-k}; mkdir -p "$out"
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
pass() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -o p -- python tools/bench_sort.py $which > "$out/$name.log" 2>&1 || echo "pass $name failed"; }
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python tools/pmc_summary.py "$out" radix > "$out/summary.json"
python - "$out/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for name, c in d.items():
  if 'radix' not in name: continue
  dur = c['mean_duration_us_under_pmc']
  wc = c.get('SQ_WAVE_CYCLES', 0) or 1
  print(f"{name[:60]:60s} {dur:7.1f} us  VALU {c.get('SQ_INSTS_VALU',0)/1e6:7.2f} M  SALU {c.get('SQ_INSTS_SALU',0)/1e6:6.2f} M  LDS {c.get('SQ_INSTS_LDS',0)/1e6:6.2f} M  "
        f"issuing {c.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f} stalled {c.get('SQ_WAIT_INST_ANY',0)/wc:.2f} waiting {c.get('SQ_WAIT_ANY',0)/wc:.2f}  "
        f"lds_busy {c.get('SQ_LDS_IDX_ACTIVE',0)/1e6:7.1f} M conflicts {c.get('SQ_LDS_BANK_CONFLICT',0)/1e6:6.1f} M  "
        f"fetch {c.get('FETCH_SIZE',0)/1024:7.1f} MB write {c.get('WRITE_SIZE',0)/1024:7.1f} MB  launches {c.get('launches_seen')}")
PY
