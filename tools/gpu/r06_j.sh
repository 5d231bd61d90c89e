#!/bin/bash
# Does a longer clock ramp change the eager number on this box?  (boxes differ: 3.05 ... 3.22 ms for the same tree)
out=gpurun_out/r06j
mkdir -p $out
for rep in 1 2; do
  for spin in 0.5 2.5 6 12; do
    MS_BENCH_SPIN_SECONDS=$spin python bench.py --steps 20 --warmup 5 --no-stages --no-sweep --no-cpu-baseline --no-train-step > $out/b_${spin}_$rep.log 2>&1
    echo "spin $spin rep $rep: $(tail -1 $out/b_${spin}_$rep.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("graph_ms_per_step"), d.get("lazy_settle",{}).get("ms_per_step"))')"
  done
done
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | head -30
