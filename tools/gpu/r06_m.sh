#!/bin/bash
# The bench after parking the collector before the spin-up: driver's command, three times; then the per-step timeline.
out=gpurun_out/r06m
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
for rep in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_$rep.log 2>&1
  tail -1 $out/bench_$rep.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("eager", d["ms_per_step"], "graph", d.get("graph_ms_per_step"), "lazy", d.get("lazy_settle",{}).get("ms_per_step"), "sweep", d.get("tile_sweep_ms"), "stages", d["frame"]["stage_ms"], "train", d["extra"]["train_step"]["iteration_ms_reference_loop"], d["extra"]["train_step"]["iteration_ms_dense_step"], d["extra"]["train_step"]["optimizer"]["step_ms"])'
done
MS_BENCH_SPIN_SECONDS=0.3 rocprofv3 --kernel-trace -d $out/eager -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stages --no-graph --no-sweep --no-train-step > $out/eager.log 2>&1
db=$(find $out/eager -name '*_results.db' | head -1); python tools/span_busy.py "$db" 27 --each | tee $out/eager_span.txt | tail -30
grep "timed" $out/eager.log
rm -rf $out/eager
