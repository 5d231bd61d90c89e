#!/bin/bash
# Eager vs replay under the tracer, per kernel, on whatever box this is (the wall-clock gap is 0.5 % on some boxes, 4.5 % on others)
out=gpurun_out/r06k
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
python bench.py --steps 20 --warmup 5 --no-stages --no-sweep --no-cpu-baseline --no-train-step > $out/bench.log 2>&1
tail -1 $out/bench.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("wall: eager", d["ms_per_step"], "graph", d.get("graph_ms_per_step"), "lazy", d.get("lazy_settle",{}).get("ms_per_step"))'
rocprofv3 --kernel-trace -d $out/eager -o t -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-stages --no-graph --no-sweep --no-train-step > $out/eager.log 2>&1
db=$(find $out/eager -name '*_results.db' | head -1); python tools/span_busy.py "$db" 25 | tee $out/eager_span.txt; python tools/rocpd_stats.py "$db" --last-steps 25 > $out/eager_kernels.txt
rocprofv3 --kernel-trace -d $out/graph -o t -- python bench.py --graph-child --steps 30 > $out/graph.log 2>&1
db=$(find $out/graph -name '*_results.db' | head -1); python tools/span_busy.py "$db" 25 | tee $out/graph_span.txt; python tools/rocpd_stats.py "$db" --last-steps 25 > $out/graph_kernels.txt
grep GRAPH_MS $out/graph.log
head -12 $out/eager_kernels.txt | cut -c1-150; head -12 $out/graph_kernels.txt | cut -c1-150
grep "timed" $out/eager.log
rm -rf $out/eager $out/graph
