#!/bin/bash
# Every step of the eager bench under the tracer: is the wall-clock excess of some boxes in the first frames behind the
# synchronisation that opens the timed region?
out=gpurun_out/r06l
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
python bench.py --steps 20 --warmup 5 --no-stages --no-sweep --no-cpu-baseline --no-train-step > $out/bench.log 2>&1
tail -1 $out/bench.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("wall: eager", d["ms_per_step"], "graph", d.get("graph_ms_per_step"), "lazy", d.get("lazy_settle",{}).get("ms_per_step"))'
MS_BENCH_SPIN_SECONDS=0.3 rocprofv3 --kernel-trace -d $out/eager -o t -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-stages --no-graph --no-sweep --no-train-step > $out/eager.log 2>&1
db=$(find $out/eager -name '*_results.db' | head -1); python tools/span_busy.py "$db" 32 --each | tee $out/eager_span.txt | tail -36
grep "timed" $out/eager.log
rm -rf $out/eager
