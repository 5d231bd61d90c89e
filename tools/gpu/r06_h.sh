#!/bin/bash
# Eager frame vs HIP-graph replay under the tracer on ONE box: kernel-busy time and idle gaps per step.  + new tests.
out=gpurun_out/r06h
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
python bench.py --steps 30 --no-stages --no-sweep --no-cpu-baseline --no-train-step > $out/bench.log 2>&1; tail -1 $out/bench.log | cut -c1-700
rocprofv3 --kernel-trace -d $out/eager -o t -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-stages --no-graph --no-sweep --no-train-step > $out/eager.log 2>&1
db=$(find $out/eager -name '*_results.db' | head -1); python tools/span_busy.py "$db" 25 | tee $out/eager_span.txt
rocprofv3 --kernel-trace -d $out/graph -o t -- python bench.py --graph-child --steps 30 > $out/graph.log 2>&1
db=$(find $out/graph -name '*_results.db' | head -1); python tools/span_busy.py "$db" 25 | tee $out/graph_span.txt
grep GRAPH_MS $out/graph.log
rm -rf $out/eager $out/graph
timeout 600 python -m pytest tests/test_gpu_sharded_static.py tests/test_gpu_round6.py tests/test_gpu_round4.py -x -q > $out/pytest_h.txt 2>&1; tail -4 $out/pytest_h.txt
