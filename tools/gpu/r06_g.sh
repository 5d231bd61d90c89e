#!/bin/bash
# Round 6, closing session: kernel trace of the default bench (no train step inside), default bench line with the fresh
# counters attached, the whole -m gpu suite.  -> gpurun_out/r06g/
out=gpurun_out/r06g
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
python bench.py > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-1500
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-stages --no-graph --no-sweep --no-train-step > $out/trace.log 2>&1
db=$(find $out/trace -name '*_results.db' | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" --last-steps 18 > $out/kernel_trace.txt; rm -f "$db"; fi
rm -rf $out/trace
head -32 $out/kernel_trace.txt | cut -c1-170
MS_DETERMINISTIC=1 rocprofv3 --kernel-trace --stats -d $out/trace_det -o t -- python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-stages --no-graph --no-sweep --no-train-step > $out/trace_det.log 2>&1
db=$(find $out/trace_det -name '*_results.db' | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" --last-steps 18 > $out/deterministic_trace.txt; rm -f "$db"; fi
rm -rf $out/trace_det
head -8 $out/deterministic_trace.txt | cut -c1-170
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.txt
( time timeout 1500 python -m pytest tests/ -q -m gpu --durations=25 ) > $out/pytest_full.txt 2>&1
tail -40 $out/pytest_full.txt
