#!/bin/bash
# Kernel trace of 15 + 3 frames of the default bench workload -> gpurun_out/<tag>_trace.txt (per-kernel table)
tag=${1:-trace}; shift
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --kernel-trace -d "$out/trace" -o t -- python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-stages --no-graph --no-sweep --spin-up 0 "$@" > "$out/trace.log" 2>&1
db=$(find "$out/trace" -name '*_results.db' | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" --steps 18 > "gpurun_out/${tag}_trace.txt"; rm -rf "$out/trace"; fi
head -30 "gpurun_out/${tag}_trace.txt"
