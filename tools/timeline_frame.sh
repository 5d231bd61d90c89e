#!/bin/bash
# Timeline (start, duration, gap) of the LAST frame of a short default-workload bench: tools/timeline_frame.sh <tag>
tag=${1:-tl}; shift
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --kernel-trace -d "$out/trace" -o t -- python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-stages --no-graph --no-sweep "$@" > "$out/trace.log" 2>&1
db=$(find "$out/trace" -name '*_results.db' | head -1)
python tools/trace_timeline.py "$db" project_fwd_kernel | cut -c1-140 > "gpurun_out/${tag}_timeline.txt"
rm -rf "$out/trace"
cat "gpurun_out/${tag}_timeline.txt"
