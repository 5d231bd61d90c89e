#!/bin/bash
# Regenerate everything profiles/ holds for one round, on the GPU box, into gpurun_out/<tag>/ (copy what is to be
# judged into profiles/ afterwards):  tools/refresh_profiles.sh r03x; then tools/pmc_to_profile.py ... > profiles/raster_bwd_counters.json
#   1. default bench line (python bench.py)                         -> bench.log
#   2. rocprofv3 --kernel-trace --stats of a 15-step bench           -> kernel_trace.txt (tools/rocpd_stats.py; the last 3 warm-up + 15 timed frames, after the spin-up)
#   3. PMC passes for the raster kernels (tools/pmc_collect.sh)      -> pmc/
tag=${1:-refresh}
out=gpurun_out/$tag
mkdir -p "$out"
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
python bench.py > "$out/bench.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/trace" -o t -- python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-stages --no-graph --no-sweep --no-train-step > "$out/trace.log" 2>&1
db=$(find "$out/trace" -name '*_results.db' | head -1)
# the table covers the LAST 18 frames (3 warm-up + 15 timed, after the clock spin-up): steady state, like the bench line
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" --last-steps 18 > "$out/kernel_trace.txt"; rm -f "$db"; fi
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
tools/pmc_collect.sh "$out/pmc" > "$out/pmc.log" 2>&1
tail -2 "$out/bench.log" | cut -c1-600
head -12 "$out/kernel_trace.txt"
