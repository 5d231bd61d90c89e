#!/bin/bash
# timeline of the LAST sync-free rank step (sharded.ShardedStep) of one emulated rank at world size W:
#   tools/prof_rank_static.sh [W] [rank] [size]
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
w=${1:-8}; r=${2:-3}; size=${3:-2048}
rm -rf gpurun_out/prof_rank
rocprofv3 --kernel-trace -d gpurun_out/prof_rank -o r -- python tools/emulate_sharded.py --static --world $w --ranks $r --size $size --steps 10 --warmup 2 > gpurun_out/prof_rank.log 2>&1
db=$(find gpurun_out/prof_rank -name '*_results.db' | head -1)
python tools/trace_timeline.py $db project_fwd_kernel | cut -c1-130 > gpurun_out/prof_rank_timeline.txt
rm -rf gpurun_out/prof_rank
tail -75 gpurun_out/prof_rank_timeline.txt
