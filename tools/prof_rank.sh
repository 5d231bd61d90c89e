#!/bin/bash
# kernel list of one emulated rank step at world size W (tools/emulate_sharded.py): tools/prof_rank.sh [W]
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
w=${1:-8}
rm -rf gpurun_out/prof_rank
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_rank -o r -- python tools/emulate_sharded.py --world $w --ranks 3 --steps 10 --warmup 2 > /dev/null 2>&1
db=$(find gpurun_out/prof_rank -name '*_results.db' | head -1)
python tools/trace_timeline.py $db route_count | cut -c1-120
rm -f $db
