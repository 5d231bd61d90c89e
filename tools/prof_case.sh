#!/bin/bash
# kernel-trace summary of one case of the component harness: tools/prof_case.sh projection "backward (everything)"
cd /tmp; export TMPDIR=/tmp; cd - > /dev/null
stage=$1; case_name=$2
cat > /tmp/pc.py <<PY
import sys, torch
sys.path.insert(0, '.')
from taichi_splatting_amd.benchmarks import components as c
torch.manual_seed(0)
cases = c.WORKLOADS["$stage"][1](torch.device('cuda', 0))
fn = cases["$case_name"]
for _ in range(20): fn()
torch.cuda.synchronize()
print("ms", c.time_ms(fn, iters=100))
PY
rm -rf gpurun_out/prof_case
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_case -o pc -- python /tmp/pc.py 2>&1 | grep "^ms"
db=$(find gpurun_out/prof_case -name '*_results.db' | head -1)
python tools/rocpd_stats.py $db | head -24
rm -f $db
