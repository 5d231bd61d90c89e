#!/bin/bash
mkdir -p gpurun_out/large
cd /root/repo
B="python bench.py --no-cpu-baseline --no-train-step --no-sweep --no-graph --no-stages --steps 5 --warmup 2"
for cfg in "24000000 4096" "48000000 4096" "96000000 8192"; do set -- $cfg; echo "== n=$1 size=$2"; timeout 900 $B --n $1 --size $2 > gpurun_out/large/n$1_s$2.log 2>&1; grep -E "timed|Traceback|Error|error" gpurun_out/large/n$1_s$2.log | head -5; tail -1 gpurun_out/large/n$1_s$2.log | cut -c1-400; done
