#!/bin/bash
# session P: counters of the raster kernels on the tree with the twelve-column heuristics rows (fingerprint refresh),
# driver-like bench line, kernel trace
mkdir -p gpurun_out/r06p
cd /root/repo
tools/pmc_collect.sh gpurun_out/r06p/pmc > gpurun_out/r06p/pmc.log 2>&1; tail -3 gpurun_out/r06p/pmc.log | cut -c1-400
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06p/bench.log 2>&1; tail -1 gpurun_out/r06p/bench.log | cut -c1-700
