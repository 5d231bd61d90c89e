#!/bin/bash
# session N: the tile-bins mapper — stage timings
mkdir -p gpurun_out/r06n
cd /root/repo
for sc in D E; do timeout 300 python tools/mapper_bins_bench.py --scene $sc > gpurun_out/r06n/bins_$sc.log 2>&1; grep -v RBINS gpurun_out/r06n/bins_$sc.log | tail -17; done
timeout 300 python tools/mapper_bins_bench.py --scene D --morton > gpurun_out/r06n/bins_D_morton.log 2>&1; grep -v RBINS gpurun_out/r06n/bins_D_morton.log | tail -17
