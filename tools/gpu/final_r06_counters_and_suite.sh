#!/bin/bash
# final session 1: counters of the raster kernels on the final tree, then the whole GPU suite
mkdir -p gpurun_out/r06f1
cd /root/repo
tools/pmc_collect.sh gpurun_out/r06f1/pmc > gpurun_out/r06f1/pmc.log 2>&1; tail -2 gpurun_out/r06f1/pmc.log | cut -c1-200
timeout 2400 python -m pytest tests -q -m gpu > gpurun_out/r06f1/suite.log 2>&1; tail -3 gpurun_out/r06f1/suite.log
