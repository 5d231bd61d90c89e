#!/bin/bash
mkdir -p gpurun_out/r06r
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "hip_graph" > gpurun_out/r06r/tests.log 2>&1; tail -25 gpurun_out/r06r/tests.log
