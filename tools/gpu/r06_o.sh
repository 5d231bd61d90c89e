#!/bin/bash
mkdir -p gpurun_out/r06o
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu -k "visibility" > gpurun_out/r06o/tests_vis.log 2>&1; tail -15 gpurun_out/r06o/tests_vis.log
