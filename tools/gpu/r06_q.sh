#!/bin/bash
# session Q: heuristics arithmetic of the raster backward (clamped mask, |q| (|tx| + |ty|)) — tests and the training iteration
mkdir -p gpurun_out/r06q
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_raster.py tests/test_gpu_frame.py tests/test_gpu_determinism.py tests/test_gpu_round5.py tests/test_gpu_round6.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r06q/tests.log 2>&1; tail -4 gpurun_out/r06q/tests.log
timeout 600 python bench.py --no-cpu-baseline --train-step --steps 20 > gpurun_out/r06q/bench.log 2>&1; grep -o '"train_step": {.*' gpurun_out/r06q/bench.log | cut -c1-900
