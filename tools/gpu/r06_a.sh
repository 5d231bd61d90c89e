#!/bin/bash
# Round 6, first GPU session: the new tests, the optimiser A/B, the default bench line.  -> gpurun_out/r06a/
out=gpurun_out/r06a
mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_optim.py tests/test_gpu_round5.py tests/test_gpu_frame.py tests/test_abi.py -x -q --durations=15 ) > $out/pytest_new.txt 2>&1
tail -25 $out/pytest_new.txt
timeout 600 python bench.py --train-step --steps 10 > $out/train_step.log 2>&1; tail -1 $out/train_step.log | cut -c1-1500
MS_OPTIM_VEC=0 timeout 600 python bench.py --train-step --steps 10 > $out/train_step_vec1.log 2>&1; tail -1 $out/train_step_vec1.log | cut -c1-1500
timeout 900 python bench.py --steps 20 --warmup 5 > $out/bench.log 2>&1; tail -1 $out/bench.log | cut -c1-3000
MS_LAZY_SETTLE=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-stages --no-sweep --no-cpu-baseline --no-train-step > $out/bench_strict.log 2>&1; tail -1 $out/bench_strict.log | cut -c1-600
