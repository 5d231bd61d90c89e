#!/bin/bash
# Round 6, fifth GPU session: replayed launch sequences (A/B on one box), optimiser with staged point words, the revised
# emulation, tests.  -> gpurun_out/r06e/
out=gpurun_out/r06e
mkdir -p $out
( time timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_frame.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_optim.py tests/test_gpu_sharded_static.py tests/test_gpu_multi.py -x -q --durations=8 ) > $out/pytest_e.txt 2>&1
tail -16 $out/pytest_e.txt
for g in 1 0 1 0; do
  MS_FRAME_GRAPHS=$g timeout 600 python bench.py --steps 40 --warmup 5 --no-stages --no-sweep --no-cpu-baseline --no-train-step > $out/bench_graphs${g}_$RANDOM.log 2>&1
done
for f in $out/bench_graphs*.log; do echo "$f $(tail -1 $f | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d.get("graph_ms_per_step"), d.get("lazy_settle"))')"; done
timeout 600 python bench.py --train-step --steps 10 > $out/train_step.log 2>&1; tail -1 $out/train_step.log | cut -c1-1500
MS_FRAME_GRAPHS=0 timeout 600 python bench.py --train-step --steps 10 > $out/train_step_nographs.log 2>&1; tail -1 $out/train_step_nographs.log | cut -c1-1500
timeout 900 python tools/emulate_sharded.py --static --world 8 --size 4096 --steps 20 --out $out/emul_sharded_8_4096.json > $out/emul.log 2>&1
tail -1 $out/emul.log | cut -c1-3500
