#!/bin/bash
# Final confirmation on the committed tree: smoke, whole -m gpu suite, default bench line.  -> gpurun_out/r06i/
out=gpurun_out/r06i
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.txt
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=12 ) > $out/pytest_full.txt 2>&1
tail -24 $out/pytest_full.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_like.log 2>&1; tail -1 $out/bench_driver_like.log | cut -c1-600
