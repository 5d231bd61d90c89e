#!/bin/bash
# last session of the round: the whole GPU suite, smoke, the driver's bench command twice (final tree)
out=gpurun_out/r06z
mkdir -p $out
cd /root/repo
( time timeout 2400 python -m pytest tests -q -m gpu ) > $out/suite.log 2>&1; tail -6 $out/suite.log
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.txt
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_$i.log 2>&1; tail -1 $out/bench_$i.log | cut -c1-200; done
