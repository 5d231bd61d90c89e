#!/bin/bash
# End-of-round check on one box: smoke, the default bench line (with the committed counters attached), the other named
# configurations, the whole -m gpu suite.  -> gpurun_out/r05f/
out=gpurun_out/r05f
mkdir -p $out
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.txt
python bench.py > $out/bench.log 2>&1
tail -1 $out/bench.log | cut -c1-400
bash tools/round_numbers.sh > $out/round_numbers.txt 2>&1
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=25 ) > $out/pytest_full.txt 2>&1
tail -22 $out/pytest_full.txt
