#!/bin/bash
# End-of-round session on one box: the profiles of the round (bench line, kernel trace, PMC passes), smoke, the whole
# -m gpu suite, the other named configurations, the twelve emulations.  -> gpurun_out/r06f/
out=gpurun_out/r06f
mkdir -p $out
tools/refresh_profiles.sh r06f > $out/refresh.log 2>&1
tail -3 $out/bench.log | cut -c1-1200
head -14 $out/kernel_trace.txt | cut -c1-160
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.txt
tail -2 $out/smoke.txt | cut -c1-600
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=25 ) > $out/pytest_full.txt 2>&1
tail -34 $out/pytest_full.txt
bash tools/round_numbers.sh > $out/round_numbers.txt 2>&1
grep -E "==|timed" $out/round_numbers.txt | cut -c1-200
bash tools/refresh_emulations.sh r06 > $out/emulations.txt 2>&1
cat $out/emulations.txt
rm -rf $out/trace
