#!/bin/bash
# final session 2 (fresh counter file in the tree): driver-like bench lines, kernel traces of the frame and of the training
# iteration, smoke, the other named configurations, the twelve emulations  -> gpurun_out/r06f2/
out=gpurun_out/r06f2
mkdir -p $out
cd /root/repo
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
for i in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_$i.log 2>&1; tail -1 $out/bench_$i.log | cut -c1-260; done
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-stages --no-graph --no-sweep --no-train-step > $out/trace.log 2>&1
db=$(find $out/trace -name '*_results.db' | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" --last-steps 18 > $out/kernel_trace.txt; rm -f "$db"; fi
head -8 $out/kernel_trace.txt | cut -c1-160
rocprofv3 --kernel-trace --stats -d $out/train_trace -o t -- python bench.py --train-step --steps 10 --no-cpu-baseline > $out/train_trace.log 2>&1
db=$(find $out/train_trace -name '*_results.db' | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $out/train_step_kernel_trace.txt; rm -f "$db"; fi
head -12 $out/train_step_kernel_trace.txt | cut -c1-160
rm -rf $out/trace $out/train_trace
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.txt 2>&1; echo "smoke rc=$?" | tee -a $out/smoke.txt
bash tools/round_numbers.sh > $out/round_numbers.txt 2>&1
grep -E "==|timed" $out/round_numbers.txt | cut -c1-200
bash tools/refresh_emulations.sh r06f2 > $out/emulations.txt 2>&1
cat $out/emulations.txt
