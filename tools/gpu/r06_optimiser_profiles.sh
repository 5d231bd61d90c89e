#!/bin/bash
# Round 6, fourth GPU session: optimiser trace + FETCH / WRITE counters, train-step kernel trace, emulation with more
# hardware queues, the fixed test.  -> gpurun_out/r06d/
out=gpurun_out/r06d
mkdir -p $out
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --kernel-trace --stats -d $out/optim_trace -o t -- python tools/prof_optim.py 6000000 8 > $out/optim_trace.log 2>&1
db=$(find $out/optim_trace -name '*_results.db' | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $out/optim_kernel_trace.txt; rm -f "$db"; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/optim_pmc/$c -o p -- python tools/prof_optim.py 6000000 3 > $out/optim_pmc_$c.log 2>&1 || echo "pass $c failed"
done
python tools/pmc_summary.py $out/optim_pmc optim > $out/optim_pmc_summary.json 2>&1; python tools/pmc_summary.py $out/optim_pmc visibility >> $out/optim_pmc_summary.json 2>&1
cat $out/optim_kernel_trace.txt | head -12; cat $out/optim_pmc_summary.json
rocprofv3 --kernel-trace --stats -d $out/train_trace -o t -- python bench.py --train-step --steps 10 > $out/train_trace.log 2>&1
db=$(find $out/train_trace -name '*_results.db' | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" > $out/train_kernel_trace.txt; rm -f "$db"; fi
head -40 $out/train_kernel_trace.txt | cut -c1-200
timeout 300 python -m pytest tests/test_gpu_round6.py -x -q -k "sweep_shapes or first_frame or lazily or overflow" > $out/pytest_d.txt 2>&1; tail -5 $out/pytest_d.txt
GPU_MAX_HW_QUEUES=8 timeout 900 python tools/emulate_sharded.py --static --world 8 --size 4096 --steps 20 --out $out/emul_sharded_8_4096_q8.json > $out/emul_q8.log 2>&1
python -c "
import json; d=json.load(open('$out/emul_sharded_8_4096_q8.json')); print(json.dumps(d['with_links'])[:1500]); print(d['max_rank_ms_graph'], d['single_gpu_ms'])"
rm -rf $out/optim_trace $out/train_trace
