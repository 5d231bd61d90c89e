mkdir -p gpurun_out/r05f
python bench.py > gpurun_out/r05f/bench_counters.log 2>&1
tail -1 gpurun_out/r05f/bench_counters.log | cut -c1-200
python bench.py --no-cpu-baseline --no-stages --no-sweep 2>&1 | grep -o '"ms_per_step": [0-9.]*'
