mkdir -p gpurun_out/r05f
python bench.py > gpurun_out/r05f/bench_counters.log 2>&1
grep "^\[bench" gpurun_out/r05f/bench_counters.log | cut -c1-100 | head -5
python bench.py --no-cpu-baseline --no-stages --no-sweep 2>&1 | grep "timed\|replay" | cut -c1-100
