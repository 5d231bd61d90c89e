mkdir -p gpurun_out/r05f gpurun_out/r05
bash tools/trace_frame.sh r05/final > /dev/null 2>&1
head -12 gpurun_out/r05/final_trace.txt | cut -c1-140
python bench.py > gpurun_out/r05f/bench_counters.log 2>&1
grep "^\[bench" gpurun_out/r05f/bench_counters.log | cut -c1-100 | sed -n 3,6p

