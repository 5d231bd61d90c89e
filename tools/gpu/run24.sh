mkdir -p gpurun_out/r5s
python -m pytest tests/test_gpu_frame.py -x -q -m gpu 2>&1 | tail -3
bash tools/trace_frame.sh r5s/dense > /dev/null 2>&1
MS_SPLAT_ROWS=1 bash tools/trace_frame.sh r5s/rows > /dev/null 2>&1
head -14 gpurun_out/r5s/dense_trace.txt | cut -c1-150
head -14 gpurun_out/r5s/rows_trace.txt | cut -c1-150
for i in 1 2; do
  python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-stages --no-sweep 2>&1 | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/dense /'
  MS_SPLAT_ROWS=1 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-stages --no-sweep 2>&1 | grep -o '"ms_per_step": [0-9.]*' | sed 's/^/rows /'
done
