mkdir -p gpurun_out/r05f gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_mapper.py tests/test_gpu_tile_sort.py tests/test_gpu_frame.py tests/test_gpu_properties.py -x -q -m gpu 2>&1 | tail -3
bash tools/trace_frame.sh r05/final > /dev/null 2>&1
head -14 gpurun_out/r05/final_trace.txt | cut -c1-140
python bench.py > gpurun_out/r05f/bench_counters.log 2>&1
grep "^\[bench" gpurun_out/r05f/bench_counters.log | cut -c1-330 | sed -n 3,8p
