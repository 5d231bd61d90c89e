mkdir -p gpurun_out/r05f gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_mapper.py tests/test_gpu_tile_sort.py tests/test_gpu_round5.py -x -q -m gpu 2>&1 | tail -3
bash tools/trace_frame.sh r05/final > /dev/null 2>&1
sed -n 6,14p gpurun_out/r05/final_trace.txt | cut -c1-140
python bench.py --no-cpu-baseline --no-sweep 2>&1 | grep "timed\|replay\|stages" | cut -c1-330
