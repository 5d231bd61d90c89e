mkdir -p gpurun_out/r5g
timeout 600 python -m pytest tests/test_gpu_mapper.py tests/test_gpu_tile_sort.py tests/test_gpu_strips.py -x -q -m gpu > gpurun_out/r5g/pytest_mapper.txt 2>&1; tail -3 gpurun_out/r5g/pytest_mapper.txt
timeout 1500 python tools/sweep_scenes.py --out gpurun_out/r5g/sweep.txt > gpurun_out/r5g/sweep.log 2>&1
grep -v "^SWEEP" gpurun_out/r5g/sweep.log | tail -48
