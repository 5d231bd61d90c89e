timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_frame.py tests/test_gpu_tile_sort.py -x -q -m gpu 2>&1 | tail -15
