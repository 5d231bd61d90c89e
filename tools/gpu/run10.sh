timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_raster.py tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -15
