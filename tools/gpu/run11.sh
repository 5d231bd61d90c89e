timeout 900 python -m pytest tests/test_gpu_raster.py -x -q -m gpu -k quantile 2>&1 | grep -v "^E   +" | tail -20
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_raster.py tests/test_gpu_fuzz.py tests/test_gpu_render.py -x -q -m gpu 2>&1 | tail -5
