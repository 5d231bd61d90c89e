timeout 900 python -m pytest tests/test_gpu_sharded_static.py tests/test_gpu_sharded.py tests/test_gpu_strips.py -x -q -m gpu 2>&1 | tail -4
tools/refresh_emulations.sh r05 2>&1 | tail -14
