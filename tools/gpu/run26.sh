timeout 600 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "long_tile_runs or splat_row" 2>&1 | tail -15
