timeout 600 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "cuts_long" 2>&1 | tail -12
