mkdir -p gpurun_out/r5n
( time timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=15 ) > gpurun_out/r5n/pytest_full.txt 2>&1
tail -30 gpurun_out/r5n/pytest_full.txt
