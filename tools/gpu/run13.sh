mkdir -p gpurun_out/r5k
timeout 600 python -m pytest tests/test_gpu_determinism.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --no-cpu-baseline --steps 50 --no-graph --no-sweep"
pick() { grep -E "timed|stages" | sed 's/\[bench [0-9:]*\] //'; }
echo "== config D"; $B 2>&1 | pick
echo "== config D deterministic"; MS_DETERMINISTIC=1 $B 2>&1 | pick
