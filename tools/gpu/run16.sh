timeout 1500 python -m pytest tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -30
python bench.py --n 120000 --size 512 --steps 3 --warmup 1 --no-cpu-baseline --gpus 8 --dry-run --no-graph --no-sweep 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d['dry_run'], indent=1))"
