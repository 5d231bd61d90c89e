for cfg in "--tile 8" "--tile 16 --n 1500000" "--tile 16 --n 3000000" "--tile 32 --n 1000000"; do
  for m in direct presort; do
    MS_MAPPER=$m python bench.py $cfg --steps 40 --warmup 5 --no-sweep --no-cpu-baseline --no-graph 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', '$m', d['ms_per_step'], 'K/N', d['frame']['K_per_N'], 'K', d['frame']['K'])"
  done
done
