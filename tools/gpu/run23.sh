mkdir -p gpurun_out/r5r
( timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_frame.py tests/test_gpu_render.py tests/test_gpu_configs.py -x -q -m gpu ) > gpurun_out/r5r/pytest.txt 2>&1
tail -15 gpurun_out/r5r/pytest.txt
for i in 1 2; do
  python bench.py --steps 30 --warmup 10 > gpurun_out/r5r/bench_dense_$i.log 2>&1
  MS_SPLAT_ROWS=1 python bench.py --steps 30 --warmup 10 > gpurun_out/r5r/bench_rows_$i.log 2>&1
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r5r/bench_*.log')):
  for l in open(f):
    if l.startswith('{'):
      d=json.loads(l); print(f, d['ms_per_step'], d['value'], {k:v for k,v in d.get('stages_ms',{}).items()} if 'stages_ms' in d else '')
PY
o=gpurun_out/r5r/rbench.txt; : > $o
for i in 1 2; do
python tools/rbench.py --scene D --tag dense$i >> $o 2>&1
python tools/rbench.py --scene D --rows --tag rows$i >> $o 2>&1
done
grep RBENCH $o | python -c "
import sys,json
for l in sys.stdin:
  d=json.loads(l.split('RBENCH ')[1]); print(d['tag'], d['scene'], d['tile'], 'bwd', d['bwd_ms_mean_med_min'], 'fwd', d['fwd_ms_mean_med_min'])
"
