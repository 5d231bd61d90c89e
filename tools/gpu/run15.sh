mkdir -p gpurun_out/r5m
python tools/emulate_sharded.py --static --world 8 --size 4096 --steps 5 --warmup 2 --out gpurun_out/r5m/emul_sharded_8_4096.json > gpurun_out/r5m/emul_sharded_8_4096.log 2>&1
python tools/emulate_sharded.py --strips --world 8 --size 4096 --steps 5 --warmup 2 --out gpurun_out/r5m/emul_strips_8_4096.json > gpurun_out/r5m/emul_strips_8_4096.log 2>&1
python - <<PY
import json
for f in ('gpurun_out/r5m/emul_sharded_8_4096.json','gpurun_out/r5m/emul_strips_8_4096.json'):
  d=json.load(open(f)); print(json.dumps(d)[:3000])
PY
