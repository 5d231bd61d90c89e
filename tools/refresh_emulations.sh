#!/bin/bash
# The twelve one-GPU emulations of the multi-GPU rank steps (both decompositions, N = 2 / 4 / 8, configs D and E)
#   tools/refresh_emulations.sh [tag]   ->   gpurun_out/<tag>_emul_{sharded,strips}_{N}_{size}.json
tag=${1:-r06}
for size in 2048 4096; do
  for w in 2 4 8; do
    python tools/emulate_sharded.py --static --world $w --size $size --steps 5 --warmup 2 --out gpurun_out/${tag}_emul_sharded_${w}_${size}.json > /dev/null 2>&1
    python tools/emulate_sharded.py --strips --world $w --size $size --steps 5 --warmup 2 --out gpurun_out/${tag}_emul_strips_${w}_${size}.json > /dev/null 2>&1
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob('gpurun_out/${tag}_emul_*.json')):
    d = json.load(open(f))
    print(f.split('/')[-1], 'single', d['single_gpu_ms'], 'max rank eager', d['max_rank_ms_eager'], 'graph', d['max_rank_ms_graph'], 'speedup', d['compute_only_speedup_eager'], d['compute_only_speedup_graph'])
PY
