mkdir -p gpurun_out/r4p
for size in 2048 4096; do for w in 2 4 8; do
  timeout 600 python tools/emulate_sharded.py --strips --world $w --size $size --out gpurun_out/r4p/emul_strips_${w}_${size}.json > /dev/null 2> gpurun_out/r4p/emul_strips_${w}_${size}.err
  timeout 600 python tools/emulate_sharded.py --static --world $w --size $size --out gpurun_out/r4p/emul_sharded_${w}_${size}.json > /dev/null 2> gpurun_out/r4p/emul_sharded_${w}_${size}.err
done; done
for f in gpurun_out/r4p/emul_*.json; do python -c "
import json
d=json.load(open('$f')); print('$f'.split('/')[-1], d['single_gpu_ms'], d['max_rank_ms_eager'], d['max_rank_ms_graph'], d['compute_only_speedup_eager'], d['compute_only_speedup_graph'])"; done
