mkdir -p gpurun_out/r4g
timeout 300 python bench.py --graph-child --steps 50 > gpurun_out/r4g/graph_child.log 2>&1
echo "rc=$?" >> gpurun_out/r4g/graph_child.log
timeout 300 python bench.py --no-sweep --no-graph --no-cpu-baseline --steps 200 > gpurun_out/r4g/bench_bcast.json 2> gpurun_out/r4g/bench_bcast.err
MS_BROADCAST_GRAD=0 timeout 300 python bench.py --no-sweep --no-graph --no-cpu-baseline --steps 200 > gpurun_out/r4g/bench_nobcast.json 2> gpurun_out/r4g/bench_nobcast.err
timeout 300 python bench.py --no-sweep --no-graph --no-cpu-baseline --steps 200 > gpurun_out/r4g/bench_bcast2.json 2> gpurun_out/r4g/bench_bcast2.err
tail -15 gpurun_out/r4g/graph_child.log
for f in bcast nobcast bcast2; do python -c "
import json
d=json.loads(open('gpurun_out/r4g/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['frame']['stage_ms'])"; done
