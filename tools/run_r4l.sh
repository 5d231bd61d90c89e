mkdir -p gpurun_out/r4l
timeout 1200 python -m pytest tests/test_gpu_frame.py tests/test_gpu_mapper.py tests/test_gpu_primitives.py tests/test_gpu_fullsize.py tests/test_gpu_render.py tests/test_gpu_strips.py tests/test_gpu_sharded_static.py tests/test_gpu_round4.py -q -x > gpurun_out/r4l/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r4l/pytest.log
for i in 1 2; do timeout 300 python bench.py --no-sweep --no-graph --no-cpu-baseline --steps 200 > gpurun_out/r4l/bench_$i.json 2> gpurun_out/r4l/bench_$i.err; done
tail -4 gpurun_out/r4l/pytest.log
for i in 1 2; do python -c "
import json
d=json.loads(open('gpurun_out/r4l/bench_$i.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['frame']['stage_ms'])"; done
