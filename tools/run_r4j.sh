mkdir -p gpurun_out/r4j
timeout 1200 python -m pytest tests/test_gpu_raster.py tests/test_gpu_fuzz.py tests/test_gpu_explained.py tests/test_gpu_configs.py tests/test_gpu_frame.py tests/test_gpu_render.py tests/test_gpu_strips.py -q -x > gpurun_out/r4j/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r4j/pytest.log
for i in 1 2; do timeout 300 python bench.py --no-sweep --no-graph --no-cpu-baseline --steps 200 > gpurun_out/r4j/bench_$i.json 2> gpurun_out/r4j/bench_$i.err; done
tail -4 gpurun_out/r4j/pytest.log
for i in 1 2; do python -c "
import json
d=json.loads(open('gpurun_out/r4j/bench_$i.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['frame']['stage_ms'])"; done
