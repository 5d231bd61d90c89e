mkdir -p gpurun_out/r4m
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
timeout 1200 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_frame.py tests/test_gpu_mapper.py tests/test_gpu_primitives.py tests/test_gpu_render.py tests/test_gpu_strips.py tests/test_gpu_sharded_static.py tests/test_gpu_round4.py tests/test_gpu_configs.py -q > gpurun_out/r4m/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r4m/pytest.log
rocprofv3 --kernel-trace --stats -d gpurun_out/r4m/trace -o t -- python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-stages --no-graph --no-sweep > gpurun_out/r4m/trace.log 2>&1
db=$(find gpurun_out/r4m/trace -name '*_results.db' | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" --last-steps 18 > gpurun_out/r4m/kernel_trace.txt; rm -f "$db"; fi
tail -4 gpurun_out/r4m/pytest.log; head -34 gpurun_out/r4m/kernel_trace.txt
