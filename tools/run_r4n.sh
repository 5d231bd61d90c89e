mkdir -p gpurun_out/r4n
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
timeout 900 python -m pytest tests/test_gpu_frame.py tests/test_gpu_mapper.py tests/test_gpu_primitives.py tests/test_gpu_render.py tests/test_gpu_strips.py tests/test_gpu_sharded_static.py -q > gpurun_out/r4n/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r4n/pytest.log
rocprofv3 --kernel-trace --stats -d gpurun_out/r4n/trace -o t -- python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-stages --no-graph --no-sweep > gpurun_out/r4n/trace.log 2>&1
db=$(find gpurun_out/r4n/trace -name '*_results.db' | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" --last-steps 18 > gpurun_out/r4n/kernel_trace.txt; rm -f "$db"; fi
for i in 1 2; do timeout 300 python bench.py --no-sweep --no-graph --no-cpu-baseline --no-stages --steps 200 2>/dev/null | cut -c1-200; done
tail -3 gpurun_out/r4n/pytest.log; grep "depth_\|per step" gpurun_out/r4n/kernel_trace.txt
