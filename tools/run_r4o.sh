mkdir -p gpurun_out/r4o
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
rm -f gpurun_out/parity_excess.jsonl
timeout 1700 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r4o/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4o/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4o/smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/r4o/bench.json 2> gpurun_out/r4o/bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/r4o/trace -o t -- python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-stages --no-graph --no-sweep > gpurun_out/r4o/trace.log 2>&1
db=$(find gpurun_out/r4o/trace -name '*_results.db' | head -1)
if [ -n "$db" ]; then python tools/rocpd_stats.py "$db" --last-steps 18 > gpurun_out/r4o/kernel_trace.txt; rm -f "$db"; fi
tail -5 gpurun_out/r4o/pytest.log; tail -1 gpurun_out/r4o/smoke.log; cut -c1-260 gpurun_out/r4o/bench.json; head -3 gpurun_out/r4o/kernel_trace.txt
