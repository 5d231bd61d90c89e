mkdir -p gpurun_out/r4f
rm -f gpurun_out/parity_excess.jsonl
timeout 1700 python -m pytest tests -m gpu -q --durations=30 > gpurun_out/r4f/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4f/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4f/smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err
tail -6 gpurun_out/r4f/pytest.log; tail -2 gpurun_out/r4f/smoke.log; cut -c1-300 gpurun_out/r4f/bench.json
