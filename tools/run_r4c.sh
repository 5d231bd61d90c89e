mkdir -p gpurun_out/r4c
rm -f gpurun_out/parity_excess.jsonl
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_explained.py tests/test_gpu_configs.py tests/test_gpu_projection_sh.py tests/test_gpu_raster.py tests/test_gpu_frame.py tests/test_gpu_render.py tests/test_gpu_sharded_static.py -q --durations=25 > gpurun_out/r4c/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4c/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4c/smoke.log 2>&1
timeout 600 python bench.py --no-sweep > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err
tail -12 gpurun_out/r4c/pytest.log; cat gpurun_out/r4c/smoke.log | tail -3
