mkdir -p gpurun_out/r4d
rm -f gpurun_out/parity_excess.jsonl
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_configs.py tests/test_gpu_frame.py tests/test_gpu_sharded_static.py tests/test_gpu_sharded.py tests/test_gpu_multi.py tests/test_gpu_strips.py tests/test_gpu_raster.py -q -x --durations=15 > gpurun_out/r4d/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4d/pytest.log
timeout 120 tools/ubench_mfma_dpp.bin > gpurun_out/r4d/ubench_mfma_dpp.log 2>&1
tail -15 gpurun_out/r4d/pytest.log; cat gpurun_out/r4d/ubench_mfma_dpp.log
