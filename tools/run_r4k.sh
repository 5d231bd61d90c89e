mkdir -p gpurun_out/r4k
(timeout 300 python tools/ab_raster_bwd.py 2>&1 | tail -14) > gpurun_out/r4k/ab16.log
(timeout 300 python tools/ab_raster_bwd.py --tile 8 2>&1 | tail -4) > gpurun_out/r4k/ab8.log
(timeout 300 python tools/ab_raster_bwd.py --dense 2>&1 | tail -4) > gpurun_out/r4k/abdense.log
timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_explained.py tests/test_gpu_configs.py tests/test_gpu_raster.py tests/test_gpu_determinism.py tests/test_gpu_frame.py -q -x > gpurun_out/r4k/pytest.log 2>&1
echo "rc=$?" >> gpurun_out/r4k/pytest.log
bash tools/pmc_collect.sh gpurun_out/r4k/pmc > gpurun_out/r4k/pmc.log 2>&1
cat gpurun_out/r4k/ab16.log; tail -3 gpurun_out/r4k/ab8.log gpurun_out/r4k/abdense.log; tail -3 gpurun_out/r4k/pytest.log
