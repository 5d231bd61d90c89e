mkdir -p gpurun_out/r4b
cd /root/repo
(timeout 300 python tools/ab_raster_bwd.py 2>&1 | tail -25) > gpurun_out/r4b/ab_rows_16.log
(MS_RASTER_BWD=scan timeout 300 python tools/ab_raster_bwd.py 2>&1 | tail -8) > gpurun_out/r4b/ab_scan_16.log
(timeout 300 python tools/ab_raster_bwd.py --tile 8 2>&1 | tail -25) > gpurun_out/r4b/ab_rows_8.log
(MS_RASTER_BWD=scan timeout 300 python tools/ab_raster_bwd.py --tile 8 2>&1 | tail -8) > gpurun_out/r4b/ab_scan_8.log
(timeout 300 python tools/ab_raster_bwd.py --dense 2>&1 | tail -25) > gpurun_out/r4b/ab_rows_dense.log
(MS_RASTER_BWD=scan timeout 300 python tools/ab_raster_bwd.py --dense 2>&1 | tail -8) > gpurun_out/r4b/ab_scan_dense.log
(timeout 300 python tools/ab_raster_bwd.py --heur 2>&1 | tail -25) > gpurun_out/r4b/ab_rows_heur.log
(timeout 300 bash tools/build_variant.sh stats -DMS_SCAN_STATS=1 2>&1 | tail -2; MS_SPLAT_LIB=tools/abl/libstats.so timeout 300 python tools/ab_raster_bwd.py 2>&1 | tail -25) > gpurun_out/r4b/ab_stats.log
timeout 900 python -m pytest tests/test_gpu_round4.py tests/test_gpu_fuzz.py tests/test_gpu_explained.py tests/test_gpu_projection_sh.py tests/test_gpu_determinism.py -q -x 2>&1 | tail -30 > gpurun_out/r4b/pytest.log
tail -5 gpurun_out/r4b/ab_rows_16.log gpurun_out/r4b/pytest.log
