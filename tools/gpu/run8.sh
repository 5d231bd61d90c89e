mkdir -p gpurun_out/r5h
python tools/rbench.py --scene D --tag fwdexit > gpurun_out/r5h/rbench_D.txt 2>&1; grep RBENCH gpurun_out/r5h/rbench_D.txt
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_render.py tests/test_gpu_frame.py tests/test_gpu_configs.py -x -q -m gpu -k "not full_size and not fullsize" > gpurun_out/r5h/pytest.txt 2>&1; tail -5 gpurun_out/r5h/pytest.txt
timeout 1500 python tools/sweep_scenes.py --out gpurun_out/r5h/sweep.txt > gpurun_out/r5h/sweep.log 2>&1
grep -v "^SWEEP" gpurun_out/r5h/sweep.log | tail -42
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5h/smoke.txt 2>&1; tail -2 gpurun_out/r5h/smoke.txt
