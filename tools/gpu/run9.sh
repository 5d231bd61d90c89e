mkdir -p gpurun_out/r5i
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_raster.py -x -q -m gpu > gpurun_out/r5i/pytest.txt 2>&1; tail -15 gpurun_out/r5i/pytest.txt
python tools/rbench.py --scene D --tag fwdexit > gpurun_out/r5i/rbench_D.txt 2>&1; grep RBENCH gpurun_out/r5i/rbench_D.txt
timeout 600 python tools/sweep_scenes.py --only pile --out gpurun_out/r5i/sweep_pile.txt > gpurun_out/r5i/sweep_pile.log 2>&1; grep -v "^SWEEP" gpurun_out/r5i/sweep_pile.log | tail -8
