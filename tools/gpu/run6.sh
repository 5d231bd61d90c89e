mkdir -p gpurun_out/r5f
timeout 1500 python tools/sweep_scenes.py --quick --out gpurun_out/r5f/sweep_quick.txt > gpurun_out/r5f/sweep_quick.log 2>&1
tail -25 gpurun_out/r5f/sweep_quick.log
