mkdir -p gpurun_out/r5c
for sc in D dense E; do
  MS_SPLAT_LIB=tools/abl/libbase.so python tools/rbench.py --scene $sc --save /tmp/ref$sc.pt --tag base > gpurun_out/r5c/base_$sc.txt 2>&1
  python tools/rbench.py --scene $sc --ref /tmp/ref$sc.pt --tag new > gpurun_out/r5c/new_$sc.txt 2>&1
done
MS_SPLAT_LIB=tools/abl/libphases.so python tools/rbench.py --scene D --ref /tmp/refD.pt > gpurun_out/r5c/phases_D.txt 2>&1
MS_SPLAT_LIB=tools/abl/libstats.so python tools/rbench.py --scene D --ref /tmp/refD.pt --iters 2 > gpurun_out/r5c/stats_D.txt 2>&1
for t in 8 32; do
  MS_SPLAT_LIB=tools/abl/libbase.so python tools/rbench.py --scene D --tile $t --save /tmp/refD$t.pt --tag base$t > gpurun_out/r5c/base_D$t.txt 2>&1
  python tools/rbench.py --scene D --tile $t --ref /tmp/refD$t.pt --tag new$t > gpurun_out/r5c/new_D$t.txt 2>&1
done
grep -h RBENCH gpurun_out/r5c/*.txt
timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_explained.py tests/test_gpu_raster.py -x -q -m gpu > gpurun_out/r5c/pytest.txt 2>&1; tail -5 gpurun_out/r5c/pytest.txt
