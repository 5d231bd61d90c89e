mkdir -p gpurun_out/r5b
tools/ubench_blend.bin > gpurun_out/r5b/ubench_blend.txt 2>&1
python tools/rbench.py --scene D --save /tmp/refD.pt --tag default > gpurun_out/r5b/rbench_default.txt 2>&1
MS_SPLAT_LIB=tools/abl/libphases.so python tools/rbench.py --scene D --ref /tmp/refD.pt > gpurun_out/r5b/rbench_phases.txt 2>&1
MS_SPLAT_LIB=tools/abl/libphases.so python tools/rbench.py --scene dense > gpurun_out/r5b/rbench_phases_dense.txt 2>&1
MS_SPLAT_LIB=tools/abl/libphases.so python tools/rbench.py --scene E > gpurun_out/r5b/rbench_phases_E.txt 2>&1
grep -h RBENCH gpurun_out/r5b/*.txt
cat gpurun_out/r5b/ubench_blend.txt
