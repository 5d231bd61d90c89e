set -x
mkdir -p gpurun_out/r5a
cd /root/repo
rocm-smi --showclocks 2>/dev/null | head -20 > gpurun_out/r5a/clocks.txt
tools/ubench_blend.bin > gpurun_out/r5a/ubench_blend.txt 2>&1
python tools/rbench.py --scene D --save /tmp/refD.pt --tag default > gpurun_out/r5a/rbench_default.txt 2>&1
MS_SPLAT_LIB=tools/abl/libphases.so python tools/rbench.py --scene D --ref /tmp/refD.pt > gpurun_out/r5a/rbench_phases.txt 2>&1
MS_SPLAT_LIB=tools/abl/libstats.so python tools/rbench.py --scene D --ref /tmp/refD.pt > gpurun_out/r5a/rbench_stats.txt 2>&1
MS_SPLAT_LIB=tools/abl/libphases.so python tools/rbench.py --scene dense > gpurun_out/r5a/rbench_phases_dense.txt 2>&1
MS_SPLAT_LIB=tools/abl/libstats.so python tools/rbench.py --scene dense > gpurun_out/r5a/rbench_stats_dense.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_projection_sh.py tests/test_gpu_frame.py tests/test_gpu_render.py -x -q -m gpu > gpurun_out/r5a/pytest_subset.txt 2>&1
tail -3 gpurun_out/r5a/pytest_subset.txt
grep RBENCH gpurun_out/r5a/*.txt
cat gpurun_out/r5a/ubench_blend.txt
