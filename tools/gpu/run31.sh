MS_DETERMINISTIC=1 bash tools/trace_frame.sh r5u/det2 > /dev/null 2>&1
head -5 gpurun_out/r5u/det2_trace.txt | cut -c1-160
timeout 600 python -m pytest tests/test_gpu_determinism.py -x -q -m gpu 2>&1 | tail -3
