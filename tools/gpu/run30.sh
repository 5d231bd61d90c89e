MS_DETERMINISTIC=1 bash tools/trace_frame.sh r5u/det > /dev/null 2>&1
head -16 gpurun_out/r5u/det_trace.txt | cut -c1-160
