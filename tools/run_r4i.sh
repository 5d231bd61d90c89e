# round-4 profiles: bench log, kernel trace (--stats), frame timeline, PMC passes for the raster kernels and over the frame
bash tools/refresh_profiles.sh r04 > gpurun_out/r04_refresh.log 2>&1
bash tools/pmc_frame.sh gpurun_out/r04/pmc_frame > gpurun_out/r04_pmc_frame.log 2>&1
python tools/pmc_frame_report.py gpurun_out/r04/pmc_frame > gpurun_out/r04/frame_counters.json 2> gpurun_out/r04/frame_counters.err
ls gpurun_out/r04 gpurun_out/r04/pmc gpurun_out/r04/pmc_frame | head -60
head -30 gpurun_out/r04/kernel_trace.txt
tail -5 gpurun_out/r04/frame_counters.err
