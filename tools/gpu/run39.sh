mkdir -p gpurun_out/r05f gpurun_out/r05
timeout 900 python -m pytest tests/test_gpu_raster.py tests/test_gpu_round5.py tests/test_gpu_frame.py tests/test_gpu_explained.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu/refresh_r05.sh > gpurun_out/refresh_r05.log 2>&1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize -S --cuda-device-only"
hipcc $F taichi_splatting_amd/csrc/raster_bwd_scan.hip -o /tmp/bwd.s 2>/dev/null
hipcc $F taichi_splatting_amd/csrc/raster_fast.hip -o /tmp/fwd.s 2>/dev/null
python tools/pmc_to_profile.py gpurun_out/r05/pmc/summary.json 6000000 2048 16 12760306 /tmp/bwd.s /tmp/fwd.s --work gpurun_out/r05/work.json > /tmp/counters.json \
  && cp /tmp/counters.json profiles/raster_bwd_counters.json && cp /tmp/counters.json gpurun_out/r05/raster_bwd_counters.json
bash tools/trace_frame.sh r05/final > /dev/null 2>&1
head -8 gpurun_out/r05/final_trace.txt | cut -c1-140
python bench.py > gpurun_out/r05f/bench_counters.log 2>&1
grep "^\[bench" gpurun_out/r05f/bench_counters.log | cut -c1-100 | sed -n 3,6p
grep -h RBENCH gpurun_out/r05/rbench_product.txt | cut -c1-300
