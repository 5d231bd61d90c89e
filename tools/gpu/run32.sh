bash tools/gpu/refresh_r05.sh > gpurun_out/refresh_r05.log 2>&1
# the condensed counter file of THIS tree, so that the bench line below attaches it (same fingerprint)
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-fast-math -fno-slp-vectorize -S --cuda-device-only"
hipcc $F taichi_splatting_amd/csrc/raster_bwd_scan.hip -o /tmp/bwd.s 2>/dev/null
hipcc $F taichi_splatting_amd/csrc/raster_fast.hip -o /tmp/fwd.s 2>/dev/null
python tools/pmc_to_profile.py gpurun_out/r05/pmc/summary.json 6000000 2048 16 12760306 /tmp/bwd.s /tmp/fwd.s --work gpurun_out/r05/work.json > /tmp/counters.json \
  && cp /tmp/counters.json profiles/raster_bwd_counters.json && cp /tmp/counters.json gpurun_out/r05/raster_bwd_counters.json
bash tools/pmc_phases.sh gpurun_out/r05/pmc_phases > /dev/null 2>&1
find gpurun_out/r05/pmc_phases -name "*.csv" -size +2M -delete 2>/dev/null; find gpurun_out/r05/pmc_phases -name "*agent_info*" -delete
bash tools/gpu/final_r05.sh
