bash tools/gpu/refresh_r05.sh > gpurun_out/refresh_r05.log 2>&1
bash tools/pmc_phases.sh gpurun_out/r05/pmc_phases > /dev/null 2>&1
find gpurun_out/r05/pmc_phases -name "*.csv" -size +2M -delete 2>/dev/null; find gpurun_out/r05/pmc_phases -name "*agent_info*" -delete
bash tools/gpu/final_r05.sh
