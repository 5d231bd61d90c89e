bash tools/pmc_phases.sh gpurun_out/r05/pmc_phases 2>&1 | tail -50
find gpurun_out/r05/pmc_phases -name "*.csv" -size +2M -delete 2>/dev/null; find gpurun_out/r05/pmc_phases -name "*agent_info*" -delete
