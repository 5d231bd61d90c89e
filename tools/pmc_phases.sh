#!/bin/bash
# Counter split of the raster backward BY PHASE: the SQ counters of the product kernel next to
# those of the builds that leave phases out (tools/build_variant.sh):  abl2 = staging only (-DMS_SCAN_ABLATE=2),
# abl1 = everything but the blend (=1), commit3 = everything but the commit's global traffic (-DMS_COMMIT_ABLATE=3).
# One counter group per run, kernel trace only.   tools/pmc_phases.sh <outdir>  ->  <outdir>/phases.txt
out=${1:-gpurun_out/pmc_phases}
mkdir -p "$out"
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
for name in default abl2 abl1 commit3; do
  lib=""; [ "$name" != default ] && lib=tools/variants/lib$name.so
  pass() {
    p=$1; shift
    MS_SPLAT_LIB=$lib rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name/$p" -o p -- \
      python tools/prof_raster.py 6000000 2048 16 2 > "$out/$name.$p.log" 2>&1 || echo "$name pass $p failed (see $out/$name.$p.log)"
  }
  pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
  pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  pass sq3 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM
  pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
done
python tools/pmc_phases_report.py "$out" > "$out/phases.txt"; cat "$out/phases.txt"
