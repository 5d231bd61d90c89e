#!/bin/bash
# rocprofv3 PMC passes (one counter group per run, kernel trace only — never combined with other trace domains)
# for the raster kernels on a synthetic scene.  usage: tools/pmc_collect.sh <outdir> [n size tile]
out=${1:-gpurun_out/pmc}; n=${2:-6000000}; size=${3:-2048}; tile=${4:-16}
mkdir -p "$out"
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -o p -- \
    python tools/prof_raster.py $n $size $tile 2 > "$out/$name.log" 2>&1 || echo "pass $name failed (see $out/$name.log)"
}
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass fetch FETCH_SIZE
pass write WRITE_SIZE
python tools/pmc_summary.py "$out" > "$out/summary.json" && cat "$out/summary.json"
