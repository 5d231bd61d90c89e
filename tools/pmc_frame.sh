#!/bin/bash
# rocprofv3 PMC passes over EVERY kernel of a config-D frame (one counter group per run, kernel trace only — never
# combined with other trace domains) plus the FETCH_SIZE / WRITE_SIZE calibration kernels of tools/ubench_fetch.hip.
# usage: tools/pmc_frame.sh <outdir> [n size tile];  then  python tools/pmc_frame_report.py <outdir> > profiles/r04_frame_counters.json
out=${1:-gpurun_out/pmc_frame}; n=${2:-6000000}; size=${3:-2048}; tile=${4:-16}
mkdir -p "$out"
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
[ -x tools/ubench_fetch.bin ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench_fetch.hip -o tools/ubench_fetch.bin 2>/dev/null
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/$name" -o p -- \
    python tools/prof_frame.py $n $size $tile 3 > "$out/$name.log" 2>&1 || echo "pass $name failed (see $out/$name.log)"
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/calib_$name" -o p -- \
    tools/ubench_fetch.bin > "$out/calib_$name.log" 2>&1 || echo "calibration pass $name failed"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
grep -h "^N=" "$out"/fetch.log | head -1 > "$out/workload.txt"
grep -h "^{" "$out"/calib_fetch.log | head -1 > "$out/calibration_bytes.json"
python tools/pmc_summary.py "$out" "" > "$out/summary.json"
