mkdir -p gpurun_out/r5l
cd /tmp; export TMPDIR=/tmp; cd - >/dev/null
for mode in 0 1; do
  MS_DETERMINISTIC=$mode rocprofv3 --kernel-trace --stats -d gpurun_out/r5l/trace$mode -o t -- python bench.py --steps 15 --warmup 3 --no-cpu-baseline --no-stages --no-graph --no-sweep > gpurun_out/r5l/trace$mode.log 2>&1
  db=$(find gpurun_out/r5l/trace$mode -name '*_results.db' | head -1)
  python tools/rocpd_stats.py "$db" --last-steps 18 > gpurun_out/r5l/kernel_trace_det$mode.txt; rm -f "$db"
  echo "== deterministic=$mode"; head -22 gpurun_out/r5l/kernel_trace_det$mode.txt | cut -c1-200
done
