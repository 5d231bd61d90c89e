#!/bin/bash
# SQ_INSTS_VALU / SALU / LDS of the raster backward for several development variants (tools/variants/lib<name>.so):
#   tools/pmc_valu_variants.sh <outdir> name1 name2 ...
out=$1; shift
mkdir -p "$out"
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
for name in "$@"; do
  lib=""; [ "$name" != default ] && lib=tools/variants/lib$name.so
  MS_SPLAT_LIB=$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv \
    -d "$out/$name" -o p -- python tools/prof_raster.py 6000000 2048 16 2 > "$out/$name.log" 2>&1 || echo "$name failed"
  python - "$out/$name" "$name" <<'PY'
import csv, glob, sys, collections
d, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    if 'raster_bwd_scan' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k, v in acc.items():
  print(name, k[:50], {c: round(x / cnt[(k, c)] / 1e6, 1) for c, x in v.items()}, 'M per launch')
PY
done
