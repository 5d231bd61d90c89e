#!/bin/bash
# Build a development variant of libmi355_splat.so into tools/variants/lib<name>.so with extra hipcc flags:
#   tools/build_variant.sh stats -DMS_SCAN_STATS=1
# Select it at run time with MS_SPLAT_LIB=tools/variants/lib<name>.so (taichi_splatting_amd/_lib.py).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/taichi_splatting_amd/csrc
out=$root/tools/variants/obj_$name
mkdir -p "$out"
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -fno-fast-math -fno-slp-vectorize"
pids=()
for f in lib projection sh mapper scan_sort tile_sort raster raster_fast raster_bwd_scan strip_route morton optim gaussian_bwd frame; do
  /opt/rocm/bin/hipcc $flags "$@" -c "$src/$f.hip" -o "$out/$f.o" &
  pids+=($!)
done
case " $* " in *MS_WITH_ROWS_KERNEL*)      # the round-4 experiment, tools/experiments/raster_bwd_rows.hip
  /opt/rocm/bin/hipcc $flags "$@" -I"$src" -c "$root/tools/experiments/raster_bwd_rows.hip" -o "$out/raster_bwd_rows.o" &
  pids+=($!);;
esac
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/variants/lib$name.so" "$out"/*.o
rm -rf "$out"
echo "built tools/variants/lib$name.so"
