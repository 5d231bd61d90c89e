B="python bench.py --no-cpu-baseline --no-train-step --steps 50"
pick() { grep -E "timed|stages" | sed 's/\[bench [0-9:]*\] //'; }
echo "== config D deterministic"; MS_DETERMINISTIC=1 $B 2>&1 | pick
echo "== config D forward only"; $B --forward-only --no-stages 2>&1 | pick
echo "== config C"; $B --n 1000000 --size 1920 --height 1080 2>&1 | pick
echo "== config B"; $B --n 1000000 --size 1024 --sh-degree 0 2>&1 | pick
echo "== config E frame tile16"; $B --size 4096 --steps 20 2>&1 | pick
echo "== config D tile 8"; $B --tile 8 --steps 20 2>&1 | pick
echo "== config D tile 32"; $B --tile 32 --steps 20 2>&1 | pick
echo "== components"; python -m taichi_splatting_amd.benchmarks rasterizer sh projection tilemapper 2>&1 | tail -40
