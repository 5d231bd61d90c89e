mkdir -p gpurun_out/r5t
python tools/sweep_scenes.py --only pile --out gpurun_out/r5t/sweep_pile.txt 2>&1 | grep -v SWEEP | tail -8
python tools/sweep_scenes.py --only s8_a0.1_4096 2>&1 | grep -v SWEEP | tail -6
python tools/sweep_scenes.py --only configD 2>&1 | grep -v SWEEP | tail -5
for i in 1 2; do python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-stages --no-sweep 2>&1 | grep -o '"ms_per_step": [0-9.]*'; done
