#!/bin/bash
# Everything profiles/ holds for round 5, from one GPU session: tools/gpu/refresh_r05.sh  -> gpurun_out/r05/
out=gpurun_out/r05
mkdir -p $out
tools/ubench_blend.bin > $out/ubench_blend.txt 2>&1
python tools/rbench.py --scene D --save /tmp/refD.pt --tag product > $out/rbench_product.txt 2>&1
for v in stats phases abl1 abl2 commit1 commit2 commit3; do
  it=20; [ $v = stats ] && it=2
  MS_SPLAT_LIB=tools/variants/lib$v.so python tools/rbench.py --scene D --ref /tmp/refD.pt --iters $it --tag $v > $out/rbench_$v.txt 2>&1
done
python tools/work_counters.py $out > $out/work.json
tools/refresh_profiles.sh r05
timeout 900 python tools/sweep_scenes.py --out $out/scene_sweep.txt > $out/scene_sweep.log 2>&1
tail -3 $out/bench.log | cut -c1-1500
grep -h RBENCH $out/rbench_*.txt | cut -c1-400
