#!/bin/bash
# Round 6, second GPU session: forward-kernel variants on one box, multi-GPU emulation with modelled link time,
# optimiser counters, remaining new tests.  -> gpurun_out/r06b/
out=gpurun_out/r06b
mkdir -p $out
python tools/rbench.py --scene D --save /tmp/refD.pt --tag product > $out/rbench_product.txt 2>&1
for v in fwdbase fwdskip fwdphases fwdbasephases; do
  MS_SPLAT_LIB=tools/variants/lib$v.so python tools/rbench.py --scene D --ref /tmp/refD.pt --iters 20 --tag $v > $out/rbench_$v.txt 2>&1
done
python tools/rbench.py --scene D --ref /tmp/refD.pt --tag product2 > $out/rbench_product2.txt 2>&1
grep -h RBENCH $out/rbench_*.txt | cut -c1-1200
( time timeout 600 python -m pytest tests/test_optim.py tests/test_gpu_round6.py tests/test_gpu_sharded_static.py tests/test_gpu_sharded.py tests/test_gpu_multi.py -x -q --durations=10 ) > $out/pytest_b.txt 2>&1
tail -20 $out/pytest_b.txt
timeout 900 python tools/emulate_sharded.py --static --world 8 --size 4096 --steps 20 --out $out/emul_sharded_8_4096.json > $out/emul_sharded_8_4096.log 2>&1
tail -3 $out/emul_sharded_8_4096.log | cut -c1-3000
