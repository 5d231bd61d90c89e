#!/bin/bash
# Round 6, third GPU session: forward hit walk in pairs (A/B on one box, interleaved), optimiser variants, the extended
# scene sweep, the tests touched since session B, a gather micro-benchmark for the deterministic-mode costing.
out=gpurun_out/r06c
mkdir -p $out
python tools/rbench.py --scene D --save /tmp/refD.pt --tag pair_a > $out/rbench_pair_a.txt 2>&1
for v in fwdnopair fwdbase; do
  MS_SPLAT_LIB=tools/variants/lib$v.so python tools/rbench.py --scene D --ref /tmp/refD.pt --iters 20 --tag $v > $out/rbench_$v.txt 2>&1
done
python tools/rbench.py --scene D --ref /tmp/refD.pt --tag pair_b > $out/rbench_pair_b.txt 2>&1
MS_SPLAT_LIB=tools/variants/libfwdnopair.so python tools/rbench.py --scene D --ref /tmp/refD.pt --iters 20 --tag fwdnopair_b > $out/rbench_fwdnopair_b.txt 2>&1
MS_SPLAT_LIB=tools/variants/libfwdphases.so python tools/rbench.py --scene D --ref /tmp/refD.pt --iters 20 --tag fwdphases > $out/rbench_fwdphases.txt 2>&1
for t in 8 32; do
  python tools/rbench.py --scene D --tile $t --tag pair_t$t > $out/rbench_pair_t$t.txt 2>&1
  MS_SPLAT_LIB=tools/variants/libfwdnopair.so python tools/rbench.py --scene D --tile $t --tag nopair_t$t > $out/rbench_nopair_t$t.txt 2>&1
done
grep -h RBENCH $out/rbench_*.txt | cut -c1-1300
( time timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_tile_sort.py tests/test_gpu_round5.py tests/test_gpu_frame.py tests/test_gpu_raster.py tests/test_optim.py -x -q --durations=10 ) > $out/pytest_c.txt 2>&1
tail -18 $out/pytest_c.txt
for mode in "fused" "MS_OPTIM_FUSED=0" "MS_SPLAT_LIB=tools/variants/liboptimnt.so"; do
  if [ "$mode" = fused ]; then env_=""; else env_="$mode"; fi
  env $env_ timeout 600 python bench.py --train-step --steps 10 > $out/train_step_${mode//[^a-zA-Z0-9]/_}.log 2>&1
  echo "$mode: $(tail -1 $out/train_step_${mode//[^a-zA-Z0-9]/_}.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); t=d["extra"]["train_step"]; print(t["optimizer"], t["iteration_ms_reference_loop"], t["iteration_ms_dense_step"], t["render_backward_ms"])')"
done
python - > $out/gather_ubench.txt 2>&1 <<'PY'
import torch
dev='cuda:0'
k=12_760_306
table=torch.rand(k,16,device=dev)
perm=torch.randperm(k,device=dev)
def t(fn,it=5):
  fn(); torch.cuda.synchronize()
  a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
  a.record()
  for _ in range(it): fn()
  b.record(); torch.cuda.synchronize()
  return a.elapsed_time(b)/it
out=torch.empty_like(table)
print("gather 64-byte rows by a random permutation (K = 12.76 M, 817 MB in, 817 MB out):", round(t(lambda: torch.index_select(table,0,perm,out=out)),3),"ms")
print("plain copy of the same table:", round(t(lambda: out.copy_(table)),3),"ms")
inv=torch.empty_like(perm)
print("inverse permutation (scatter of K int64):", round(t(lambda: inv.scatter_(0,perm,torch.arange(k,device=dev))),3),"ms")
PY
cat $out/gather_ubench.txt
timeout 1500 python tools/sweep_scenes.py --out $out/scene_sweep.txt > $out/scene_sweep.log 2>&1
grep -v "^SWEEP" $out/scene_sweep.txt | cut -c1-330 | tail -60
