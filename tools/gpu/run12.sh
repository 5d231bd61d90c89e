mkdir -p gpurun_out/r5j
python tools/rbench.py --scene D --save /tmp/refD.pt --tag new > gpurun_out/r5j/new_D.txt 2>&1
for v in prio1 prio2; do
  MS_SPLAT_LIB=tools/abl/lib$v.so python tools/rbench.py --scene D --ref /tmp/refD.pt --tag $v > gpurun_out/r5j/${v}_D.txt 2>&1
done
python tools/rbench.py --scene D --tag new_again > gpurun_out/r5j/new2_D.txt 2>&1
grep -h RBENCH gpurun_out/r5j/*.txt
