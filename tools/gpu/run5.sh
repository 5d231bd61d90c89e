mkdir -p gpurun_out/r5e
python tools/rbench.py --scene D --save /tmp/refD.pt --tag new > gpurun_out/r5e/new_D.txt 2>&1
for v in commit1 commit2 commit3 abl1 abl2; do
  MS_SPLAT_LIB=tools/abl/lib$v.so python tools/rbench.py --scene D --tag $v > gpurun_out/r5e/${v}_D.txt 2>&1
done
python tools/rbench.py --scene D --tag new_again > gpurun_out/r5e/new2_D.txt 2>&1
grep -h RBENCH gpurun_out/r5e/*.txt
