mkdir -p gpurun_out/r4e
timeout 1500 python -m pytest tests/test_gpu_frame.py tests/test_gpu_sharded_static.py tests/test_gpu_sharded.py tests/test_gpu_multi.py tests/test_gpu_strips.py tests/test_gpu_raster.py tests/test_gpu_render.py tests/test_gpu_explained.py -q --durations=10 > gpurun_out/r4e/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4e/pytest.log
for w in 2 4 8; do
  timeout 600 python tools/emulate_sharded.py --strips --world $w --out gpurun_out/r4e/emul_strips_${w}_2048.json > /dev/null 2> gpurun_out/r4e/emul_strips_${w}_2048.err
  timeout 600 python tools/emulate_sharded.py --static --world $w --out gpurun_out/r4e/emul_sharded_${w}_2048.json > /dev/null 2> gpurun_out/r4e/emul_sharded_${w}_2048.err
done
for w in 2 4 8; do
  timeout 600 python tools/emulate_sharded.py --strips --world $w --size 4096 --out gpurun_out/r4e/emul_strips_${w}_4096.json > /dev/null 2> gpurun_out/r4e/emul_strips_${w}_4096.err
  timeout 600 python tools/emulate_sharded.py --static --world $w --size 4096 --out gpurun_out/r4e/emul_sharded_${w}_4096.json > /dev/null 2> gpurun_out/r4e/emul_sharded_${w}_4096.err
done
tail -8 gpurun_out/r4e/pytest.log; cat gpurun_out/r4e/emul_*_8_*.json | cut -c1-400
