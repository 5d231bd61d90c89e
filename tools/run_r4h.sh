mkdir -p gpurun_out/r4h
timeout 300 python bench.py --graph-child --steps 100 > gpurun_out/r4h/graph_child.log 2>&1
echo "rc=$?" >> gpurun_out/r4h/graph_child.log
timeout 600 python -m pytest tests/test_gpu_frame.py tests/test_gpu_round4.py tests/test_gpu_sharded_static.py -q -x > gpurun_out/r4h/pytest.log 2>&1
tail -3 gpurun_out/r4h/graph_child.log; tail -3 gpurun_out/r4h/pytest.log
