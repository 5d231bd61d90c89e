timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_raster.py tests/test_gpu_frame.py tests/test_gpu_render.py tests/test_gpu_tile_sort.py -x -q -m gpu 2>&1 | tail -4
python tools/sweep_scenes.py --only pile 2>&1 | grep "^pile" | cut -c95-330
