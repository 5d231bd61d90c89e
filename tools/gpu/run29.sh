for seg in 4096 2048 1024 512; do
  echo "== MS_SPLIT_SEG=$seg"
  MS_SPLIT_SEG=$seg python tools/sweep_scenes.py --only pile 2>&1 | grep "^pile" | cut -c95-330
done
timeout 600 python -m pytest tests/test_gpu_round5.py -x -q -m gpu -k "long_ or cuts_long" 2>&1 | tail -3
