"""Randomised cross-check of the product raster kernels on 2 x 150 random scenes (tools/fuzz_raster_bwd.py): the
float32 forward with and without visibility against the float64 generic forward (and against each other), the
splat-per-lane scan backward (plain and deterministic) against the pixel-per-lane backward — image sizes that are
not tile multiples, tile 8 / 16 / 32, tiny and huge splats, thresholds, strips of tile rows, heuristics."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.gpu
@pytest.mark.parametrize("first", [0, 1000])
def test_random_scenes_forward_and_both_backwards_agree(first):
  proc = subprocess.run([sys.executable, str(ROOT / 'tools' / 'fuzz_raster_bwd.py'), '--seeds', '150', '--first', str(first)],
                        capture_output=True, text=True, timeout=900)
  assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
  assert 'failures 0' in proc.stdout.splitlines()[-1]
