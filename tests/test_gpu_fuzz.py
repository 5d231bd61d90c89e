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


@pytest.mark.gpu
def test_random_tile_runs_sort_like_a_stable_composite_sort():
  # tools/fuzz_tile_sort.py: ms_tile_depth_sort on random run lengths (every size class, the cost fallback, the global
  # radix path) and key shapes (bits, floats, binades, clusters, ties, denormal / inf / NaN patterns, 16 bit keys)
  proc = subprocess.run([sys.executable, str(ROOT / 'tools' / 'fuzz_tile_sort.py'), '--rounds', '120', '--seed', '3'],
                        capture_output=True, text=True, timeout=600)
  assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-2000:]
  assert 'rounds identical' in proc.stdout.splitlines()[-1]
