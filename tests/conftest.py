"""pytest configuration: ``gpu`` marker, repo on sys.path, shared helpers."""
import ctypes
import os
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
  sys.path.insert(0, str(ROOT))

GOLDEN = Path(__file__).resolve().parent / 'golden'


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: test needs an MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
  if torch.cuda.is_available():
    return
  skip = pytest.mark.skip(reason="no GPU visible")
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)


@pytest.fixture(scope='session')
def lib():
  """The C-ABI library, built in-tree if missing (hipcc cross-compiles without a GPU)."""
  from taichi_splatting_amd import _lib
  if not _lib.LIB_PATH.exists():
    _lib.build()
  return _lib.load()


@pytest.fixture(scope='session')
def hostmath():
  """csrc/splat_math.h compiled for the host with g++ (test infrastructure only)."""
  src = Path(__file__).resolve().parent / 'hostmath' / 'hostmath.cpp'
  out_dir = Path(__file__).resolve().parent / 'hostmath' / '_build'
  out_dir.mkdir(exist_ok=True)
  so = out_dir / 'libhostmath.so'
  header = ROOT / 'taichi_splatting_amd' / 'csrc' / 'splat_math.h'
  if not so.exists() or so.stat().st_mtime < max(src.stat().st_mtime, header.stat().st_mtime):
    subprocess.run(['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-ffp-contract=off', str(src), '-o', str(so)],
                   check=True)
  return ctypes.CDLL(str(so))


def load_golden(name):
  return torch.load(GOLDEN / name, weights_only=False)


def covariance_of(points):
  """(V, 3) [a, b, c] of the 2D covariance [[a, b], [b, c]] = s1^2 u u^T + s2^2 w w^T rebuilt from a packed row's
  (axis u, sigma): the well-conditioned content of the four columns — the axis of a nearly isotropic splat is not."""
  p = torch.as_tensor(points).double()
  ax, ay, s1, s2 = p[:, 2], p[:, 3], p[:, 4] ** 2, p[:, 5] ** 2
  return torch.stack([s1 * ax * ax + s2 * ay * ay, (s1 - s2) * ax * ay, s1 * ay * ay + s2 * ax * ax], 1)
