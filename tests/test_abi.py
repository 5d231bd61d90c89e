"""The C-ABI library loads, exports every function include/mi355_splat.h declares, and validates
its arguments (no compute: runs without a GPU)."""
import ctypes
import re
from pathlib import Path

import pytest

from taichi_splatting_amd import _lib

HEADER = Path(__file__).resolve().parent.parent / 'include' / 'mi355_splat.h'


def declared_functions():
  text = re.sub(r'/\*.*?\*/', '', HEADER.read_text(), flags=re.S)
  return sorted(set(re.findall(r'\b(ms_[a-z0-9_]+)\s*\(', text)))


def test_header_declares_functions():
  names = declared_functions()
  assert 'ms_raster_fwd' in names and 'ms_radix_sort_pairs' in names and len(names) >= 15


def test_every_declared_symbol_is_exported(lib):
  for name in declared_functions():
    assert hasattr(lib, name), f"{name} declared in mi355_splat.h but not exported"
  assert set(_lib.SIGNATURES) == set(declared_functions())


def test_version_and_error_string(lib):
  assert lib.ms_version() == 400     # MS_VERSION of include/mi355_splat.h (0.4.0: mapper selector, per-tile depth sort)
  assert isinstance(lib.ms_last_error_string(), bytes)


def test_scratch_size_queries(lib):
  n = ctypes.c_size_t(0)
  assert lib.ms_exclusive_scan_i32(None, 1000000, None, None, None, ctypes.byref(n), None) == 0
  assert n.value >= 4 * ((1000000 + 4095) // 4096)
  assert lib.ms_radix_sort_pairs(None, None, None, None, 1 << 20, 8, 0, 48, None, ctypes.byref(n), None) == 0
  assert n.value >= (1 << 20) * 12


def test_argument_errors_are_reported(lib):
  n = ctypes.c_size_t(0)
  assert lib.ms_radix_sort_pairs(None, None, None, None, 10, 3, 0, 8, None, ctypes.byref(n), None) == -1
  assert b'key_bytes' in lib.ms_last_error_string()
  assert lib.ms_radix_sort_pairs(None, None, None, None, 10, 4, 0, 40, None, ctypes.byref(n), None) == -1
  assert lib.ms_sh_fwd(None, None, None, None, 10, 3, 7, None, 0, None) == -1
  assert b'degree' in lib.ms_last_error_string()
  cfg = _lib.RasterConfigC(tile_size=12, antialias=0, use_alpha_blending=1, compute_visibility=0,
                           compute_point_heuristic=0, reserved=0, clamp_max_alpha=0.99,
                           alpha_threshold=1 / 255., saturate_threshold=0.9999)
  assert lib.ms_raster_fwd(None, None, None, None, 64, 64, 3, ctypes.byref(cfg), None, None, None, 0, 4, 0, None) == -2
  with pytest.raises(NotImplementedError):
    _lib.check(-2, "raster")
  with pytest.raises(ValueError):
    _lib.check(-1, "x")


def test_product_has_no_cpu_fallback():
  import torch
  from taichi_splatting_amd import RasterConfig, rasterize
  g = torch.zeros((4, 7))
  with pytest.raises(RuntimeError, match="no CPU fallback"):
    rasterize(g, torch.zeros((4, 1)), torch.zeros((4, 3)), (32, 32), RasterConfig())


def test_bench_gpus_flag_fails_loudly_without_the_devices():
  """`bench.py --gpus N` must launch N ranks itself or refuse: it may never fall back to a 1-GPU number
  (this container has no GPU, so the request cannot be met)."""
  import subprocess, sys
  import torch
  if torch.cuda.device_count() >= 2:
    pytest.skip("node has the devices; covered by tests/test_gpu_multi.py")
  root = Path(__file__).resolve().parent.parent
  proc = subprocess.run([sys.executable, str(root / 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                        capture_output=True, text=True, timeout=300)
  assert proc.returncode == 2, proc.stderr[-2000:]
  assert '--gpus 2 requested' in proc.stderr
  assert proc.stdout.strip() == ''
