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
  assert lib.ms_version() == 500     # MS_VERSION of include/mi355_splat.h (0.5.0: sized structs, run-time split parameters)
  assert lib.ms_version() == _lib.ABI_VERSION
  assert re.search(r'#define MS_VERSION (\d+)', HEADER.read_text()).group(1) == '500'
  import taichi_splatting_amd
  assert taichi_splatting_amd.__version__ == '0.5.0'
  assert isinstance(lib.ms_last_error_string(), bytes)


def header_struct_sizes(tmp_path):
  """sizeof / offsetof of the header's structs as a C compiler sees them (gcc on include/mi355_splat.h)"""
  import subprocess
  src = tmp_path / 'sizes.c'
  src.write_text("""
    #include <stdio.h>
    #include <stddef.h>
    #include "mi355_splat.h"
    #define S(T) printf(#T " %zu\\n", sizeof(T))
    #define O(T, f) printf(#T "." #f " %zu\\n", offsetof(T, f))
    int main(void) {
      S(ms_raster_config); S(ms_frame_desc); S(ms_frame_layout); S(ms_frame_inputs); S(ms_frame_grads);
      O(ms_frame_desc, struct_size); O(ms_frame_desc, abi_version); O(ms_frame_desc, n); O(ms_frame_desc, split_long_runs);
      O(ms_frame_desc, split_seg_len); O(ms_frame_desc, near_plane); O(ms_frame_desc, raster);
      O(ms_frame_layout, splat_rows); O(ms_frame_layout, split_scratch);
      O(ms_frame_inputs, position); O(ms_frame_inputs, longest_run_host);
      O(ms_frame_grads, image); O(ms_frame_grads, stage); O(ms_frame_grads, boundary_form); O(ms_frame_grads, grad_image_broadcast);
      return 0;
    }
  """)
  exe = tmp_path / 'sizes'
  subprocess.run(['gcc', '-std=c99', '-I', str(HEADER.parent), str(src), '-o', str(exe)], check=True)
  out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
  return {k: int(v) for k, v in (line.split() for line in out.strip().splitlines())}


def test_struct_layouts_match_the_header(tmp_path):
  """The ctypes mirrors in _lib.py against the C header itself: total sizes and the offsets of the fields that moved in
  rounds 5 and 6 (two structs had grown under an unchanged MS_VERSION)."""
  c = header_struct_sizes(tmp_path)
  for cname, py in (('ms_raster_config', _lib.RasterConfigC), ('ms_frame_desc', _lib.FrameDescC),
                    ('ms_frame_layout', _lib.FrameLayoutC), ('ms_frame_inputs', _lib.FrameInputsC),
                    ('ms_frame_grads', _lib.FrameGradsC)):
    assert ctypes.sizeof(py) == c[cname], (cname, ctypes.sizeof(py), c[cname])
  for key, want in c.items():
    if '.' in key:
      cname, field = key.split('.')
      py = {'ms_frame_desc': _lib.FrameDescC, 'ms_frame_layout': _lib.FrameLayoutC, 'ms_frame_inputs': _lib.FrameInputsC,
            'ms_frame_grads': _lib.FrameGradsC}[cname]
      assert getattr(py, field).offset == want, (key, getattr(py, field).offset, want)
  d = _lib.FrameDescC(n=1)
  assert d.struct_size == c['ms_frame_desc'] and d.abi_version == 500
  assert _lib.FrameInputsC().struct_size == c['ms_frame_inputs'] and _lib.FrameGradsC().struct_size == c['ms_frame_grads']


def frame_desc(**kw):
  args = dict(n=1000, k_capacity=5000, image_w=256, image_h=128, dtype=_lib.MS_F32, f=3, sh_degree=3, depth16=0,
              tile_row_begin=0, tile_row_end=1 << 30, projected_input=0, mapper=0, near_plane=0.1, far_plane=100.0,
              blur_cov=0.3, clamp_margin=0.15,
              raster=_lib.RasterConfigC(tile_size=16, antialias=0, use_alpha_blending=1, compute_visibility=0,
                                        compute_point_heuristic=0, reserved=0, clamp_max_alpha=0.99,
                                        alpha_threshold=1 / 255., saturate_threshold=0.9999))
  args.update(kw)
  return _lib.FrameDescC(**args)


def test_frame_calls_reject_structs_of_another_abi(lib):
  """A caller compiled against another header passes structs whose fields are not where this library reads them: every
  ms_frame_* entry point checks the leading struct_size / abi_version and returns MS_ERR_ABI (-4) before touching
  anything else (no GPU needed: the checks come first)."""
  lay = _lib.FrameLayoutC()
  d = frame_desc()
  assert lib.ms_frame_layout_query(ctypes.byref(d), ctypes.byref(lay)) == 0 and lay.keep_n_bytes > 0
  for field, value in (('struct_size', ctypes.sizeof(d) - 8), ('struct_size', 0), ('abi_version', 400), ('abi_version', 0),
                       ('abi_version', 600)):
    bad = frame_desc()
    setattr(bad, field, value)
    assert lib.ms_frame_layout_query(ctypes.byref(bad), ctypes.byref(lay)) == -4, (field, value)
    assert b'ABI' in lib.ms_last_error_string()
    for call in (lambda: lib.ms_frame_project(ctypes.byref(bad), ctypes.byref(_lib.FrameInputsC()), 1, None),
                 lambda: lib.ms_frame_project_count(ctypes.byref(bad), ctypes.byref(_lib.FrameInputsC()), 1, 1, None, None, None),
                 lambda: lib.ms_frame_map_raster(ctypes.byref(bad), ctypes.byref(_lib.FrameInputsC()), 1, 1, 1, 1, 1, 1, None, None),
                 lambda: lib.ms_frame_backward(ctypes.byref(bad), ctypes.byref(_lib.FrameInputsC()), 1, 1, ctypes.byref(_lib.FrameGradsC()), None)):
      assert call() == -4
  ok_minor = frame_desc()
  ok_minor.abi_version = 507                  # same generation: accepted
  assert lib.ms_frame_layout_query(ctypes.byref(ok_minor), ctypes.byref(lay)) == 0
  # the other two structs
  inputs, grads = _lib.FrameInputsC(), _lib.FrameGradsC()
  inputs.struct_size -= 8
  assert lib.ms_frame_project(ctypes.byref(d), ctypes.byref(inputs), 1, None) == -4
  assert b'ms_frame_inputs' in lib.ms_last_error_string()
  grads.struct_size = 0
  assert lib.ms_frame_backward(ctypes.byref(d), ctypes.byref(_lib.FrameInputsC()), 1, 1, ctypes.byref(grads), None) == -4
  assert b'ms_frame_grads' in lib.ms_last_error_string()
  with pytest.raises(RuntimeError, match="ABI mismatch"):
    _lib.check(-4, "frame")


def test_split_parameters_size_the_scratch(lib):
  """ms_raster_split_scratch_bytes with run-time (min run, segment length): smaller thresholds need more plan / state
  rows; 0 = the defaults; invalid arguments give 0 / MS_ERR_BAD_ARG (host-side checks, no GPU)."""
  k = 1 << 20
  default = lib.ms_raster_split_scratch_bytes(k, 16, 0, 0)
  assert default == lib.ms_raster_split_scratch_bytes(k, 16, 16384, 1024) > 0
  assert lib.ms_raster_split_scratch_bytes(k, 16, 512, 256) > default
  assert lib.ms_raster_split_scratch_bytes(k, 16, 100, 100) == lib.ms_raster_split_scratch_bytes(k, 16, 256, 256)
  assert lib.ms_raster_split_scratch_bytes(k, 12, 0, 0) == 0 and lib.ms_raster_split_scratch_bytes(-1, 16, 0, 0) == 0
  assert lib.ms_raster_split_scratch_bytes(k, 16, -5, 0) == 0
  cfg = _lib.RasterConfigC(tile_size=16, antialias=0, use_alpha_blending=1, compute_visibility=0, compute_point_heuristic=0,
                           reserved=0, clamp_max_alpha=0.99, alpha_threshold=1 / 255., saturate_threshold=0.9999)
  # (the argument checks come before any launch)
  assert lib.ms_raster_bwd_moments_split(1, 1, 1, 1, -1, 1, 1, 64, 64, ctypes.byref(cfg), 1, 0, None, 256, 0, 0, 0, 4, None) == -1
  assert b'k_capacity' in lib.ms_last_error_string()
  assert lib.ms_raster_bwd_moments_split(1, 1, 1, 1, 100, 1, 1, 64, 64, ctypes.byref(cfg), 1, 0, None, 256, -1, 0, 0, 4, None) == -1
  assert lib.ms_raster_bwd_moments_split(1, 1, 1, 1, 100, 1, 1, 64, 64, ctypes.byref(cfg), 1, 0, None, 100, 0, 0, 0, 4, None) == -1
  assert b'aligned' in lib.ms_last_error_string()
  cfg.tile_size = 12
  assert lib.ms_raster_bwd_moments_split(1, 1, 1, 1, 100, 1, 1, 64, 64, ctypes.byref(cfg), 1, 0, None, 256, 0, 0, 0, 4, None) == -2
  d = frame_desc(split_long_runs=1)
  d2 = frame_desc(split_long_runs=512, split_seg_len=256)
  la, lb = _lib.FrameLayoutC(), _lib.FrameLayoutC()
  assert lib.ms_frame_layout_query(ctypes.byref(d), ctypes.byref(la)) == 0 and lib.ms_frame_layout_query(ctypes.byref(d2), ctypes.byref(lb)) == 0
  assert lb.keep_k_bytes > la.keep_k_bytes
  assert lib.ms_frame_layout_query(ctypes.byref(frame_desc(split_long_runs=-1)), ctypes.byref(la)) == -1


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
