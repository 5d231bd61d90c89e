"""ctypes binding of ``libmi355_splat.so`` (C-ABI declared in ``include/mi355_splat.h``).

The library is the product: there is no CPU or eager-PyTorch fallback.  ``load()`` raises
``RuntimeError`` if the shared object has not been built (``python __graft_entry__.py`` or
``make -C taichi_splatting_amd/csrc``), and every wrapper raises if a tensor is not on a GPU.
"""
from __future__ import annotations

import ctypes
from ctypes import c_int, c_int32, c_int64, c_double, c_float, c_void_p, c_size_t, c_char_p, POINTER
import os
from pathlib import Path
import subprocess
from typing import Optional

import torch

PACKAGE_DIR = Path(__file__).resolve().parent
CSRC_DIR = PACKAGE_DIR / 'csrc'
LIB_PATH = Path(os.environ.get('MS_SPLAT_LIB') or PACKAGE_DIR / 'libmi355_splat.so')   # env override: profiling builds

MS_F32, MS_F64 = 0, 1
BACKWARD_ALL, BACKWARD_GAUSSIANS, BACKWARD_RASTER = 0, 1, 2   # ms_frame_grads.stage
BOUNDARY_AXIS_SIGMA, BOUNDARY_COVARIANCE = 0, 1
MAPPER_DIRECT, MAPPER_PRESORT = 0, 1               # ms_frame_grads.boundary_form
ABI_VERSION = 500  # MS_VERSION of include/mi355_splat.h this binding was written against (tests/test_abi.py)
MOMENT_ROW = 16   # MS_MOMENT_ROW of include/mi355_splat.h
SPLAT_ROW = 16    # MS_SPLAT_ROW

_lib: Optional[ctypes.CDLL] = None


class RasterConfigC(ctypes.Structure):
  """``ms_raster_config`` of include/mi355_splat.h"""
  _fields_ = [
    ('tile_size', c_int32),
    ('antialias', c_int32),
    ('use_alpha_blending', c_int32),
    ('compute_visibility', c_int32),
    ('compute_point_heuristic', c_int32),
    ('reserved', c_int32),
    ('clamp_max_alpha', c_double),
    ('alpha_threshold', c_double),
    ('saturate_threshold', c_double),
  ]


class _SizedStructure(ctypes.Structure):
  """C-ABI structs that begin with their own size (include/mi355_splat.h, MS_VERSION 500): filled in on construction,
  checked by the library on every ms_frame_* call."""

  def __init__(self, *args, **kw):
    assert not args, "keyword arguments only: the leading fields are the struct's size / ABI version"
    super().__init__(**kw)
    self.struct_size = ctypes.sizeof(type(self))
    if hasattr(self, 'abi_version'):
      self.abi_version = ABI_VERSION


class FrameDescC(_SizedStructure):
  """``ms_frame_desc`` of include/mi355_splat.h"""
  _fields_ = [
    ('struct_size', ctypes.c_uint32), ('abi_version', ctypes.c_uint32),
    ('n', c_int64), ('k_capacity', c_int64),
    ('image_w', c_int32), ('image_h', c_int32),
    ('dtype', c_int32), ('f', c_int32), ('sh_degree', c_int32), ('depth16', c_int32),
    ('tile_row_begin', c_int32), ('tile_row_end', c_int32),
    ('projected_input', c_int32), ('mapper', c_int32), ('split_long_runs', c_int32), ('split_seg_len', c_int32),
    ('near_plane', c_double), ('far_plane', c_double), ('blur_cov', c_double), ('clamp_margin', c_double),
    ('raster', RasterConfigC),
  ]


class FrameLayoutC(ctypes.Structure):
  """``ms_frame_layout``: byte sizes of the four blocks and offsets of the arrays inside them"""
  _fields_ = [(name, c_size_t) for name in (
    'keep_n_bytes', 'scratch_n_bytes', 'keep_k_bytes', 'scratch_k_bytes',
    'points7', 'depth', 'colours', 'points7_f32', 'camera_position', 'counters', 'tile_ranges',
    'sorted_keys', 'order', 'counts', 'cum', 'ordered_points', 'tmp_n',
    'overlap_to_point',
    'keys', 'values', 'keys_sorted', 'tmp_k',
    'splat_rows', 'split_scratch')]


class FrameInputsC(_SizedStructure):
  """``ms_frame_inputs``"""
  _fields_ = [('struct_size', ctypes.c_uint32), ('reserved', ctypes.c_uint32)] + [(name, c_void_p) for name in (
    'position', 'log_scaling', 'rotation', 'alpha_logit', 'feature', 'T_camera_world', 'projection',
    'points7', 'depth', 'colours', 'longest_run_host', 'colours_ready_event')]


class FrameGradsC(_SizedStructure):
  """``ms_frame_grads``"""
  _fields_ = [
    ('struct_size', ctypes.c_uint32), ('reserved', ctypes.c_uint32),
    ('image', c_void_p), ('grad_image', c_void_p),
    ('extra_points7', c_void_p), ('extra_depth', c_void_p), ('extra_colours', c_void_p),
    ('moments', c_void_p), ('deterministic', c_int32), ('stage', c_int32), ('fixed_exp', c_void_p),
    ('grad_points7', c_void_p), ('grad_colours', c_void_p),
    ('grad_position', c_void_p), ('grad_log_scaling', c_void_p), ('grad_rotation', c_void_p),
    ('grad_alpha_logit', c_void_p), ('grad_feature', c_void_p), ('grad_camera', c_void_p),
    ('point_heuristic', c_void_p),
    ('point_visibility', c_void_p),
    ('boundary_stride', c_int32), ('gather_world', c_int32),
    ('gather_rows', c_void_p), ('gather_slots', c_void_p), ('gather_route', c_void_p),
    ('boundary_form', c_int32), ('grad_image_broadcast', c_int32),
  ]


class OptimGroupC(ctypes.Structure):
  """``ms_optim_group``"""
  _fields_ = [
    ('struct_size', ctypes.c_uint32), ('group_type', c_int32),
    ('param', c_void_p), ('grad', c_void_p), ('m', c_void_p), ('v', c_void_p),
    ('basis', c_void_p), ('mask_lr', c_void_p), ('point_lr', c_void_p),
    ('d', c_int32), ('bias_correction', c_int32),
    ('lr', c_float), ('beta1', c_float), ('beta2', c_float), ('eps', c_float), ('clip', c_float), ('reserved', c_float),
  ]


# name -> (restype, argtypes); must list every function declared in include/mi355_splat.h
SIGNATURES = {
  'ms_version': (c_int, []),
  'ms_last_error_string': (c_char_p, []),
  'ms_project_fwd': (c_int, [c_void_p] * 6 + [c_int, c_int] + [c_double] * 5 + [c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
  'ms_project_gather': (c_int, [c_void_p] * 4 + [c_int64, c_double, c_double] + [c_void_p] * 4 + [c_int, c_void_p]),
  'ms_project_bwd': (c_int, [c_void_p] * 6 + [c_int, c_int, c_double, c_double, c_void_p, c_int64] + [c_void_p] * 7 + [c_int, c_void_p]),
  'ms_sh_fwd': (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_void_p, c_int, c_void_p]),
  'ms_sh_bwd': (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int] + [c_void_p] * 5 + [c_int, c_int, c_void_p]),
  'ms_tile_count': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p, c_void_p, c_void_p]),
  'ms_depth_sort_keys': (c_int, [c_void_p, c_int64, c_int, c_double, c_double, c_void_p, c_void_p, c_int, c_void_p]),
  'ms_depth_argsort': (c_int, [c_void_p, c_int64, c_int, c_double, c_double, c_int, c_void_p, c_void_p, c_void_p, POINTER(c_size_t), c_void_p]),
  'ms_exclusive_scan_i32': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, POINTER(c_size_t), c_void_p]),
  'ms_tile_emit_keys64': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_int, c_int, c_int, ctypes.c_double, ctypes.c_double, c_void_p, c_void_p, c_void_p]),
  'ms_tile_emit': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_float, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
  'ms_radix_sort_pairs': (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, c_int, c_void_p, POINTER(c_size_t), c_void_p]),
  'ms_segmented_sort_pairs': (c_int, [c_void_p] * 4 + [c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
  'ms_find_ranges': (c_int, [c_void_p, c_int64, c_int, c_int, c_int64, c_void_p, c_void_p]),
  'ms_tile_depth_sort': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
  'ms_fractional_step': (c_int, [c_int, c_int] + [c_void_p] * 7 + [c_int64, c_int, c_float, c_float, c_float, c_float, c_int, c_void_p]),
  'ms_morton_codes64': (c_int, [c_void_p, c_int64, c_void_p, c_void_p, ctypes.c_uint32, c_void_p, c_void_p]),
  'ms_camera_position': (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
  'ms_strip_route_blocks': (c_int, [c_int]),
  'ms_strip_route_count': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
  'ms_strip_route_pack': (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int64] + [c_void_p] * 3 + [c_int64, c_void_p] + [c_void_p] * 2 + [c_void_p]),
  'ms_strip_unpack': (c_int, [c_void_p, c_int64, c_int] + [c_void_p] * 4 + [c_void_p]),
  'ms_strip_return_grads': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p, c_void_p]),
  'ms_strip_route_pack_slots': (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int64] + [c_void_p] * 3 + [c_int64] + [c_void_p] * 5),
  'ms_strip_route_pack_split': (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, c_int64] + [c_void_p] * 3 + [c_int64] + [c_void_p] * 6),
  'ms_strip_return_rows': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_void_p, c_void_p]),
  'ms_fractional_update': (c_int, [c_int, c_int] + [c_void_p] * 11 + [c_int64, c_int, c_float, c_float, c_float, c_float, c_float, c_int, c_void_p]),
  'ms_raster_fwd': (c_int, [c_void_p] * 4 + [c_int, c_int, c_int, POINTER(RasterConfigC)] + [c_void_p] * 3 + [c_int, c_int, c_int, c_void_p]),
  'ms_fixed_point_exponents': (c_int, [c_void_p, c_void_p, c_void_p]),
  'ms_raster_bwd_moments': (c_int, [c_void_p] * 6 + [c_int, c_int, POINTER(RasterConfigC), c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
  'ms_raster_moments_finalize': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
  'ms_raster_split_scratch_bytes': (c_size_t, [c_int64, c_int, c_int, c_int]),
  'ms_raster_fwd_split': (c_int, [c_void_p] * 4 + [c_int64, c_int, c_int, POINTER(RasterConfigC)] + [c_void_p] * 4 + [c_int, c_int, c_int, c_int, c_void_p]),
  'ms_raster_bwd_moments_split': (c_int, [c_void_p] * 4 + [c_int64, c_void_p, c_void_p, c_int, c_int, POINTER(RasterConfigC), c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
  'ms_splat_rows_pack': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
  'ms_raster_fwd_rows': (c_int, [c_void_p] * 3 + [c_int, c_int, POINTER(RasterConfigC)] + [c_void_p] * 3 + [c_int, c_int, c_void_p]),
  'ms_raster_bwd_moments_rows': (c_int, [c_void_p] * 5 + [c_int, c_int, POINTER(RasterConfigC), c_void_p, c_int, c_void_p, c_int, c_int, c_void_p]),
  'ms_frame_layout_query': (c_int, [POINTER(FrameDescC), POINTER(FrameLayoutC)]),
  'ms_frame_uses_moments': (c_int, [POINTER(FrameDescC), c_int]),
  'ms_frame_project': (c_int, [POINTER(FrameDescC), POINTER(FrameInputsC), c_void_p, c_void_p]),
  'ms_frame_sh_colours': (c_int, [POINTER(FrameDescC), POINTER(FrameInputsC), c_void_p, c_void_p]),
  'ms_frame_project_count': (c_int, [POINTER(FrameDescC), POINTER(FrameInputsC), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
  'ms_frame_map_raster': (c_int, [POINTER(FrameDescC), POINTER(FrameInputsC)] + [c_void_p] * 8),
  'ms_frame_backward': (c_int, [POINTER(FrameDescC), POINTER(FrameInputsC), c_void_p, c_void_p, POINTER(FrameGradsC), c_void_p]),
  'ms_probe_raster_bwd': (c_int, [c_void_p, c_void_p]),
  'ms_optim_step_groups': (c_int, [c_int, POINTER(OptimGroupC), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
  'ms_optim_visibility_weights': (c_int, [c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_float] + [c_void_p] * 5),
  'ms_raster_bwd': (c_int, [c_void_p] * 6 + [c_int, c_int, c_int, POINTER(RasterConfigC)] + [c_void_p] * 3 + [c_int, c_int, c_int, c_void_p]),
}


def build(verbose: bool = False, jobs: Optional[int] = None) -> Path:
  """Compile the HIP sources in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
  jobs = jobs or os.cpu_count() or 4
  cmd = ['make', '-C', str(CSRC_DIR), f'-j{jobs}']
  proc = subprocess.run(cmd, capture_output=True, text=True)
  if proc.returncode != 0:
    raise RuntimeError(f"building libmi355_splat.so failed:\n{proc.stdout}\n{proc.stderr}")
  if verbose:
    print(proc.stdout)
  return LIB_PATH


def load() -> ctypes.CDLL:
  global _lib
  if _lib is not None:
    return _lib
  if not LIB_PATH.exists():
    raise RuntimeError(
      f"{LIB_PATH} not found: the gfx950 kernel library has not been built. "
      "Run `python -c 'import __graft_entry__ as g; g.build()'` or `make -C taichi_splatting_amd/csrc`. "
      "There is no CPU fallback for this path.")
  lib = ctypes.CDLL(str(LIB_PATH))
  for name, (restype, argtypes) in SIGNATURES.items():
    fn = getattr(lib, name)   # AttributeError if the symbol is missing
    fn.restype = restype
    fn.argtypes = argtypes
  _lib = lib
  return lib


def check(rc: int, what: str):
  if rc != 0:
    msg = load().ms_last_error_string().decode('utf-8', 'replace')
    if rc == -2:
      raise NotImplementedError(f"{what}: {msg}")
    if rc == -4:
      raise RuntimeError(f"{what}: ABI mismatch between taichi_splatting_amd/_lib.py and libmi355_splat.so: {msg}")
    if rc < 0:
      raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what}: HIP error {rc}: {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
  """Device pointer of a contiguous tensor (None -> NULL)."""
  if t is None:
    return None
  assert t.is_contiguous(), "tensor must be contiguous"
  return t.data_ptr()


def require_gpu(*tensors: torch.Tensor):
  for t in tensors:
    if t is not None and not t.is_cuda:
      raise RuntimeError(
        "taichi_splatting_amd runs on MI355X (gfx950) only: got a tensor on "
        f"{t.device}. There is no CPU fallback; move the inputs to the GPU.")


def dtype_code(dtype: torch.dtype) -> int:
  if dtype == torch.float32:
    return MS_F32
  if dtype == torch.float64:
    return MS_F64
  raise TypeError(f"unsupported dtype {dtype}: float32 or float64 expected")


def current_stream(device) -> int:
  return torch.cuda.current_stream(device).cuda_stream


def raster_config_c(config) -> RasterConfigC:
  return RasterConfigC(
    tile_size=config.tile_size, antialias=int(config.antialias),
    use_alpha_blending=int(config.use_alpha_blending),
    compute_visibility=int(config.compute_visibility),
    compute_point_heuristic=int(config.compute_point_heuristic), reserved=0,
    clamp_max_alpha=config.clamp_max_alpha, alpha_threshold=config.alpha_threshold,
    saturate_threshold=config.saturate_threshold)


def fixed_point_exponents(grad_image: torch.Tensor) -> torch.Tensor:
  """int32[2] device tensor for the deterministic (fixed-point) raster backward: binary exponents of the units the
  per-(patch, splat) sums are committed in, derived on the device from max |dL/dimage| (no host read)."""
  amax = grad_image.detach().abs().amax().to(torch.float32).reshape(1)
  out = torch.empty((2,), dtype=torch.int32, device=grad_image.device)
  check(load().ms_fixed_point_exponents(amax.data_ptr(), out.data_ptr(), current_stream(grad_image.device)),
        "fixed_point_exponents")
  return out
