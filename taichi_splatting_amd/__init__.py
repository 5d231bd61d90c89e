"""taichi_splatting_amd — MI355X (gfx950) native back end for the taichi_splatting render path.

Same public surface as the reference package (``taichi_splatting/__init__.py:1-33``); every
device stage is a hand-written HIP kernel in ``csrc/`` reached through the C-ABI of
``include/mi355_splat.h``.  ``install_as_taichi_splatting()`` registers this package under the
reference's import name so existing callers run unchanged.
"""
from .renderer import render_gaussians, render_projected, viewspace_gradient
from .rendering import Rendering, RenderedPoints
from .data_types import Gaussians2D, Gaussians3D, RasterConfig
from .mapper.tile_mapper import map_to_tiles, pad_to_tile
from .rasterizer import rasterize, rasterize_with_tiles, RasterOut
from .spherical_harmonics import evaluate_sh_at
from . import perspective
from . import cuda_lib
from . import cuda_lib as hip_lib
from . import optim
from .perspective import CameraParams
from .taichi_queue import TaichiQueue, taichi_queue, queued

__version__ = '0.5.0'       # = MS_VERSION 500 of include/mi355_splat.h (tests/test_abi.py holds the two together)

__all__ = [
  'render_gaussians', 'Rendering',
  'map_to_tiles', 'pad_to_tile',
  'Gaussians2D', 'Gaussians3D',
  'RasterConfig', 'evaluate_sh_at',
  'rasterize', 'rasterize_with_tiles',
  'perspective', 'TaichiQueue',
]


def install_as_taichi_splatting():
  """Register this package (and its submodules) as ``taichi_splatting`` in ``sys.modules``."""
  import importlib
  import sys
  me = sys.modules[__name__]
  sys.modules.setdefault('taichi_splatting', me)
  for sub in ('data_types', 'renderer', 'rendering', 'taichi_queue', 'spherical_harmonics',
              'perspective', 'perspective.params',
              'perspective.projection', 'mapper', 'mapper.tile_mapper', 'rasterizer',
              'rasterizer.function', 'cuda_lib', 'misc', 'misc.renderer2d', 'misc.morton_sort', 'optim', 'optim.fractional',
              'optim.visibility_aware', 'optim.parameter_class', 'optim.util', 'benchmarks', 'benchmarks.util',
              'benchmarks.bench_projection', 'benchmarks.bench_rasterizer', 'benchmarks.bench_tilemapper',
              'benchmarks.bench_sh', 'examples',
              'examples.fit_image_gaussians'):
    mod = importlib.import_module(f'{__name__}.{sub}')
    sys.modules.setdefault(f'taichi_splatting.{sub}', mod)
  # module names of the reference whose contents live elsewhere here: evaluate_sh_at (reference
  # indexed_spherical_harmonics.py:166) is in spherical_harmonics, restore_grad (optim/autograd.py) in optim.fractional
  for alias, target in (('indexed_spherical_harmonics', 'spherical_harmonics'), ('optim.autograd', 'optim.fractional')):
    sys.modules.setdefault(f'taichi_splatting.{alias}', importlib.import_module(f'{__name__}.{target}'))
  # the reference keeps its scene generators under tests/ (tests/random_data.py)
  testing = importlib.import_module(f'{__name__}.testing')
  sys.modules.setdefault('taichi_splatting.tests', testing)
  sys.modules.setdefault('taichi_splatting.tests.random_data', importlib.import_module(f'{__name__}.testing.random_data'))
  return me
