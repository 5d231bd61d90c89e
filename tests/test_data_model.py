"""Host-side data model: same fields, defaults and helper behaviour as the reference containers
(data_types.py:17-145, perspective/params.py:11-105, rendering.py:27-157)."""
from dataclasses import replace

import pytest
import torch

import taichi_splatting_amd as ts
from taichi_splatting_amd import Gaussians3D, RasterConfig, CameraParams
from taichi_splatting_amd.testing import random_camera, random_3d_gaussians


def test_raster_config_defaults_and_hashable():
  c = RasterConfig()
  assert (c.tile_size, c.pixel_stride, c.clamp_margin, c.antialias, c.blur_cov) == (16, (2, 2), 0.15, False, 0.3)
  assert (c.clamp_max_alpha, c.saturate_threshold, c.use_alpha_blending) == (0.99, 0.9999, True)
  assert abs(c.alpha_threshold - 1 / 255) < 1e-15 and c.median_threshold == 0.25
  assert not c.compute_point_heuristic and not c.compute_visibility
  assert hash(c) == hash(RasterConfig()) and {c: 1}[RasterConfig()] == 1
  c2 = replace(c, tile_size=32, use_alpha_blending=False)
  assert c2.tile_size == 32 and c2 != c
  with pytest.raises(Exception):
    c.tile_size = 8
  with pytest.raises(AssertionError):
    RasterConfig(tile_size=8, pixel_stride=(2, 2))     # reference backward.py:32-33
  with pytest.raises(TypeError):
    RasterConfig(16)                                    # kw_only like the reference


def test_gaussians3d_container():
  torch.manual_seed(0)
  cam = random_camera(image_size=(64, 48))
  g = random_3d_gaussians(10, cam)
  assert g.batch_size == (10,) and g.packed().shape == (10, 11)
  assert len(g.shape_tensors()) == 4
  assert torch.allclose(g.scale, g.log_scaling.exp()) and torch.allclose(g.alpha, g.alpha_logit.sigmoid())
  sub = g[torch.tensor([True] * 3 + [False] * 7)]
  assert sub.batch_size == (3,) and sub.feature.shape == (3, 3)
  g64 = g.to(dtype=torch.float64)
  assert g64.position.dtype == torch.float64
  g64.requires_grad_(True)
  assert g64.rotation.requires_grad
  both = Gaussians3D.concat_batch([g, g])
  assert both.batch_size == (20,)
  s = g.scaled(2.0)
  assert torch.allclose(s.position, g.position * 2)
  t = g.translated(torch.tensor([1., 2., 3.]))
  assert torch.allclose(t.position - g.position, torch.tensor([[1., 2., 3.]]).expand(10, 3))
  # rigid transform by identity keeps positions and rotations (up to sign)
  r = g.transform_rigid(torch.eye(4))
  assert torch.allclose(r.position, g.position, atol=1e-6)
  dots = (r.rotation * g.rotation).sum(-1).abs()
  assert torch.allclose(dots, torch.ones(10), atol=1e-5)
  with pytest.raises(AssertionError):
    Gaussians3D(position=torch.zeros(4, 2), log_scaling=torch.zeros(4, 3), rotation=torch.zeros(4, 4),
                alpha_logit=torch.zeros(4, 1), feature=torch.zeros(4, 3), batch_size=(4,))


def test_camera_params_helpers():
  torch.manual_seed(1)
  cam = random_camera(image_size=(100, 80))
  assert cam.depth_range == (cam.near_plane, cam.far_plane)
  assert cam.T_image_world.shape == (4, 4)
  pos = cam.camera_position
  assert torch.allclose((cam.T_camera_world @ torch.cat([pos, torch.ones(1)]))[:3], torch.zeros(3), atol=1e-5)
  half = cam.scale_image(0.5)
  assert half.image_size == (50, 40) and torch.allclose(half.projection, cam.projection * 0.5)
  assert cam.to(dtype=torch.float64).projection.dtype == torch.float64
  assert 'CameraParams' in repr(cam)


def test_public_surface_and_alias():
  for name in ('render_gaussians', 'Rendering', 'map_to_tiles', 'pad_to_tile', 'Gaussians2D', 'Gaussians3D',
               'RasterConfig', 'evaluate_sh_at', 'rasterize', 'rasterize_with_tiles', 'perspective', 'TaichiQueue'):
    assert hasattr(ts, name), name
  assert ts.pad_to_tile((100, 60), 16) == (112, 64)
  mod = ts.install_as_taichi_splatting()
  import taichi_splatting
  from taichi_splatting.rasterizer.function import rasterize_with_tiles  # noqa: F401
  from taichi_splatting.perspective import CameraParams as C2
  assert taichi_splatting is mod and C2 is CameraParams
