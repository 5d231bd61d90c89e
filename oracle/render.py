"""Oracle: the full render path on the CPU (render_gaussians, reference renderer.py:23-108)."""
from __future__ import annotations

import numpy as np
import torch

from . import mapper, projection, raster, sh


def render_forward(position, log_scaling, rotation, alpha_logit, feature, T_camera_world, proj,
                   image_size, depth_range, cfg, use_sh=False, blur_cov=0.3, clamp_margin=0.15,
                   tile_rows=None):
  """Returns dict(image, alpha, points, depths, indexes, features, o2p, ranges)."""
  points, depths, idx = projection.apply(position, log_scaling, rotation, alpha_logit, T_camera_world,
                                         proj, image_size, depth_range, blur_cov, clamp_margin,
                                         cfg.alpha_threshold)
  if use_sh:
    cam_pos = torch.inverse(T_camera_world)[0:3, 3]
    feats = sh.evaluate_sh_at(feature, position, idx, cam_pos)
  else:
    feats = feature[idx]
  ndc = projection.ndc_depth(depths, depth_range[0], depth_range[1])
  o2p, ranges, _ = mapper.map_to_tiles(points.detach().numpy().astype(np.float32),
                                       ndc.detach().numpy().astype(np.float32), image_size,
                                       cfg.tile_size, cfg.alpha_threshold, tile_rows=tile_rows)
  o2p_t, ranges_t = torch.from_numpy(o2p), torch.from_numpy(ranges)
  image, alpha, vis = raster.forward(points.detach(), feats.detach(), ranges_t, o2p_t, image_size, cfg,
                                     tile_rows=tile_rows)
  return dict(image=image, alpha=alpha, visibility=vis, points=points, depths=depths, indexes=idx,
              features=feats, o2p=o2p_t, ranges=ranges_t)
