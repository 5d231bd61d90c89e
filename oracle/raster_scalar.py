"""Oracle, second opinion: the tile rasterizer as literal scalar loops (pure Python floats, one pixel and one
splat at a time), written independently of the vectorised ``oracle/raster.py`` — no matrices, no cumulative
products, no shared helpers — to cross-check it on small cases, in particular tiles whose lists span several
groups of the reference's shared-memory staging.  Test infrastructure only.

Follows rasterizer/forward.py:39-135 and rasterizer/backward.py:97-224 with gaussian_pdf /
gaussian_pdf_with_grad (taichi_lib/generic.py:311-336); plain pdf and alpha blending only.  Every splat of a
tile's range is visited exactly once (the reference's in-group loop bound is not copied).
"""
from __future__ import annotations

import math


def _pdf(px, py, g):
  mx, my, ax, ay, sx, sy = g[0], g[1], g[2], g[3], g[4], g[5]
  dx, dy = px - mx, py - my
  tx = (dx * ax + dy * ay) / sx                 # d . axis / sigma_x            (generic.py:313-314)
  ty = (-dx * ay + dy * ax) / sy                # d . perp(axis) / sigma_y, perp(x, y) = (-y, x)
  return math.exp(-0.5 * (tx * tx + ty * ty)), dx, dy, tx, ty


def forward(points, feats, ranges, o2p, image_size, tile_size=16, clamp_max_alpha=0.99, alpha_threshold=1. / 255.):
  """points: list of 7-lists, feats: list of F-lists, ranges: list of (start, end) per tile, o2p: list of ints.
  Returns (image[y][x][c], alpha[y][x], visibility[point])."""
  w, h = image_size
  tiles_wide = (w + tile_size - 1) // tile_size
  F = len(feats[0]) if feats else 0
  image = [[[0.0] * F for _ in range(w)] for _ in range(h)]
  alpha_img = [[0.0] * w for _ in range(h)]
  vis = [0.0] * len(points)
  for y in range(h):
    for x in range(w):
      tile = (x // tile_size) + (y // tile_size) * tiles_wide
      start, end = ranges[tile]
      total = 0.0
      acc = [0.0] * F
      for k in range(start, end):
        pid = o2p[k]
        g = points[pid]
        p, *_ = _pdf(x + 0.5, y + 0.5, g)                      # pixel centre (forward.py:46)
        a = min(g[6] * p, clamp_max_alpha)                     # clamp, then threshold (forward.py:99-101)
        if a > alpha_threshold:
          wgt = a * (1.0 - total)
          total += wgt
          for c in range(F):
            acc[c] += feats[pid][c] * wgt
          vis[pid] += wgt
      image[y][x] = acc
      alpha_img[y][x] = total
  return image, alpha_img, vis


def backward(points, feats, ranges, o2p, image, grad_image, image_size, tile_size=16, clamp_max_alpha=0.99,
             alpha_threshold=1. / 255., saturate_threshold=0.9999):
  """Returns (grad_points[point][7], grad_feats[point][F], heuristic[point][2])."""
  w, h = image_size
  tiles_wide = (w + tile_size - 1) // tile_size
  F = len(feats[0]) if feats else 0
  gp = [[0.0] * 7 for _ in points]
  gf = [[0.0] * F for _ in points]
  heur = [[0.0, 0.0] for _ in points]
  for y in range(h):
    for x in range(w):
      tile = (x // tile_size) + (y // tile_size) * tiles_wide
      start, end = ranges[tile]
      total = 0.0
      remaining = list(image[y][x])                            # backward.py:106
      G = grad_image[y][x]
      for k in range(start, end):
        if total >= saturate_threshold:                        # backward.py:116,142,154
          break
        pid = o2p[k]
        g = points[pid]
        p, dx, dy, tx, ty = _pdf(x + 0.5, y + 0.5, g)
        a_raw = g[6] * p
        if a_raw <= alpha_threshold:
          continue
        a = min(a_raw, clamp_max_alpha)
        T = 1.0 - total
        wgt = a * T
        total += wgt
        f = feats[pid]
        alpha_grad = 0.0
        for c in range(F):
          remaining[c] -= f[c] * wgt                           # backward.py:171-174
          alpha_grad += (f[c] * T - remaining[c] / (1.0 - a)) * G[c]
          gf[pid][c] += wgt * G[c]
        # straight-through clamp (backward.py:158-163): d alpha / d (alpha_pt * p) = 1
        aag = g[6] * alpha_grad
        ax, ay, sx, sy = g[2], g[3], g[4], g[5]
        txs, tys = tx / sx, ty / sy
        dmx = p * (txs * ax + tys * -ay)                       # generic.py:330-334
        dmy = p * (txs * ay + tys * ax)
        dax = p * (txs * -dx + tys * -dy)                      # perp(d) = (-dy, dx)
        day = p * (txs * -dy + tys * dx)
        dsx = tx * tx * p / sx
        dsy = ty * ty * p / sy
        for i, v in enumerate((dmx, dmy, dax, day, dsx, dsy)):
          gp[pid][i] += aag * v
        gp[pid][6] += p * alpha_grad
        heur[pid][0] += aag * aag                              # backward.py:190-194
        heur[pid][1] += abs(aag * dmx) + abs(aag * dmy)
  return gp, gf, heur
