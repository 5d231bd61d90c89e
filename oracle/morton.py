"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  CPU restatement of the Morton ordering of
reference ``misc/morton_sort.py``: grid :102-106 + Grid.get_inc/grid_cell :42-52, bit spreading :25-33,
code :56-64, argsort :109-115.

Parity status: the reference kernels need the Taichi runtime, which is not installed here, so no reference
output could be generated: the restatement is pinned by known answers of the published bit-interleave
(axis unit vectors, all-ones cells, the 21-bit limit) and by an independent bit-by-bit interleave
(tests/test_oracle_golden.py::test_morton_*) — "parity unpinned" against reference outputs.
"""
import numpy as np


def spread21(x: np.ndarray) -> np.ndarray:
  x = x.astype(np.uint64) & np.uint64(0x1fffff)
  x = (x | (x << np.uint64(32))) & np.uint64(0x1f00000000ffff)
  x = (x | (x << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
  x = (x | (x << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
  x = (x | (x << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
  x = (x | (x << np.uint64(2))) & np.uint64(0x1249249249249249)
  return x


def cell_code64(cell: np.ndarray) -> np.ndarray:
  """cell (N, 3) unsigned -> 63-bit code, x in bit 0."""
  return spread21(cell[:, 0]) | (spread21(cell[:, 1]) << np.uint64(1)) | (spread21(cell[:, 2]) << np.uint64(2))


def interleave_bitwise(cell: np.ndarray) -> np.ndarray:
  """Independent formulation: place bit b of axis a at bit 3 b + a."""
  out = np.zeros(cell.shape[0], dtype=np.uint64)
  for b in range(21):
    for a in range(3):
      out |= ((cell[:, a].astype(np.uint64) >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
  return out


def grid_at_resolution(points: np.ndarray, resolution: float, size: int = 2 ** 20):
  lower = points.astype(np.float32).min(axis=0)
  upper = (lower + np.float32(size) * np.float32(resolution)).astype(np.float32)
  inc = ((upper - lower) / np.float32(size)).astype(np.float32)
  return lower, inc, size


def morton_codes(points: np.ndarray, resolution: float, size: int = 2 ** 20) -> np.ndarray:
  p = points.astype(np.float32)
  lower, inc, size = grid_at_resolution(p, resolution, size)
  v = ((p - lower) / inc).astype(np.float32)
  v = np.minimum(np.maximum(v, np.float32(0)), np.float32(size - 1))
  return cell_code64(v.astype(np.uint32))


def argsort(points: np.ndarray, resolution: float) -> np.ndarray:
  return np.argsort(morton_codes(points, resolution), kind='stable')


def argsort_dedup(points: np.ndarray, resolution: float) -> np.ndarray:
  codes = morton_codes(points, resolution)
  order = np.argsort(codes, kind='stable')
  sc = codes[order]
  last = np.append(np.nonzero(sc[1:] != sc[:-1])[0], sc.shape[0] - 1) if sc.shape[0] else np.zeros(0, dtype=np.int64)
  return order[last]
