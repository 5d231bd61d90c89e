"""CPU oracle of the splatting hot path — TEST INFRASTRUCTURE ONLY.

A plain torch/numpy restatement of the reference algorithm for the path
project -> SH colour -> tile mapper -> alpha composite (forward and backward).  It is imported only
by ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``, always as
the checker / reported baseline and never as the thing shipped: the product
(``taichi_splatting_amd``) never imports this package and has no CPU fallback.

Parity status (see DESIGN.md "Oracle"):
  * projection, SH, ndc depth, random-data generators: PINNED against the reference's own
    ``torch_lib`` (imported in the build container by tests/golden/make_fixtures.py; outputs and
    autograd gradients committed under tests/golden/).
  * rasterizer and tile mapper: the reference holds no golden vector or CPU implementation for them
     — "parity unpinned" by reference data.  They are pinned instead by
    hand-computable known-answer tests, gradcheck (the reference's own rasterizer test protocol,
    tests/test_rasterizer.py:62-90), autograd-vs-literal-backward agreement, the visibility identity
    (tests/test_visibility.py:34-64) and brute-force mapper invariants.

  * Morton codes (N4): reference kernels need the Taichi runtime (absent) — pinned by known answers of the
    published bit interleave and an independent bitwise formulation ("parity unpinned" by reference data).

Each function cites the reference file:line (relative to /root/reference/taichi_splatting) it follows.
"""
from . import projection, sh, mapper, raster, render, optim, morton  # noqa: F401
