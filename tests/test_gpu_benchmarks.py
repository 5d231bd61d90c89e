"""-m gpu: the component benchmark harness runs every stage on its (reduced) workload and reports a time per
case (the reference's tests/test_benchmarks.py is the same kind of smoke test)."""
import pytest

pytestmark = pytest.mark.gpu

SMALL = {'projection': dict(n=200_000), 'sh': dict(n=100_000), 'tilemapper': dict(n=100_000, size=(512, 384)),
         'rasterizer': dict(n=100_000, size=(512, 384)), 'rasterizer_d': dict(n=200_000, size=(384, 384))}


@pytest.mark.parametrize('stage', sorted(SMALL))
def test_component_benchmark(stage):
  from taichi_splatting_amd import benchmarks
  results = benchmarks.run(stage, iters=3, **SMALL[stage])
  assert len(results) >= 2 and all(ms > 0 for ms in results.values())
