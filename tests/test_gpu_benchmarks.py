"""-m gpu: the component benchmarks run end to end with their default workloads (reference
tests/test_benchmarks.py), few iterations."""
import pytest

pytestmark = pytest.mark.gpu


def test_bench_tilemapper():
  from taichi_splatting_amd.benchmarks import bench_tilemapper, util
  bench_tilemapper.bench_tilemapper(bench_tilemapper.parse_args(['--iters', '5']))
  assert util.RESULTS['tile_mapper'] > 0


def test_bench_projection():
  from taichi_splatting_amd.benchmarks import bench_projection, util
  bench_projection.bench_projection(bench_projection.parse_args(['--iters', '5']))
  assert all(util.RESULTS[k] > 0 for k in ('forward', 'backward (gaussians)', 'backward (extrinsics)',
                                           'backward (intrinsics)', 'backward (everything)'))


def test_bench_rasterizer():
  from taichi_splatting_amd.benchmarks import bench_rasterizer, util
  bench_rasterizer.bench_rasterizer(bench_rasterizer.parse_args(['--iters', '3']))
  assert util.RESULTS['backward (all)'] > 0 and util.RESULTS['forward_vis'] > 0


def test_bench_sh():
  from taichi_splatting_amd.benchmarks import bench_sh, util
  bench_sh.bench_sh(bench_sh.parse_args(['--iters', '5']))
  assert util.RESULTS['backward (all)'] > 0
