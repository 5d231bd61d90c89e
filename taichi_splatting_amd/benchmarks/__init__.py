"""Component benchmarks with the command lines of the reference's ``benchmarks/`` package
(bench_projection, bench_sh, bench_tilemapper, bench_rasterizer): same workloads and flags, timed with
HIP events on the launch stream.  ``python -m taichi_splatting_amd.benchmarks.bench_rasterizer``."""
