"""Component benchmarks: ``python -m taichi_splatting_amd.benchmarks [stage ...] [--iters N]`` times the stages of
the render path on the workloads the reference's ``benchmarks/`` package and SURVEY.md 8(d) name
(see ``components.WORKLOADS``), with HIP events on the launch stream."""
from .components import WORKLOADS, run, time_ms   # noqa: F401
