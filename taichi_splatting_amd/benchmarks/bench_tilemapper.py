"""Entry point under the reference's module path (``benchmarks/bench_tilemapper.py``, a console script in its pyproject):
runs the ``tilemapper`` stage of the table-driven harness (``benchmarks/components.py``) on the same workload.
Options the harness does not have (``--profile``, ``--debug``, ...) are accepted and ignored."""
import argparse

from .components import run


def main(argv=None):
  ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
  ap.add_argument('--iters', type=int, default=100)
  ap.add_argument('--device', type=str, default='cuda:0')
  ap.add_argument('--seed', type=int, default=0)
  args, _ignored = ap.parse_known_args(argv)
  return run('tilemapper', iters=args.iters, device=args.device, seed=args.seed)


if __name__ == '__main__':
  main()
