"""bench.py launcher path on real devices: `--gpus N` must spawn N RCCL ranks itself and report n_gpus = N in
both decompositions (tests/test_distributed_cpu.py covers the collectives' logic on gloo)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
SMALL = ['--n', '200000', '--size', '512', '--steps', '3', '--warmup', '1', '--no-cpu-baseline']


def run_bench(*extra):
  proc = subprocess.run([sys.executable, str(ROOT / 'bench.py'), *SMALL, *extra], capture_output=True, text=True,
                        timeout=900)
  assert proc.returncode == 0, proc.stderr[-3000:]
  lines = [l for l in proc.stdout.splitlines() if l.startswith('{')]
  assert len(lines) == 1, proc.stdout
  return json.loads(lines[0])


@pytest.mark.gpu
def test_launcher_world_size_one_uses_rccl():
  """one rank under torch.distributed.run: init_process_group('nccl'), barrier and all_gather on the device"""
  out = run_bench('--gpus', '1', '--launcher', '--mode', 'both')
  assert out['n_gpus'] == 1 and out['value'] > 0
  assert out['job']['world_size'] == 1 and 'nccl' in out['job']['backend'] and len(out['job']['devices']) == 1


@pytest.mark.gpu
@pytest.mark.parametrize('extra', [(), ('--rank-graph',), ('--legacy-steps',)])
def test_rank_steps_on_one_rank_report_stages(extra):
  """the N > 1 code path of bench.py with world size 1 (RCCL collectives on one device): sync-free steps, eagerly and
  replayed from a HIP graph, and the round-2 steps"""
  out = run_bench('--gpus', '1', '--launcher', '--mode', 'sharded', '--no-graph', '--no-sweep', *extra)
  assert out['config']['mode'] == 'sharded' and out['value'] > 0
  if '--legacy-steps' not in extra:
    assert out['host_syncs_per_step'] == 0
    # the timed loop moves the camera: four poses, each with its own bounds / capacities (probed outside the loop)
    assert out['modes']['sharded']['rank0_step']['views'] == 4


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['strips', 'sharded'])
def test_rank_step_with_rccl_collectives_in_a_hip_graph(mode):
  """RCCL world size 1, BOTH decompositions: the rank step — reduce-scatter + all-gather, or the two all-to-alls —
  captured into a HIP graph and replayed (what `--rank-graph` does on a real node)"""
  out = run_bench('--gpus', '1', '--launcher', '--mode', mode, '--no-graph', '--no-sweep', '--rank-graph')
  assert out['config']['mode'] == mode and out['value'] > 0 and out['host_syncs_per_step'] == 0
  assert out['modes'][mode]['rank0_step'].get('hip_graph') is True


@pytest.mark.gpu
def test_dry_run_of_the_eight_rank_bench_on_one_gpu():
  """`bench.py --gpus 8 --dry-run`: the code path the driver's 8-GPU run takes — per-view balanced bounds (all-reduce),
  probes (unequal-split all-to-all, MAX all-reduce), both rank steps over four camera poses, stage gathering — with
  eight gloo ranks sharing this GPU; rank 0 checks that all ranks issue the same collectives with matching shapes and
  that every all-to-all split pairs up (bench.check_collective_logs), and that no fixed-capacity buffer overflowed."""
  out = run_bench('--gpus', '8', '--dry-run', '--no-graph', '--no-sweep', '--n', '120000')
  assert out['n_gpus'] == 8 and set(out['modes']) == {'strips', 'sharded'} and not out.get('failed_modes')
  dry = out['dry_run']
  assert dry['ranks'] == 8 and dry['sequences_match'] and dry['splits_pair_up']
  ops = dry['collectives_issued_by_rank0']
  assert ops.get('all_to_all_single', 0) > 0 and ops.get('reduce_scatter_tensor', 0) > 0 and ops.get('all_gather_into_tensor', 0) > 0
  for m in out['modes'].values():
    assert len(m['rank_ms_per_step']) == 8 and len(m['rank0_step']['stage_ms_per_rank']) == 8


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs")
def test_two_gpus_both_modes():
  out = run_bench('--gpus', '2')
  assert out['n_gpus'] == 2
  assert set(out['modes']) == {'strips', 'sharded'}
  for m in out['modes'].values():
    assert len(m['rank_ms_per_step']) == 2 and m['value'] > 0 and m['rank0_step']['stage_ms_per_rank'] and len(m['rank0_step']['stage_ms_per_rank']) == 2
  assert out['config']['mode'] in out['modes']
