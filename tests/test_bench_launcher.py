"""The N > 1 path of bench.py itself on CPU (world size 2, gloo, kernels on the CPU emulator through bench.py's
`--emulator-lib` test hook): `python bench.py --gpus 2` with no launcher must start two ranks by itself, report the world
size the process group saw, shard images (timed region), split one image + all-gather the tiles, and run data-parallel
training steps with one gradient all-reduce - and it must refuse to print a line with the wrong `n_gpus`."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT
from emu_util import emu_lib

BENCH = os.path.join(ROOT, 'bench.py')


def run_bench(args, env=None, timeout=900):
    e = dict(os.environ, NEURAY_EMU_THREADS='2', OMP_NUM_THREADS='2')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.fixture(scope='module')
def emu_path():
    import build_emu
    emu_lib()
    return build_emu.build()


def test_gpus_2_without_a_launcher_starts_two_ranks(emu_path):
    p = run_bench(['--gpus', '2', '--steps', '1', '--warmup', '1', '--emulator-lib', emu_path])
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout                  # ONE JSON line, printed by rank 0 only
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['world_size_seen_by_process_group'] == 2 and line['scaling'] == 'weak'
    assert 'NOT a measurement' in line['data']
    rays = 16 * 24
    assert abs(line['value'] - 2 * rays / (line['ms_per_step'] * 1e-3)) <= 1e-6 * line['value']     # whole-job aggregate over both ranks
    assert line['split_image'].get('gathered_rays') == rays and line['split_image']['scaling'] == 'strong', line['split_image']
    t = line['train_ddp']
    assert t.get('world_size') == 2 and t['replicas_identical_after_steps'] is True and t['allreduce_bytes_per_step'] > 2e5, t


def test_a_rank_lost_inside_a_side_leg_costs_the_side_legs_not_the_line(emu_path):
    """rank 1 raises inside the first side leg and never joins another collective: rank 0 is left waiting in one.  The watchdog must
    print the (complete) headline line once and end every rank with exit code 0."""
    p = run_bench(['--gpus', '2', '--steps', '1', '--warmup', '0', '--emulator-lib', emu_path, '--fail-leg-on-rank', '1', '--side-leg-timeout', '12'],
                  timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['value'] > 0 and line['world_size_seen_by_process_group'] == 2
    assert 'did not finish within 12 s' in line['side_legs_error'] and 'train_ddp' not in line


def test_split_image_as_the_timed_region(emu_path):
    p = run_bench(['--gpus', '2', '--steps', '1', '--warmup', '0', '--split-image', '--no-side-legs', '--emulator-lib', emu_path])
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith('{')][0])
    assert line['n_gpus'] == 2 and line['scaling'] == 'strong' and 'all-gather' in line['config']['parallelism']
    assert abs(line['value'] - 16 * 24 / (line['ms_per_step'] * 1e-3)) <= 1e-6 * line['value']


def test_refuses_a_wrong_world_size():
    p = run_bench(['--gpus', '2', '--steps', '1'])            # no GPU in this container, no emulator hook
    assert p.returncode != 0 and 'GPU(s) visible' in p.stderr and '{' not in p.stdout
    p = run_bench(['--gpus', '4', '--steps', '1'], env={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert p.returncode != 0 and 'WORLD_SIZE=2' in p.stderr and '{' not in p.stdout
    p = run_bench(['--gpus', '1', '--steps', '1'], env={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert p.returncode != 0 and '{' not in p.stdout


def test_gpus_8_world_size_of_the_target_node(emu_path):
    """the driver's largest launch (8 ranks, one per GPU of an MI355X node) on gloo + emulator: eight ranks start, the line
    reports 8, the split image (16 x 24 = 384 rays, 48 per rank) is gathered whole, replicas stay identical"""
    p = run_bench(['--gpus', '8', '--steps', '1', '--warmup', '0', '--emulator-lib', emu_path],
                  env={'NEURAY_EMU_THREADS': '1', 'OMP_NUM_THREADS': '1'}, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line['n_gpus'] == 8 and line['world_size_seen_by_process_group'] == 8
    assert abs(line['value'] - 8 * 16 * 24 / (line['ms_per_step'] * 1e-3)) <= 1e-6 * line['value']
    assert line['split_image'].get('gathered_rays') == 16 * 24, line['split_image']
    assert line['train_ddp'].get('world_size') == 8 and line['train_ddp']['replicas_identical_after_steps'] is True, line['train_ddp']
