"""bench.py --gpus N must mean N ranks, or fail loudly ("Next round" item 1; contract: SURVEY.md 8(e)).
No GPU here: the device-count refusals, the launcher's command line, and a gloo rehearsal of the self-launch
(parent never touches a device, child output relayed, child exit code returned)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout)


def test_launcher_command_line():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(8, ['--gpus', '8', '--steps', '20', '--warmup', '5'], 29511)
    assert cmd[:3] == [sys.executable, '-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and '--nproc-per-node=8' in cmd
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1'
    assert cmd[cmd.index('--master-port') + 1] == '29511'
    i = cmd.index(os.path.abspath(BENCH))
    assert cmd[i + 1:] == ['--gpus', '8', '--steps', '20', '--warmup', '5']      # same script, same arguments


def test_gpus_2_without_two_devices_exits_nonzero_with_a_message():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('two devices visible: the refusal cannot be provoked here')
    r = _run(['--gpus', '2', '--steps', '1', '--warmup', '0'])
    assert r.returncode != 0
    assert '--gpus 2 requested but only' in r.stderr
    assert 'n_gpus' not in r.stdout                        # no JSON line at all, certainly not one that says n_gpus 1


def test_gpus_disagreeing_with_world_size_exits_nonzero():
    r = _run(['--gpus', '8', '--steps', '1', '--warmup', '0'], {'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert r.returncode != 0
    assert 'disagrees with WORLD_SIZE=2' in r.stderr
    assert r.stdout.strip() == ''


def test_gpus_1_without_a_device_fails_loudly_instead_of_falling_back():
    import torch
    if torch.cuda.device_count() >= 1:
        pytest.skip('a device is visible')
    r = _run(['--gpus', '1', '--steps', '1', '--warmup', '0'])
    assert r.returncode != 0 and 'no GPU visible' in r.stderr


def test_self_launch_rehearsal_over_gloo_relays_rank0_line_and_exit_code():
    r = _run(['--gpus', '2', '--selftest-gloo'])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1                                  # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['gpus_arg'] == 2 and d['sum_of_ranks_plus_one'] == 3.0
    assert 'torch.distributed.run' in r.stderr and '--nproc-per-node=2' in r.stderr
    # a failing rank makes the parent fail with a non-zero code
    r = _run(['--gpus', '2', '--selftest-gloo', '--selftest-fail-rank', '1'])
    assert r.returncode != 0
    assert 'job failed with exit code' in r.stderr
