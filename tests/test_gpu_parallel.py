"""Data-parallel path on the GPU: two ranks (gloo transport, sharing GPU 0) and a one-rank RCCL group run tests/dp_worker.py."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_two_ranks_average_gradients_and_stay_in_sync():
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', '29533', os.path.join(ROOT, 'tests', 'dp_worker.py')]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert 'DP_OK rank=0' in out and 'DP_OK rank=1' in out, out[-3000:]


def test_one_rank_rccl_group_runs_the_nccl_path():
    """RCCL init, ReduceOp.AVG, broadcast and the overlapped bucketed all-reduce on the comm stream, on the one GPU of
    the test box (the 8-GPU run is the driver's): results must equal the purely local ones."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    env['DP_BACKEND'] = 'nccl'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29534', os.path.join(ROOT, 'tests', 'dp_worker.py')]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    out = p.stdout + p.stderr
    assert p.returncode == 0, out[-3000:]
    assert 'DP_OK rank=0' in out, out[-3000:]


def test_bench_under_torchrun_one_rank_rccl():
    """bench.py exactly as the driver launches it for N > 1, with one rank (--force-dp keeps the collective path on)."""
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', '29535', os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '3', '--warmup', '1',
           '--no-cpu-baseline', '--no-profile', '--no-fp32-reference', '--force-dp']
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    import json
    line = [l for l in p.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 1 and d['value'] > 100 and d['config']['parallelism'] == 'dp1'
