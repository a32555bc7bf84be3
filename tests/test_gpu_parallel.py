"""Data-parallel path on the GPU: two ranks (gloo transport, sharing GPU 0) run tests/dp_worker.py."""
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
