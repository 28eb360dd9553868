"""RCCL path on one GPU: a one-rank 'nccl' process group (see tests/_nccl_worker.py).
Multi-GPU runs are the driver's; this pins the pieces that only they exercise."""

import os
import subprocess
import sys

import pytest

from conftest import REPO


@pytest.mark.gpu
def test_one_rank_nccl_reducer(gpu_backend):
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29641', OMP_NUM_THREADS='1')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tests', '_nccl_worker.py')],
                       env=env, timeout=600, cwd=REPO, capture_output=True, text=True)
    assert r.returncode == 0 and 'NCCL_WORKER_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
