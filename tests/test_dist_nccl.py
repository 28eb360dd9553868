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


@pytest.mark.gpu
def test_two_ranks_share_one_gpu(gpu_backend, tmp_path):
    """Two real processes on ONE MI355X, gloo between them (tests/_gloo_gpu_worker.py): the
    device-driven loop with the all-reduce hook between the local sums and the control kernel,
    one image per rank, against the single-process run of both images -- same stopping
    iteration, same rho schedule, aligned collectives after an early stop noticed at different
    times."""
    import numpy as np
    from conftest import rel_l2
    out = str(tmp_path / 'gg')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', '29647',
           os.path.join(REPO, 'tests', '_gloo_gpu_worker.py'), out]
    r = subprocess.run(cmd, env=env, timeout=900, cwd=REPO, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.count('GLOO_GPU_WORKER_OK') == 2, \
        r.stdout[-2000:] + r.stderr[-4000:]
    p = [np.load(out + '.%d.npz' % k) for k in range(2)]
    for name in ('fixed', 'early'):
        k1 = int(p[0][name + '_k1'])
        if name == 'early':
            assert 3 < k1 < 60
        for q in p:
            assert int(q[name + '_k']) == k1
            assert float(q[name + '_after']) == 3.0
            assert np.array_equal(q[name + '_Rho'], p[0][name + '_Rho'])      # every rank alike
            for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
                assert rel_l2(q[name + '_' + f], p[0][name + '_' + f + '1']) < 1e-5, (name, f)
        Y = np.concatenate([q[name + '_Y'] for q in p], axis=3)
        assert rel_l2(Y, p[0][name + '_Y1']) < 1e-5
