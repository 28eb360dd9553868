"""RCCL inside the library (sporco_amd_comm_*, sporco_amd.dist.NativeReducer): the plumbing on the
CPU simulator (single-rank communicators: the collective is the identity, the sharded code paths
of sporco_amd_csc_admm_run are what runs), and the real RCCL calls with one rank on the GPU."""

import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, rel_l2


def test_native_reducer_single_rank(backend):
    from sporco_amd import _lib
    from sporco_amd.admm import cbpdn
    from sporco_amd.dist import NativeReducer
    from test_fused_xstep import problem
    red = NativeReducer(0, 1, NativeReducer.unique_id())
    assert red.world_size == 1 and red.rank == 0
    assert red.sum([1.0, 2.0]) == [1.0, 2.0] and red.max(3.0) == 3.0
    D, S = problem(128, 128, 8, 2, seed=4)
    res = []
    for r in (red, None):
        b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': 6, 'RelStopTol': 0.0}),
                           reducer=r)
        res.append((b.solve(), np.asarray(b.getitstat().Rho), b._dev.uses_fused_rows()))
    assert res[0][2] and np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    with pytest.raises(_lib.BackendError):
        NativeReducer(2, 2, NativeReducer.unique_id())          # rank out of range
    red.close()


@pytest.mark.gpu
def test_one_rank_native_rccl_without_torch(gpu_backend):
    env = dict(os.environ, OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    r = subprocess.run([sys.executable, os.path.join(REPO, 'tests', '_native_rccl_worker.py')],
                       env=env, timeout=600, cwd=REPO, capture_output=True, text=True)
    assert r.returncode == 0 and 'NATIVE_RCCL_WORKER_OK' in r.stdout, \
        r.stdout[-2000:] + r.stderr[-4000:]
