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


def _gpu_count():
    import sporco_amd
    try:
        return sporco_amd.device_count()
    except Exception:       # noqa: BLE001
        return 0


@pytest.mark.gpu
def test_two_rank_native_rccl(gpu_backend, tmp_path):
    """N > 1 over RCCL inside the library, when the box has two GPUs (the 1-GPU boxes of this build
    skip it: RCCL refuses two ranks on one device): unequal image shards, the device-driven loop
    stopping early with unequal host lag, the host-driven loop, the dictionary-learning gradient
    all-reduce -- against the same problems solved by one rank."""
    if _gpu_count() < 2:
        pytest.skip('needs two GPUs (this box has %d)' % _gpu_count())
    world = 2
    idfile = str(tmp_path / 'rccl_id')
    env = dict(os.environ, OMP_NUM_THREADS='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, 'tests', '_native_rccl_worker2.py'), str(r),
                               str(world), idfile], env=env, cwd=REPO, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=900) for p in procs]
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and 'NATIVE_RCCL_WORKER2_OK %d' % r in so, so[-2000:] + se[-4000:]
    one = np.load(idfile + '.single.npz')
    parts = [np.load(idfile + '.admm.%d.npz' % r) for r in range(world)]
    rest = [np.load(idfile + '.rest.%d.npz' % r) for r in range(world)]
    assert 3 < int(one['k']) < 200
    for p in parts:
        assert int(p['k']) == int(one['k'])
        assert np.array_equal(p['Rho'], parts[0]['Rho'])
        assert rel_l2(p['Rho'], one['Rho']) < 1e-5
    assert rel_l2(np.concatenate([p['Y'] for p in parts], axis=3), one['Y']) < 1e-4
    assert rel_l2(np.concatenate([p['Yc'] for p in rest], axis=3), one['Yc']) < 1e-5
    for p in rest:
        assert np.array_equal(p['D1'], rest[0]['D1'])
        assert rel_l2(p['D1'], one['D1']) < 1e-5 and rel_l2(p['Obj'], one['Obj']) < 1e-5
