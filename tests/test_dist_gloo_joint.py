"""Image-sharded ConvBPDNJoint on CPU (gloo ranks + the simulator build): the class BASELINE
configs[2] attaches the 8-GPU target to (sporco/admm/cbpdn.py:636-807).  The sharded runs must
reproduce (1) the unmodified reference's own two-image run, (2) the single-process run of the
same problem under uneven shards -- generic float64 chain, and the register-resident joint
kernels under the device-driven and the host-driven loop.  Per-image split precedent:
sporco/dictlrn/prlcnscdl.py:241,508."""

import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO, build_hostsim, load_golden, rel_l2

F = ('ObjFun', 'DFid', 'RegL1', 'RegL21', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho')


@pytest.mark.parametrize('world', [2])
def test_sharded_joint(tmp_path, world):
    build_hostsim()
    out = str(tmp_path / 'joint')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % world,
           '--master-addr', '127.0.0.1', '--master-port', str(29650 + world),
           os.path.join(REPO, 'tests', '_dist_worker_joint.py'), out]
    subprocess.run(cmd, check=True, env=env, timeout=1500, cwd=REPO)
    parts = [np.load(out + '.%d.npz' % r) for r in range(world)]
    p0 = parts[0]
    assert [int(p['lo']) for p in parts[1:]] == [int(p['hi']) for p in parts[:-1]]
    assert int(p0['hi']) - int(p0['lo']) == 2                       # the uneven shard
    if world == 2:
        # 1. against the reference's own run (float64: 1e-9)
        g = load_golden('admm_joint_f64')
        assert rel_l2(np.concatenate([p['ref_Y'] for p in parts], axis=3), g['Y']) < 1e-9
        for p in parts:
            assert int(p['ref_k']) == int(g['k_final'])
            for f in F:
                assert rel_l2(p['ref_' + f], g['it_' + f]) < 1e-9, f
    # 2. float64 generic chain, uneven shards, AutoRho: the single-process run
    assert rel_l2(np.concatenate([p['g_Y'] for p in parts], axis=3), p0['g1_Y']) < 1e-9
    for p in parts:
        assert int(p['g_k']) == int(p0['g1_k'])
        for f in F:
            assert rel_l2(p['g_' + f], p0['g1_' + f]) < 1e-9, f
            assert np.array_equal(p['g_' + f], p0['g_' + f]), f     # identical on every rank
    # 3. register-resident joint kernels: device-driven sharded == host-driven sharded == single
    #    process (float32 sums differ only in the order the images are added)
    assert rel_l2(np.concatenate([p['d_Y'] for p in parts], axis=3), p0['d1_Y']) < 2e-6
    assert rel_l2(np.concatenate([p['h_Y'] for p in parts], axis=3), p0['d1_Y']) < 2e-6
    for p in parts:
        assert int(p['d_k']) == 4
        assert float(p['d_after']) == world * (world + 1) / 2        # collectives aligned
        for f in F:
            assert np.array_equal(p['d_' + f], p0['d_' + f]), f
            assert rel_l2(p['d_' + f], p0['d1_' + f]) < 1e-5, f
            assert rel_l2(p['h_' + f], p0['d1_' + f]) < 1e-5, f
