"""N > 1 path on CPU: two gloo ranks, images sharded over ranks, the only
communication being the all-reduce of the per-iteration scalars.  The sharded
run must reproduce the single-process reference trace (same rho schedule,
same iterates) -- SURVEY.md section 8(e)."""

import os
import subprocess
import sys

import numpy as np

from conftest import REPO, build_hostsim, load_golden, rel_l2


def test_two_rank_image_sharding(tmp_path):
    build_hostsim()
    out = str(tmp_path / 'shard')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', '29613',
           os.path.join(REPO, 'tests', '_dist_worker.py'), out]
    subprocess.run(cmd, check=True, env=env, timeout=900, cwd=REPO)
    g = load_golden('admm_multichan_f64')
    parts = [np.load(out + '.%d.npz' % r) for r in range(2)]
    Y = np.concatenate([p['Y'] for p in parts], axis=3)      # axisK = image axis
    assert int(parts[0]['k']) == int(g['k_final'])
    assert rel_l2(Y, g['Y']) < 1e-9
    for p in parts:                                             # identical on every rank
        assert rel_l2(p['ObjFun'], g['it_ObjFun']) < 1e-9
        assert rel_l2(p['Rho'], g['it_Rho']) < 1e-9
        assert rel_l2(p['PrimalRsdl'], g['it_PrimalRsdl']) < 1e-9
        assert rel_l2(p['DualRsdl'], g['it_DualRsdl']) < 1e-9
    # staged path (overridden step) under sharding
    hp = [np.load(out + '.hook.%d.npz' % r) for r in range(2)]
    assert rel_l2(np.concatenate([p['Y'] for p in hp], axis=3), g['Y']) < 1e-9
    for p in hp:
        assert int(p['k']) == int(g['k_final'])
        for f in ('ObjFun', 'DFid', 'Rho'):
            assert rel_l2(p[f], g['it_' + f]) < 1e-9, f
    # device-driven loop under sharding == single-process host-driven loop on both images
    # (same float32 kernels; the sums differ only in the order the two images are added)
    dp = [np.load(out + '.devloop.%d.npz' % r) for r in range(2)]
    Ys = dp[0]['Y_single']
    assert rel_l2(np.concatenate([p['Y'] for p in dp], axis=3), Ys) < 1e-6
    for p in dp:
        assert int(p['k']) == 3
        for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
            assert rel_l2(p[f], dp[0][f + '_single']) < 1e-6, f
            assert np.array_equal(p[f], dp[0][f])              # identical on every rank
    # ... stopping early with unequal host lags: same stopping iteration as the single-process
    # run, aligned collectives afterwards
    ep = [np.load(out + '.earlystop.%d.npz' % r) for r in range(2)]
    k1 = int(ep[0]['k_single'])
    assert 3 < k1 < 60                                          # (it did stop on the tolerance)
    for p in ep:
        assert int(p['k']) == k1
        assert float(p['after']) == 3.0
        assert np.array_equal(p['Rho'], ep[0]['Rho'])
    assert rel_l2(np.concatenate([p['Y'] for p in ep], axis=3), ep[0]['Y_single']) < 1e-5
    # dictionary learning: both ranks hold the dictionary of the single-process run, each its
    # own images' coefficient maps; every statistic is the global one
    g = load_golden('cbpdndl_shard_f64')
    parts = [np.load(out + '.dl.%d.npz' % r) for r in range(2)]
    assert rel_l2(np.concatenate([p['X'] for p in parts], axis=3), g['X']) < 1e-9
    for p in parts:
        assert rel_l2(p['D1'], g['D1']) < 1e-9
        for f in ('ObjFun', 'DFid', 'RegL1', 'Cnstr', 'XPrRsdl', 'XDlRsdl', 'XRho', 'D_L',
                  'D_Rsdl'):
            assert rel_l2(p[f], g['it_' + f]) < 1e-9 or \
                np.max(np.abs(p[f] - g['it_' + f])) < 1e-12, f
    assert np.array_equal(parts[0]['D1'], parts[1]['D1'])
    # FISTA sparse coding sharded over the two images
    for name in ('pgm_btstd_f64', 'pgm_stepbb_f64'):
        g = load_golden(name)
        parts = [np.load(out + '.%s.%d.npz' % (name, r)) for r in range(2)]
        assert rel_l2(np.concatenate([p['X'] for p in parts], axis=3), g['X']) < 1e-9
        for p in parts:
            assert int(p['k']) == int(g['k_final'])
            for f in ('ObjFun', 'Rsdl', 'L'):
                assert rel_l2(p[f], g['it_' + f]) < 1e-9, (name, f)
    # dictionary learning on the sharded FISTA X-step against the single-process run of the
    # same classes (rank-local, all four images)
    for r in range(2):
        p = np.load(out + '.dlpgm.%d.npz' % r)
        assert rel_l2(p['D1'], p['D1_single']) < 1e-9
        assert rel_l2(p['ObjFun'], p['ObjFun_single']) < 1e-9
    # mask decoupling, sharded
    g = load_golden('maskdcpl_f64')
    parts = [np.load(out + '.mdcpl.%d.npz' % r) for r in range(2)]
    assert rel_l2(np.concatenate([p['Y1'] for p in parts], axis=3), g['Y1']) < 1e-9
    for p in parts:
        assert int(p['k']) == int(g['k_final'])
        for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual'):
            assert rel_l2(p[f], g['it_' + f]) < 1e-9, f
    for xm in ('admm', 'pgm'):
        g = load_golden('cbpdndlmd_shard_%s_f64' % xm)
        parts = [np.load(out + '.dlmd_%s.%d.npz' % (xm, r)) for r in range(2)]
        assert rel_l2(np.concatenate([p['X'] for p in parts], axis=3), g['X']) < 1e-9
        for p in parts:
            assert rel_l2(p['D1'].squeeze(), g['D1'].squeeze()) < 1e-9
            for f in ('ObjFun', 'DFid', 'RegL1'):
                assert rel_l2(p[f], g['it_' + f]) < 1e-9, (xm, f)
    # consensus dictionary updates with the average as an all-reduce: every rank reproduces the
    # single-process dictionary and traces of the four-image reference runs
    for name in ('ccmod_cns_shard_f64', 'ccmodmd_cns_shard_f64'):
        g = load_golden(name)
        parts = [np.load(out + '.%s.%d.npz' % (name, r)) for r in range(2)]
        for p in parts:
            assert int(p['k']) == int(g['k_final'])
            assert rel_l2(p['D'], g['D']) < 1e-9 and rel_l2(p['Y'], g['Y']) < 1e-9
            for f in ('DFid', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho'):
                assert rel_l2(p[f], g['it_' + f]) < 1e-9, (name, f)
        assert np.array_equal(parts[0]['D'], parts[1]['D'])
    for tag, name in (('dlcns', 'cbpdndl_shard_cns_f64'), ('dlmdcns', 'cbpdndlmd_shard_cns_f64')):
        g = load_golden(name)
        parts = [np.load(out + '.%s.%d.npz' % (tag, r)) for r in range(2)]
        assert rel_l2(np.concatenate([p['X'] for p in parts], axis=3), g['X']) < 1e-9
        for p in parts:
            assert rel_l2(p['D1'].squeeze(), g['D1'].squeeze()) < 1e-9
            for f in g.keys():
                if f.startswith('it_') and f[3:] in p.files and f[3:] not in ('Cnstr', 'Iter'):
                    assert rel_l2(p[f[3:]], g[f]) < 1e-9, (name, f)
