"""The N > 1 path at world sizes 4 and 8 with UNEVEN image shards, on CPU (gloo ranks + the
simulator build): bugs that depend on the number of ranks or on equal shards cannot hide behind the
two-rank test.  Each sharded run is compared with the same problem solved in one process.
Precedent for the per-image split: sporco/dictlrn/prlcnscdl.py:241,508."""

import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import HOSTSIM_LIB, REPO, build_hostsim, rel_l2


def test_shard_bounds_cover_the_images():
    from sporco_amd.dist import shard_bounds
    for n, w in ((7, 4), (13, 8), (8, 8), (256, 8), (9, 2)):
        b = [shard_bounds(n, r, w) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
    with pytest.raises(ValueError):
        shard_bounds(3, 0, 4)


@pytest.mark.parametrize('world', [4, 8])
def test_uneven_shards(tmp_path, world):
    build_hostsim()
    out = str(tmp_path / 'shard')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % world,
           '--master-addr', '127.0.0.1', '--master-port', str(29620 + world),
           os.path.join(REPO, 'tests', '_dist_worker_n.py'), out]
    subprocess.run(cmd, check=True, env=env, timeout=1500, cwd=REPO)
    parts = [np.load(out + '.%d.npz' % r) for r in range(world)]
    p0 = parts[0]
    assert [int(p['lo']) for p in parts[1:]] == [int(p['hi']) for p in parts[:-1]]
    # 1. float64 generic chain: iterates and every trace of the single-process run
    assert rel_l2(np.concatenate([p['g_Y'] for p in parts], axis=3), p0['g1_Y']) < 1e-9
    for p in parts:
        assert int(p['g_k']) == int(p0['g1_k'])
        for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho'):
            assert rel_l2(p['g_' + f], p0['g1_' + f]) < 1e-9, f
    # 2. device-driven loop, early stop, unequal host lag: same stopping iteration everywhere,
    #    collectives aligned afterwards, identical records on every rank
    k1 = int(p0['d1_k'])
    assert 3 < k1 < 24
    for p in parts:
        assert int(p['d_k']) == k1
        assert float(p['d_after']) == world * (world + 1) / 2
        for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
            assert np.array_equal(p['d_' + f], p0['d_' + f]), f
            assert rel_l2(p['d_' + f], p0['d1_' + f]) < 1e-5, f
    assert rel_l2(np.concatenate([p['d_Y'] for p in parts], axis=3), p0['d1_Y']) < 1e-4
    # 3. dictionary learning with either dictionary update
    for dm in ('pgm', 'cns'):
        t, t1 = 'dl%s_' % dm, 'dl%s1_' % dm
        assert rel_l2(np.concatenate([p[t + 'X'] for p in parts], axis=3), p0[t1 + 'X']) < 1e-8, dm
        for p in parts:
            assert rel_l2(p[t + 'D'], p0[t1 + 'D']) < 1e-9, dm
            assert np.array_equal(p[t + 'D'], p0[t + 'D'])
            for f in p0.files:
                if f.startswith(t1) and f[len(t1):] not in ('D', 'X', 'Cnstr'):
                    a, c = p[t + f[len(t1):]], p0[f]
                    assert rel_l2(a, c) < 1e-8 or np.max(np.abs(a - c)) < 1e-12, (dm, f)


def test_bench_multi_rank_fields_under_gloo(tmp_path):
    """`bench.py --gpus 2` end to end on CPU: two gloo ranks on the simulator build, a tiny
    workload -- the per-rank spread and all-reduce cost fields the scaling runs rely on exist and are
    sane, and the line is the one-line JSON contract."""
    build_hostsim()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1', SPORCO_AMD_LIBRARY=HOSTSIM_LIB,
               SPORCO_AMD_BENCH_BACKEND='gloo')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
           '--master-addr', '127.0.0.1', '--master-port', '29641', os.path.join(REPO, 'bench.py'),
           '--gpus', '2', '--steps', '3', '--warmup', '1', '--size', '128', '--filters', '8', '--images', '1',
           '--steady-steps', '4', '--no-cpu-baseline', '--no-parity', '--no-time-to-tol', '--configs', 'config3',
           '--tiny']
    r = subprocess.run(cmd, env=env, timeout=1500, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['scaling'] == 'weak' and d['value'] > 0
    rk = d['ranks']
    assert rk['backend'] == 'gloo' and rk['reducer'] == 'TorchReducer'
    assert rk['ms_per_step_slowest_rank'] >= rk['ms_per_step_fastest_rank'] > 0
    assert rk['allreduce_calls_timed'] > 0 and rk['allreduce_ms_max'] >= rk['allreduce_ms_mean_worst_rank'] >= 0
    assert d['frac_check']['violations'] == []
    # the N = 1 figure of the same invocation (every rank alone, no reducer) beside the sharded one
    assert rk['unsharded_ms_per_step_slowest_rank'] >= rk['unsharded_ms_per_step_fastest_rank'] > 0
    assert rk['sharded_over_unsharded_time'] > 0
    # BASELINE configs[2] under the reducer: ConvBPDNJoint, images sharded over the two ranks
    c3 = d['configs']['config3_sharded']
    assert 'error' not in c3, c3
    assert c3['fused_kernels_engaged'] and c3['value'] > 0 and c3['reducer'] == 'TorchReducer'
    assert c3['ms_per_step'] >= c3['ms_per_step_fastest_rank'] > 0
    # the compact summary is the LAST key of the line (a tail of the line keeps it)
    assert list(d)[-1] == 'summary' and lines[0].rstrip().endswith('}}')
    sm = d['summary']
    assert sm['value'] == d['value'] and sm['roofline_frac'] == d['roofline']['frac']
    assert sm['configs'] == {'config3_sharded': round(c3['value'], 2)}
    assert not any('algorithmic_frac' in k for k in json.dumps(d).split('"'))
