"""The one-launch solve of small problems (csc_rows.h admm_persist: a run of iterations inside
one kernel launch, barriers across the grid between the passes) against the launch-per-pass
device loop it replaces (sporco/admm/admm.py:331-377 either way).  The passes are the same device
functions, the sums are reduced in the same order and the control update is the same code: on
the simulator the two agree bit for bit.  On the GPU the column pass is compiled inside another
kernel and the compiler contracts other multiply-add pairs of its butterflies, so there the two
agree to float32 rounding (same stopping iteration, iterates and statistics to 2e-6 over a
few iterations, 2e-5 over a run to the stopping tolerance)."""

import os

import numpy as np
import pytest

from conftest import rel_l2

FIELDS = ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho')


def solve(D, S, lmbda, optd, persist, again=0):
    from sporco_amd import _lib
    from sporco_amd.admm import cbpdn
    old = os.environ.get('SPORCO_AMD_PERSIST')
    os.environ['SPORCO_AMD_PERSIST'] = '1' if persist else '0'
    try:
        b = cbpdn.ConvBPDN(D, S, lmbda, cbpdn.ConvBPDN.Options(optd))
        b.solve()
        for _ in range(again):
            b.solve()
        runs = b._dev.query(_lib.QUERY_PERSIST_RUNS)
    finally:
        if old is None:
            del os.environ['SPORCO_AMD_PERSIST']
        else:
            os.environ['SPORCO_AMD_PERSIST'] = old
    return b, runs


def same(a, b, exact, tol=2e-6):
    assert a.k == b.k
    for name in ('Y', 'U', 'X'):
        if exact:
            assert np.array_equal(getattr(a, name), getattr(b, name)), name
        else:
            assert rel_l2(getattr(a, name), getattr(b, name)) < tol, name
    ia, ib = a.getitstat(), b.getitstat()
    if ia is not None and len(ia.Iter):
        for f in FIELDS:
            va, vb = np.asarray(getattr(ia, f)), np.asarray(getattr(ib, f))
            assert np.array_equal(va, vb) if exact else rel_l2(va, vb) < tol, f
    assert float(a.rho) == float(b.rho) if exact else abs(float(a.rho) / float(b.rho) - 1) < tol


CASES = [
    ({'MaxMainIter': 8}, 0),                                                  # rho moves: rows_fwd runs
    ({'MaxMainIter': 6, 'NonNegCoef': True}, 0),
    ({'MaxMainIter': 5, 'rho': 2.0, 'AutoRho': {'Enabled': False}}, 0),       # every spectrum emitted
    ({'MaxMainIter': 5, 'FastSolve': True, 'AutoRho': {'Enabled': False}}, 0),
    ({'MaxMainIter': 40, 'RelStopTol': 5e-2}, 0),                             # stops inside the launch
    ({'MaxMainIter': 4, 'AuxVarObj': True, 'RelaxParam': 1.0}, 1),            # and a second solve()
]


@pytest.mark.parametrize('case', range(len(CASES)))
def test_one_launch_equals_launch_per_pass(backend, case):
    optd, again = CASES[case]
    rng = np.random.RandomState(3 + case)
    D = rng.randn(6, 6, 4).astype(np.float32)
    S = rng.randn(128, 128, 1).astype(np.float32)
    a, ra = solve(D, S, 0.05, optd, False, again)
    b, rb = solve(D, S, 0.05, optd, True, again)
    if optd.get('AuxVarObj'):
        assert rb == 0          # the data fidelity at Y is not part of the one-launch form
    else:
        assert ra == 0 and rb == 1 + again
    same(a, b, backend == 'hostsim')


def test_what_keeps_the_launch_per_pass_loop(backend):
    """Options and shapes outside the one-launch form run as before."""
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(0)
    D = rng.randn(6, 6, 4).astype(np.float32)
    S = rng.randn(128, 128, 1).astype(np.float32)
    for optd in ({'MaxMainIter': 4, 'NoBndryCross': True},
                 {'MaxMainIter': 4, 'L1Weight': rng.rand(1, 1, 1, 1, 4).astype(np.float32)},
                 {'MaxMainIter': 2}):
        b, runs = solve(D, S, 0.05, optd, True)
        assert runs == 0 and b.k == optd['MaxMainIter']
    S2 = rng.randn(128, 256, 1).astype(np.float32)          # not square
    b, runs = solve(D, S2, 0.05, {'MaxMainIter': 4}, True)
    assert runs == 0


@pytest.mark.gpu
@pytest.mark.parametrize('H,K,N', [(256, 32, 1), (256, 64, 1), (128, 64, 4), (256, 8, 3)])
def test_one_launch_at_config1_shapes(gpu_backend, H, K, N):
    """BASELINE config 1 (256 x 256, K = 32, N = 1) and neighbours, to the stopping iteration:
    both loops against each other (bit-identical), and the first against the float64 oracle."""
    from oracle import cbpdn_oracle as orc
    rng = np.random.RandomState(H + K + N)
    D = rng.randn(8, 8, K).astype(np.float32)
    S = rng.randn(H, H, N).astype(np.float32)
    optd = {'MaxMainIter': 60, 'RelStopTol': 5e-3}
    a, ra = solve(D, S, 0.1, optd, False)
    b, rb = solve(D, S, 0.1, optd, True)
    assert ra == 0 and rb == 1
    same(a, b, False, 2e-5)      # (tens of iterations of rounding differences)
    if H * H * K * N <= (1 << 21):
        r = orc.admm_cbpdn(D.reshape(8, 8, 1, 1, K).astype(np.float64),
                           S.reshape(H, H, 1, N, 1).astype(np.float64), 0.1, dtype=np.float64,
                           maxiter=60, rel_tol=5e-3)
        assert r['iters'] == b.k
        assert rel_l2(b.Y, r['Y']) < 1e-4
        for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
            assert rel_l2(getattr(b.getitstat(), f), r[f]) < 1e-4, f
