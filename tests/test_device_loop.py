"""The device-driven solve (sporco_amd_csc_admm_run): residuals, tolerances, the adaptive
penalty parameter and the stopping test evaluated on the device after every iteration
(restating sporco/admm/admm.py:462-486, 549-575, 375-377 as sporco_amd/admm/admm.py evaluates
them on the host), the iteration kernels reading rho, lambda/rho and the pending U scale from
device memory.  The contract: iterates, statistics and the stopping iteration are IDENTICAL
(bit for bit) to the host-driven loop of the same library (SPORCO_AMD_HOST_LOOP=1), which in
turn is pinned to the reference by the fixtures of test_admm_cbpdn.py / test_fused_xstep.py.
"""

import os

import numpy as np
import pytest

from test_fused_xstep import problem

FIELDS = ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho')


def run(D, S, optd, host, lag=0, lmbda=0.05):
    from sporco_amd.admm import cbpdn
    env = {'SPORCO_AMD_HOST_LOOP': '1'} if host else {}
    if lag:
        env['SPORCO_AMD_RUN_LAG'] = str(lag)
    os.environ.update(env)
    try:
        b = cbpdn.ConvBPDN(D, S, lmbda, cbpdn.ConvBPDN.Options(optd))
        Y = b.solve()
    finally:
        for k in env:
            os.environ.pop(k, None)
    return b, Y


def same_stats(b0, b1):
    i0, i1 = b0.getitstat(), b1.getitstat()
    for f in FIELDS:
        a0, a1 = np.asarray(getattr(i0, f), float), np.asarray(getattr(i1, f), float)
        assert a0.shape == a1.shape, f
        assert np.array_equal(a0, a1, equal_nan=True), f
    assert list(i0.Iter) == list(i1.Iter)
    assert np.all(np.diff(np.asarray(i1.Time, float)) >= 0) and i1.Time[0] > 0


CASES = [
    {'MaxMainIter': 8, 'RelStopTol': 0.0},
    {'MaxMainIter': 40, 'RelStopTol': 5e-2},                       # stops early
    {'MaxMainIter': 7, 'RelStopTol': 0.0, 'AutoRho': {'Period': 3}},
    {'MaxMainIter': 12, 'RelStopTol': 1e-2, 'AbsStopTol': 1e-4,
     'AutoRho': {'StdResiduals': True, 'AutoScaling': False, 'Scaling': 1.5, 'RsdlRatio': 1.1}},
    {'MaxMainIter': 6, 'RelStopTol': 0.0, 'NonNegCoef': True, 'NoBndryCross': True,
     'AuxVarObj': False, 'gEvalY': True},
]


@pytest.mark.gpu
@pytest.mark.parametrize('case', range(len(CASES)))
def test_device_loop_is_bit_identical_to_host_loop(gpu_backend, case):
    backend = gpu_backend
    H, W, K, N = 256, 256, 4, (1 if backend == 'hostsim' else 3)
    D, S = problem(H, W, K, N, seed=60 + case)
    optd = CASES[case]
    b0, Y0 = run(D, S, optd, host=True)
    b1, Y1 = run(D, S, optd, host=False)
    assert b1._device_loop_ok() and b0._dev.uses_fused_rows()
    assert b0.k == b1.k and (case != 1 or b0.k < 40)
    assert np.array_equal(Y0, Y1) and np.array_equal(b0.U, b1.U)
    assert float(b0.rho) == float(b1.rho) and b0._u_scale == b1._u_scale
    same_stats(b0, b1)
    # X (rebuilt on demand from the previous iterate) and a continued solve
    assert np.array_equal(b0.X, b1.X)
    b0.solve()
    b1.solve()
    assert np.array_equal(b0.Y, b1.Y) and b0.k == b1.k
    same_stats(b0, b1)


@pytest.mark.parametrize('lag', [1, pytest.param(2, marks=pytest.mark.gpu),
                                 pytest.param(3, marks=pytest.mark.gpu)])
def test_launches_past_the_stopping_iteration_do_nothing(backend, lag):
    """The host enqueues a few iterations ahead of the last record it has seen; those that
    land behind the stopping iteration must leave every array untouched (and the ping-pong
    buffer roles are put right afterwards).  (On the CPU simulator: a loose tolerance, so
    that the run is a handful of iterations.)"""
    D, S = problem(128 if backend == 'hostsim' else 256, 256, 4, 1, seed=71)
    optd = {'MaxMainIter': 40, 'RelStopTol': 0.25 if backend == 'hostsim' else 5e-2}
    b0, Y0 = run(D, S, optd, host=True)
    b1, Y1 = run(D, S, optd, host=False, lag=lag)
    assert 1 < b0.k == b1.k < 40
    assert np.array_equal(Y0, Y1) and np.array_equal(b0.U, b1.U) and np.array_equal(b0.X, b1.X)
    same_stats(b0, b1)


@pytest.mark.gpu
def test_fastsolve_and_weights(gpu_backend):
    D, S = problem(256, 256, 4, 2, seed=81)
    rng = np.random.RandomState(4)
    wl1 = (0.5 + rng.rand(256, 256, 1, 1, 4)).astype(np.float32)
    for optd in ({'MaxMainIter': 5, 'FastSolve': True, 'AutoRho': {'Enabled': False}, 'rho': 2.0},
                 {'MaxMainIter': 5, 'FastSolve': True, 'RelStopTol': 0.0},     # rho adapts, no stats
                 {'MaxMainIter': 4, 'RelStopTol': 0.0, 'L1Weight': wl1}):
        b0, Y0 = run(D, S, optd, host=True)
        b1, Y1 = run(D, S, optd, host=False)
        assert np.array_equal(Y0, Y1) and float(b0.rho) == float(b1.rho)
        assert len(b1.itstat) == len(b0.itstat)
        if b0.itstat:
            same_stats(b0, b1)


def test_host_loop_is_kept_when_the_host_has_to_see_iterations(backend):
    """A callback (or Verbose, or an overridden step) needs the per-iteration loop."""
    from sporco_amd.admm import cbpdn
    H = 256 if backend == 'gpu' else 48      # (the CPU simulator: a shape off the fused path)
    D, S = problem(H, H, 4, 1, seed=91)
    seen = []

    def cb(obj):
        seen.append(obj.k)
        return obj.k == 2

    opt = cbpdn.ConvBPDN.Options({'MaxMainIter': 9, 'RelStopTol': 0.0, 'Callback': cb})
    b = cbpdn.ConvBPDN(D, S, 0.05, opt)
    assert not b._device_loop_ok()
    b.solve()
    assert seen == [0, 1, 2] and b.k == 3

    class Patched(cbpdn.ConvBPDN):
        def update_rho(self, k, r, s):
            pass
    p = Patched(D, S, 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': 3, 'RelStopTol': 0.0}))
    assert not p._device_loop_ok()
    p.solve()
    assert len(set(p.getitstat().Rho)) == 1
    # shapes outside the three-launch path run the host loop inside solve() unchanged
    g = cbpdn.ConvBPDN(D, S[:40, :40], 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': 3,
                                                                   'RelStopTol': 0.0}))
    assert not g._dev.uses_fused_rows()
    g.solve()
    assert g.k == 3 and len(g.itstat) == 3


def test_host_methods_the_device_loop_never_calls_keep_the_host_loop(backend):
    """`rhochange` (the reference's documented hook, sporco/admm/admm.py:575), `iteration`,
    the display methods: an override of any of them -- on the class or on the instance --
    must run, so the solve stays on the per-iteration loop."""
    from sporco_amd.admm import cbpdn
    H = 128
    D, S = problem(H, H, 4, 1, seed=92)
    optd = {'MaxMainIter': 6, 'RelStopTol': 0.0, 'AutoRho': {'Period': 2}}
    plain = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    assert plain._device_loop_ok()
    calls = []

    class WithHook(cbpdn.ConvBPDN):
        def rhochange(self):
            calls.append(self.k)
    b = WithHook(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    assert not b._device_loop_ok()
    b.solve()
    # update_rho acts at k = 1, 3, 5 ((k + 1) % Period == 0, k != 0) and calls the hook whenever
    # it changes rho
    assert set(calls) <= {1, 3, 5}
    assert calls, "rhochange() was never called"
    Y0 = plain.solve()
    assert np.array_equal(Y0, b.Y)           # the hook does nothing: same iterates
    # instance-level overrides
    c = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    c.display_status = lambda fmtstr, itst: None
    assert not c._device_loop_ok()
    d = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    d.iteration = d.iteration
    assert not d._device_loop_ok()


def test_pickle_from_before_the_signal_property(backend):
    """A state pickled when `S` was a plain attribute restores (the key is migrated)."""
    from sporco_amd.admm import cbpdn
    D, S = problem(48, 48, 4, 1, seed=93)
    b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options({'MaxMainIter': 2, 'RelStopTol': 0.0}))
    b.solve()
    st = b.__getstate__()
    st['S'] = st.pop('_S_host')
    st.pop('_S_dev')
    r = cbpdn.ConvBPDN.__new__(cbpdn.ConvBPDN)
    r.__setstate__(st)
    assert np.array_equal(r.S, b.S) and np.array_equal(r.Y, b.Y)
