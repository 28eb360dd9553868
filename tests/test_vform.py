"""The single-array state of the fused ADMM iteration (csc_rows.h, "V form").

The row epilogue forms V' = AX + U (scaled), Y' = prox(V'), U' = V' - Y' (sporco/admm/cbpdn.py:
614-620, sporco/prox/_lp.py:144-183, sporco/admm/admm.py:434-437): the new iterate is a
function of V' alone, so runs of several fused iterations store V' instead of (Y', U') and
derive Y, U from it wherever they are read -- seven X-sized passes per iteration instead of
ten.  The contract tested here: iterates (Y, U, X, Yprev, AX), statistics and stopping
iterations are IDENTICAL, bit for bit, to the (Y, U) form of the same library
(SPORCO_AMD_NO_VFORM=1), which the fixtures of test_admm_cbpdn.py / test_fused_xstep.py /
test_parity_baseline_shapes.py pin to the reference; plus the float64 oracle directly.
"""

import os

import numpy as np
import pytest

from conftest import rel_l2
from test_fused_xstep import problem

FIELDS = ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho')


def run(D, S, optd, vform, host=False, lmbda=0.05, calls=1, joint_mu=None):
    from sporco_amd import _lib
    from sporco_amd.admm import cbpdn
    env = {}
    if not vform:
        env['SPORCO_AMD_NO_VFORM'] = '1'
    if host:
        env['SPORCO_AMD_HOST_LOOP'] = '1'
    os.environ.update(env)
    try:
        if joint_mu is not None:
            b = cbpdn.ConvBPDNJoint(D, S, lmbda, joint_mu, cbpdn.ConvBPDNJoint.Options(optd))
        else:
            b = cbpdn.ConvBPDN(D, S, lmbda, cbpdn.ConvBPDN.Options(optd))
        live = []
        for _ in range(calls):
            b._return_min = False         # (solve() without fetching Y: the state stays put)
            b.solve()
            live.append(b._dev.query(_lib.QUERY_VFORM_LIVE))
        out = dict(Y=b.Y.copy(), U=b.U.copy(), X=b.X.copy(), Yprev=b._fetch(_lib.VAR_YPREV).copy(),
                   AX=b._fetch(_lib.VAR_AX).copy(), k=b.k, live=live,
                   stats={f: np.asarray(getattr(b.getitstat(), f), float) for f in FIELDS}
                   if b.itstat else {})
    finally:
        for k in env:
            os.environ.pop(k, None)
    return b, out


def same(o0, o1):
    assert o0['k'] == o1['k']
    for f in ('Y', 'U', 'X', 'Yprev', 'AX'):
        assert np.array_equal(o0[f], o1[f]), f
    assert sorted(o0['stats']) == sorted(o1['stats'])
    for f in o0['stats']:
        assert np.array_equal(o0['stats'][f], o1['stats'][f], equal_nan=True), f


CASES = {
    'default': {'MaxMainIter': 9, 'RelStopTol': 0.0},
    'nonneg_period3': {'MaxMainIter': 8, 'RelStopTol': 0.0, 'NonNegCoef': True,
                       'AutoRho': {'Period': 3}},
    'fixed_rho_fast': {'MaxMainIter': 7, 'RelStopTol': 0.0, 'FastSolve': True,
                       'AutoRho': {'Enabled': False}, 'rho': 2.5},
    'rlx1_std': {'MaxMainIter': 6, 'RelStopTol': 0.0, 'RelaxParam': 1.0,
                 'AutoRho': {'StdResiduals': True}},
    'stops_at_once': {'MaxMainIter': 12, 'RelStopTol': 10.0},          # one iteration
    'stops_early': {'MaxMainIter': 40, 'RelStopTol': 5e-2},
}


@pytest.mark.parametrize('case', sorted(CASES))
@pytest.mark.parametrize('host', [False, True])
def test_v_form_is_bit_identical_to_the_yu_form(backend, case, host):
    H = 256 if backend == 'gpu' else 128
    K, N = (16, 3) if backend == 'gpu' else (4, 2)
    D, S = problem(H, H, K, N, seed=11)
    optd = CASES[case]
    b0, o0 = run(D, S, optd, vform=False, host=host)
    b1, o1 = run(D, S, optd, vform=True, host=host)
    assert b1._dev.uses_fused_rows()
    assert o0['live'] == [0]
    # the device-driven run enters the V form at once (MaxMainIter >= 4); the host-driven loop
    # from its second iteration on
    if o1['k'] >= 2 or not host:
        assert o1['live'] == [1], (case, host, o1['k'])
    same(o0, o1)


def test_v_form_survives_restarts_and_reads_in_between(backend):
    """solve() three times on one object (the second call starts from the V left by the first);
    reading Y between calls takes the handle back to the (Y, U) form and on again."""
    H = 256 if backend == 'gpu' else 128
    D, S = problem(H, H, 4, 2, seed=12)
    optd = {'MaxMainIter': 5, 'RelStopTol': 0.0}
    b0, o0 = run(D, S, optd, vform=False, calls=3)
    b1, o1 = run(D, S, optd, vform=True, calls=3)
    assert o1['live'] == [1, 1, 1] and o1['k'] == 15
    same(o0, o1)
    from sporco_amd import _lib
    from sporco_amd.admm import cbpdn
    b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    for _ in range(3):
        Y = b.solve()                     # (fetches Y: back to (Y, U))
        assert b._dev.query(_lib.QUERY_VFORM_LIVE) == 0
    assert np.array_equal(Y, o0['Y']) and np.array_equal(b.U, o0['U'])
    assert np.array_equal(b.X, o0['X'])


def test_v_form_against_the_oracle(backend):
    from oracle import cbpdn_oracle as orc
    H = 256 if backend == 'gpu' else 128
    K, N = 8, 2
    D, S = problem(H, H, K, N, seed=13)
    optd = {'MaxMainIter': 12, 'RelStopTol': 0.0}
    b, o = run(D, S, optd, vform=True)
    assert o['live'] == [1]
    ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, H, 1, N, 1), 0.05,
                         dtype=np.float64, maxiter=12, rel_tol=0.0)
    assert rel_l2(o['Y'], ref['Y']) < 1e-4
    assert rel_l2(o['U'], ref['U']) < 1e-4
    assert rel_l2(o['X'], ref['X']) < 1e-4
    for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(o['stats'][f], ref[f]) < 1e-3, f


@pytest.mark.parametrize('host', [False, True])
@pytest.mark.parametrize('case', ['default', 'nonneg_period3', 'stops_early', 'fixed_rho'])
def test_joint_v_form_is_bit_identical_and_matches_the_oracle(backend, case, host):
    """ConvBPDNJoint: Y = prox_sl1l2(V) over the channels (sporco/admm/cbpdn.py:785-794,
    sporco/prox/_l21.py:51-88) re-derived from V by rows_fwd (joint tiling) and by the joint
    epilogue; the materialised (Y, U) through the joint split kernel."""
    from oracle import cbpdn_oracle as orc
    H = 256 if backend == 'gpu' else 128
    C, N, K = 3, (2 if backend == 'gpu' else 1), 32
    if backend == 'hostsim' and (case == 'stops_early' or (host and case not in ('default', 'fixed_rho'))):
        pytest.skip("kept short on the CPU simulator")
    D, S = problem(H, H, K, N, seed=21, C=C)
    # ('fixed_rho': the emitting joint epilogue runs from the third iteration on)
    optd = dict(CASES[case]) if case != 'fixed_rho' else \
        {'MaxMainIter': 7, 'RelStopTol': 0.0, 'AutoRho': {'Enabled': False}, 'rho': 4.0}
    if backend == 'hostsim':
        optd['MaxMainIter'] = min(optd['MaxMainIter'], 6)
    b0, o0 = run(D, S, optd, vform=False, host=host, lmbda=0.1, joint_mu=0.02)
    b1, o1 = run(D, S, optd, vform=True, host=host, lmbda=0.1, joint_mu=0.02)
    assert b1._dev.uses_fused_rows() and b1._fused_ok()
    assert o0['live'] == [0]
    if o1['k'] >= 2 or not host:
        assert o1['live'] == [1]
    same(o0, o1)
    if case in ('default', 'fixed_rho') and not host:
        n = optd['MaxMainIter']
        kw = {} if case == 'default' else {'rho': 4.0, 'auto_rho': False}
        ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, H, C, N, 1), 0.1, mu=0.02,
                             dtype=np.float64, maxiter=n, rel_tol=0.0, **kw)
        assert rel_l2(o1['Y'], ref['Y']) < 1e-4
        assert rel_l2(o1['U'], ref['U']) < 1e-4
        for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
            assert rel_l2(o1['stats'][f], ref[f]) < 1e-3, f


@pytest.mark.parametrize('opt_case', ['l1weight', 'nobndry', 'nobndry_nonneg_weight'])
@pytest.mark.parametrize('host', [False, True])
def test_v_form_under_weights_and_nobndrycross(backend, opt_case, host):
    """An L1Weight array, NoBndryCross: the derivation of Y from V repeats the weight, the clamp and
    the boundary band of the row epilogue (sporco/admm/cbpdn.py:297-311, 614-620)."""
    from oracle import cbpdn_oracle as orc
    if backend == 'hostsim' and host and opt_case != 'l1weight':
        pytest.skip("kept short on the CPU simulator")
    H = 256 if backend == 'gpu' else 128
    K, N = (16, 3) if backend == 'gpu' else (4, 2)
    D, S = problem(H, H, K, N, seed=15)
    w = (0.5 + np.abs(np.random.RandomState(3).randn(H, H, 1, 1, K))).astype(np.float32)
    extra = {'l1weight': {'L1Weight': w}, 'nobndry': {'NoBndryCross': True},
             'nobndry_nonneg_weight': {'NoBndryCross': True, 'NonNegCoef': True, 'L1Weight': w}}[opt_case]
    optd = dict({'MaxMainIter': 7, 'RelStopTol': 0.0}, **extra)
    b0, o0 = run(D, S, optd, vform=False, host=host)
    b1, o1 = run(D, S, optd, vform=True, host=host)
    assert b1._dev.uses_fused_rows() and b1._fused_ok()
    assert o0['live'] == [0] and o1['live'] == [1]
    same(o0, o1)
    if not host:
        ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, H, 1, N, 1), 0.05,
                             dtype=np.float64, maxiter=7, rel_tol=0.0,
                             wl1=extra.get('L1Weight', 1.0), nonneg='NonNegCoef' in extra,
                             nobndry='NoBndryCross' in extra)
        assert rel_l2(o1['Y'], ref['Y']) < 1e-4 and rel_l2(o1['U'], ref['U']) < 1e-4


def test_v_form_under_addmasksim_and_gradreg(backend):
    """AddMaskSim (the impulse slice: no shrinkage, masked) and ConvBPDNGradReg run the V form
    too; against the (Y, U) form of the same library, bit for bit."""
    from sporco_amd import _lib
    from sporco_amd.admm import cbpdn
    H = 256 if backend == 'gpu' else 128
    K, N = (7, 2) if backend == 'gpu' else (3, 1)
    D, S = problem(H, H, K, N, seed=16)
    W = (np.random.RandomState(4).rand(H, H, N) > 0.3).astype(np.float32)
    outs = {}
    for vform in (False, True):
        if not vform:
            os.environ['SPORCO_AMD_NO_VFORM'] = '1'
        try:
            opt = cbpdn.ConvBPDN.Options({'MaxMainIter': 6, 'RelStopTol': 0.0})
            a = cbpdn.AddMaskSim(cbpdn.ConvBPDN, D, S, W, 0.05, opt=opt)
            Ya = a.solve()
            live_a = a.cbpdn._dev.query(_lib.QUERY_VFORM_LIVE) if hasattr(a, 'cbpdn') else None
            og = cbpdn.ConvBPDNGradReg.Options({'MaxMainIter': 6, 'RelStopTol': 0.0})
            g = cbpdn.ConvBPDNGradReg(D, S, 0.05, 0.1, og)
            g._return_min = False
            g.solve()
            live_g = g._dev.query(_lib.QUERY_VFORM_LIVE)
            outs[vform] = (Ya, a.getitstat(), g.Y.copy(), g.U.copy(), g.getitstat(), live_g)
        finally:
            os.environ.pop('SPORCO_AMD_NO_VFORM', None)
    assert outs[False][5] == 0 and outs[True][5] == 1
    assert np.array_equal(outs[False][0], outs[True][0])
    assert np.array_equal(np.asarray(outs[False][1].ObjFun), np.asarray(outs[True][1].ObjFun))
    assert np.array_equal(outs[False][2], outs[True][2]) and np.array_equal(outs[False][3], outs[True][3])
    assert np.array_equal(np.asarray(outs[False][4].ObjFun), np.asarray(outs[True][4].ObjFun))


GENERIC_CASES = {
    # (H, W, dtype, multi-channel dictionary, options)
    'f64_default': (64, 64, np.float64, False, {'MaxMainIter': 9, 'RelStopTol': 0.0}),
    'f64_nonneg_period3': (32, 64, np.float64, False,
                           {'MaxMainIter': 8, 'RelStopTol': 0.0, 'NonNegCoef': True,
                            'AutoRho': {'Period': 3}}),
    'f32_48x40_fixed_rho': (48, 40, np.float32, False,
                            {'MaxMainIter': 7, 'RelStopTol': 0.0, 'AutoRho': {'Enabled': False},
                             'rho': 2.5}),
    'f32_36x60_stops_early': (36, 60, np.float32, False, {'MaxMainIter': 40, 'RelStopTol': 5e-2}),
    'f64_mcdict': (32, 32, np.float64, True, {'MaxMainIter': 6, 'RelStopTol': 0.0}),
}


@pytest.mark.parametrize('case', sorted(GENERIC_CASES))
def test_generic_chain_v_form_is_bit_identical_to_the_yu_form(backend, case):
    """The generic chain (sizes / precisions the register kernels do not cover) keeps the single
    array too: its row transform derives Y - s U from V while loading, its epilogue reads V and
    writes V' in place (fft.hip r2c loader, ck_admm.hip admm_post_kernel).  Same contract: the
    (Y, U) form of the same library, bit for bit; that form is what the float64 fixtures of
    test_admm_cbpdn.py pin to the reference."""
    from sporco_amd import _lib
    H, W, dt, mc, optd = GENERIC_CASES[case]
    K, N = 5, 2
    rng = np.random.RandomState(31)
    if mc:
        D = rng.randn(4, 4, 3, K).astype(dt)
        S = rng.randn(H, W, 3, N).astype(dt)
    else:
        D = rng.randn(4, 4, K).astype(dt)
        S = rng.randn(H, W, N).astype(dt)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    optd = dict(optd, DataType=dt)

    def go(vform, calls=1):
        from sporco_amd.admm import cbpdn
        if not vform:
            os.environ['SPORCO_AMD_NO_VFORM'] = '1'
        os.environ['SPORCO_AMD_HOST_LOOP'] = '1'
        try:
            b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
            live = []
            for _ in range(calls):
                b._return_min = False
                b.solve()
                live.append(b._dev.query(_lib.QUERY_VFORM_LIVE))
            return b, dict(Y=b.Y.copy(), U=b.U.copy(), X=b.X.copy(), k=b.k, live=live,
                           stats={f: np.asarray(getattr(b.getitstat(), f), float) for f in FIELDS})
        finally:
            os.environ.pop('SPORCO_AMD_NO_VFORM', None)
            os.environ.pop('SPORCO_AMD_HOST_LOOP', None)

    for calls in (1, 2):
        b0, o0 = go(False, calls)
        b1, o1 = go(True, calls)
        assert not b1._dev.uses_fused_rows() or mc or dt == np.float64
        assert o0['live'] == [0] * calls
        if o1['k'] >= 3 * calls:
            assert o1['live'] == [1] * calls, (case, o1['k'])
        assert o0['k'] == o1['k']
        # Element for element the two forms perform the same operations; since round 5 the sums of the
        # V form come out of the fused row pass (fft.h fft_c2r_vpost), i.e. the same double-precision
        # terms added in another order, so rho -- formed from them -- may differ in its last bit and
        # the iterates with it: equal to rounding of the working precision, not bit for bit.
        # (float64 only: the float32 chain keeps its three kernels and stays bit for bit)
        tol = 1e-12 if dt == np.float64 else 0.0
        for f in ('Y', 'U', 'X'):
            assert np.array_equal(o0[f], o1[f]) or rel_l2(o1[f], o0[f]) < tol, f
        # (the sums of the fused row pass -- fft.h fft_c2r_vpost, round 5 -- are reduced over other
        # workgroups than the epilogue kernel's: the same double-precision terms in another order)
        for f in o0['stats']:
            assert np.allclose(o0['stats'][f], o1['stats'][f], rtol=(1e-10 if dt == np.float64 else 0.0), atol=0.0, equal_nan=True), f
    # reading the iterates takes the handle back to the (Y, U) form
    assert b1._dev.query(_lib.QUERY_VFORM_LIVE) == 0
