"""The fused FISTA iteration (csc_pgm.hip + csc_rows.hip: three launches, tile-major
spectral iterates, X rebuilt on demand) against the NumPy oracle and against the
generic composition of the same library.  Engages for float32, H and W in
{256, 512}, even K <= 256 and default policies or BacktrackStandard (no step-size policy /
monotone restart); anything else composes the staged calls.

Tolerance: 1e-5 relative l2 against the float64 oracle after 4 iterations (observed
5e-7); the generic float32 path agrees with the fused one to the same level.
"""

import os
import pickle

import numpy as np
import pytest

from conftest import rel_l2
from test_fused_xstep import problem


def make(D, S, optd, generic=False):
    from sporco_amd.pgm import cbpdn as pc
    if generic:
        os.environ['SPORCO_AMD_OLD_ROWS'] = '1'
        os.environ['SPORCO_AMD_NO_PAD'] = '1'
    try:
        return pc.ConvBPDN(D, S, 0.05, pc.ConvBPDN.Options(optd))
    finally:
        os.environ.pop('SPORCO_AMD_OLD_ROWS', None)
        os.environ.pop('SPORCO_AMD_NO_PAD', None)


@pytest.mark.parametrize('H,W,K,N', [(256, 256, 4, 1), (256, 512, 6, 1),
                                     (256, 128, 4, 2),      # W = 128: the 32 x 4 row kernels
                                     (128, 128, 4, 2),      # H = 128: the 32 x 4 column kernels
                                     # H = 512, K = 64: the persistent column kernels (65 tiles
                                     # over the simulator's 8 workgroups; 256 on the GPU)
                                     (512, 128, 64, 1),
                                     pytest.param(256, 256, 5, 2, marks=pytest.mark.gpu),
                                     # K > 64: cooperating slab workgroups in the gradient step,
                                     # per-slab momentum kernels + the slab statistics kernel
                                     (256, 256, 128, 1),
                                     pytest.param(512, 512, 128, 1, marks=pytest.mark.gpu),
                                     pytest.param(512, 256, 250, 1, marks=pytest.mark.gpu),
                                     pytest.param(256, 512, 192, 2, marks=pytest.mark.gpu)])
def test_fused_pgm_matches_oracle(backend, H, W, K, N):
    from oracle import cbpdn_oracle as orc
    if backend == 'hostsim' and W == 512:
        pytest.skip("W = 512 row kernels run under the simulator in test_fused_xstep; here GPU only")
    D, S = problem(H, W, K, N, seed=H + W)
    slow = backend == 'hostsim' and (K > 64 or H * K >= 512 * 64)   # (keeps the CPU suite short)
    iters = 2 if slow else 3
    optd = {'MaxMainIter': iters, 'RelStopTol': 0.0, 'L': 50.0}
    b = make(D, S, optd)
    assert b.dev.uses_fused_rows() and b._fused_ok()
    X = b.solve()
    ref = orc.pgm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05,
                        dtype=np.float64, maxiter=iters, L=50.0, rel_tol=0.0)
    assert rel_l2(X, ref['X']) < 1e-5
    its = b.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'Rsdl'):
        assert rel_l2(getattr(its, f), ref[f]) < 1e-5, f
    # the spectral iterates come back in the reference layout
    assert rel_l2(b.Xf, np.fft.rfftn(ref['X'], axes=(0, 1))) < 1e-5
    if slow:
        return
    if backend == 'hostsim':
        # after the layout round trip the solver continues in step with the oracle
        b.solve()
        ref8 = orc.pgm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05,
                             dtype=np.float64, maxiter=6, L=50.0, rel_tol=0.0)
        assert rel_l2(b.X, ref8['X']) < 1e-5
        return       # (the comparison with the generic composition runs on the GPU)
    b0 = make(D, S, optd, generic=True)
    assert not b0.dev.uses_fused_rows()
    b0.solve()
    for name in ('Yf', 'Xfprv', 'Yfprv'):
        assert rel_l2(getattr(b, name), getattr(b0, name)) < 1e-5, name
    # X is exactly sparse: it is the prox output itself, not a transform of Xf (a value on
    # the threshold may round either way in two float32 implementations: allow a handful, one
    # per two million elements)
    assert abs(np.count_nonzero(X) - np.count_nonzero(b0.X)) <= max(4, X.size // 2000000)
    assert np.count_nonzero(X) < X.size
    # continue after the layout round trip
    b.solve()
    b0.solve()
    assert rel_l2(b.X, b0.X) < 1e-5


@pytest.mark.parametrize('K,fused', [(66, False), (74, True), (65, False)])
def test_filter_counts_just_above_64(backend, K, fused):
    """64 < K <= 72 (and the odd counts padded into that range): the ADMM tail kernels of such a
    handle pad the rows of its Xf buffer, so FISTA composes the staged calls there; from 74 on the
    slab kernels serve the fused iteration.  Both against the float64 oracle."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.pgm import cbpdn as pc
    H = W = 128
    D, S = problem(H, W, K, 1, seed=K)
    b = pc.ConvBPDN(D, S, 0.05, pc.ConvBPDN.Options({'MaxMainIter': 2, 'L': 50.0, 'RelStopTol': 0.0}))
    assert b._fused_ok() == fused
    X = b.solve()
    ref = orc.pgm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, 1, 1), 0.05, dtype=np.float64,
                        maxiter=2, L=50.0, rel_tol=0.0)
    assert rel_l2(X, ref['X']) < 1e-5
    for f in ('ObjFun', 'Rsdl'):
        assert rel_l2(getattr(b.getitstat(), f), ref[f]) < 1e-5, f


def test_fused_pgm_options_and_pickle(backend):
    """Linear momentum, NonNegCoef + L1Weight array (GENERAL prox), FastSolve, pickling."""
    if backend == 'hostsim':
        pytest.skip("kept for the GPU run: the CPU suite covers these kernels in other cases")
    from sporco_amd.pgm.momentum import MomentumLinear
    H, W, K, N = 256, 256, 4, (1 if backend == 'hostsim' else 2)
    D, S = problem(H, W, K, N, seed=77)
    rng = np.random.RandomState(5)
    wl1 = (0.5 + rng.rand(H, W, 1, 1, K)).astype(np.float32)
    optd = {'MaxMainIter': 3, 'RelStopTol': 0.0, 'L': 60.0, 'NonNegCoef': True, 'L1Weight': wl1,
            'Momentum': MomentumLinear()}
    b = make(D, S, optd)
    X = b.solve()
    assert b._fused_ok() and np.all(X >= 0)
    if backend == 'hostsim':
        from oracle import cbpdn_oracle as orc
        # weighted non-negative prox of the last step, restated: X = max(V - (lmbda/L) w, 0)
        assert np.count_nonzero(X) < X.size
    else:
        b0 = make(D, S, optd, generic=True)
        X0 = b0.solve()
        assert rel_l2(X, X0) < 1e-5
        for f in ('ObjFun', 'RegL1', 'Rsdl'):
            assert rel_l2(getattr(b.getitstat(), f), getattr(b0.getitstat(), f)) < 1e-5, f
    b2 = pickle.loads(pickle.dumps(b))
    b.solve()
    b2.solve()
    assert np.array_equal(b.X, b2.X)
    if backend == 'hostsim':
        return       # (keeps the CPU suite short; the rest runs on the GPU)
    # FastSolve: no statistics are read back
    optf = dict(optd, FastSolve=True)
    bf, bf0 = make(D, S, optf), make(D, S, optf, generic=True)
    assert rel_l2(bf.solve(), bf0.solve()) < 1e-5


def test_fused_backtracking_against_the_reference(backend):
    """BacktrackStandard inside the fused iteration (a held pgm_iter per trial, F and the terms
    of Q_L out of the momentum kernel, pgm_commit on acceptance) against the reference's own
    float32 and float64 runs (tests/golden/pgm_bt256_*.npz: 7 trials in the first iteration,
    one in each later one; sporco/pgm/backtrack.py:50-117)."""
    from conftest import load_golden
    from sporco_amd.pgm import cbpdn as pc
    from sporco_amd.pgm.backtrack import BacktrackStandard, BacktrackRobust
    g32, g64 = load_golden('pgm_bt256_f32'), load_golden('pgm_bt256_f64')
    iters = 4 if backend == 'hostsim' else 14
    optd = {'MaxMainIter': iters, 'RelStopTol': 0.0, 'L': 1.0,
            'Backtrack': BacktrackStandard(gamma_u=1.5)}
    b = pc.ConvBPDN(g32['D'], g32['S'], float(g32['lmbda']), pc.ConvBPDN.Options(optd))
    assert b._fused_ok()
    X = b.solve()
    its = b.getitstat()
    for g, tol in ((g32, 2e-5), (g64, 1e-4)):
        assert np.array_equal(np.asarray(its.IterBTrack, float), g['it_IterBTrack'][:iters])
        for f in ('ObjFun', 'DFid', 'RegL1', 'Rsdl', 'L', 'F_Btrack', 'Q_Btrack'):
            assert rel_l2(np.asarray(getattr(its, f), float), g['it_' + f][:iters]) < tol, f
        if iters == 14:
            assert rel_l2(X[::16, ::16], g['X_sub']) < tol
            assert abs(np.linalg.norm(X.astype(np.float64)) - float(g['X_l2'])) < tol * float(g['X_l2'])
            assert abs(float(b.L) - float(g['L_final'])) < 1e-6 * float(g['L_final'])
    # a search the fused call does not restate (a subclass of the rules) composes the staged calls
    class MyRule(BacktrackRobust):
        pass
    optr = dict(optd, Backtrack=MyRule(), MaxMainIter=1)
    br = pc.ConvBPDN(g32['D'], g32['S'], float(g32['lmbda']), pc.ConvBPDN.Options(optr))
    assert not br._fused_ok()


def test_fused_robust_backtracking_against_the_reference(backend):
    """BacktrackRobust (sporco/pgm/backtrack.py:120-208) on the fused kernels: Yf = (Tk Xf + t Z) / T
    formed on the device in the tile-major layout, a held trial without momentum output per L,
    Z += t L (Xf - Yf) after the commit, the residual against the Yf the iteration started from --
    against the reference's own float32 and float64 runs (tests/golden/pgm_btrobust256_*.npz) and
    against the staged composition of the same library."""
    from conftest import load_golden
    from sporco_amd.pgm import cbpdn as pc
    from sporco_amd.pgm.backtrack import BacktrackRobust
    g32, g64 = load_golden('pgm_btrobust256_f32'), load_golden('pgm_btrobust256_f64')
    iters = 4 if backend == 'hostsim' else 14
    optd = {'MaxMainIter': iters, 'RelStopTol': 0.0, 'L': 1.0, 'Backtrack': BacktrackRobust()}
    b = pc.ConvBPDN(g32['D'], g32['S'], float(g32['lmbda']), pc.ConvBPDN.Options(optd))
    assert b._fused_ok()
    b.dev.profile(True)
    X = b.solve()
    prof = b.dev.profile_read()
    assert prof['pgm_fft_momentum'][1] >= iters          # (the trials ran on the fused kernels)
    its = b.getitstat()
    for g, tol in ((g32, 5e-5), (g64, 1e-4)):
        assert np.array_equal(np.asarray(its.IterBTrack, float), g['it_IterBTrack'][:iters])
        for f in ('ObjFun', 'DFid', 'RegL1', 'Rsdl', 'L', 'F_Btrack', 'Q_Btrack'):
            assert rel_l2(np.asarray(getattr(its, f), float), g['it_' + f][:iters]) < tol, f
        if iters == 14:
            assert rel_l2(X[::16, ::16], g['X_sub']) < tol
            assert abs(float(b.L) - float(g['L_final'])) < 1e-6 * float(g['L_final'])
    # continuing the solve (restart from the state the first call left) and reading the iterates
    b.opt['MaxMainIter'] = 2
    b.solve()
    assert np.all(np.isfinite(np.asarray(b.getitstat().ObjFun, float)))
    assert rel_l2(np.asarray(b.Xf), np.fft.rfft2(np.asarray(b.X, np.float64), axes=(0, 1))) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize('K', [4, 128])
def test_backtracking_fused_and_composed_agree(gpu_backend, K):
    from sporco_amd.pgm.backtrack import BacktrackStandard
    H, W, N = 256, 256, 2
    D, S = problem(H, W, K, N, seed=8)
    optd = {'MaxMainIter': 6, 'RelStopTol': 0.0, 'L': 1.0, 'Backtrack': BacktrackStandard()}
    b = make(D, S, optd)
    assert b._fused_ok()
    b0 = make(D, S, optd, generic=True)
    assert not b0._fused_ok()
    assert rel_l2(b.solve(), b0.solve()) < 1e-5
    for f in ('L', 'IterBTrack', 'F_Btrack', 'Q_Btrack', 'ObjFun', 'Rsdl'):
        assert rel_l2(np.asarray(getattr(b.getitstat(), f), float),
                      np.asarray(getattr(b0.getitstat(), f), float)) < 1e-5, f


def test_fused_monotone_fista_against_the_reference(backend):
    """Monotone FISTA (sporco/pgm/pgm.py:804-811, :826-829) on the fused kernels: a held trial, the
    objective from its sums, the commit, and the reference's fall-back (composed from the staged
    calls) in the iterations where the objective went up -- three of the 14 of the reference's
    own float32 / float64 runs at L = 8 (tests/golden/pgm_monotone256_*.npz)."""
    from conftest import load_golden
    from sporco_amd.pgm import cbpdn as pc
    g32, g64 = load_golden('pgm_monotone256_f32'), load_golden('pgm_monotone256_f64')
    o = g64['it_ObjFun']
    assert int(np.sum(o[1:] == o[:-1])) == 3 and o[3] == o[2]        # (the third step is one of them)
    iters = 5 if backend == 'hostsim' else 14
    optd = {'MaxMainIter': iters, 'RelStopTol': 0.0, 'L': 8.0, 'Monotone': True}
    b = pc.ConvBPDN(g32['D'], g32['S'], float(g32['lmbda']), pc.ConvBPDN.Options(optd))
    assert b._fused_ok()
    b.dev.profile(True)
    X = b.solve()
    prof = b.dev.profile_read()
    assert prof['pgm_fft_momentum'][1] == iters - 1      # (all but the first iteration)
    its = b.getitstat()
    for g, tol in ((g32, 5e-5), (g64, 1e-4)):
        for f in ('ObjFun', 'DFid', 'RegL1', 'Rsdl', 'L'):
            assert rel_l2(np.asarray(getattr(its, f), float), g['it_' + f][:iters]) < tol, f
        if iters == 14:
            assert rel_l2(X[::16, ::16], g['X_sub']) < tol
            assert abs(np.linalg.norm(np.asarray(b.Xf).astype(np.complex128)) - float(g['Xf_l2'])) < tol * float(g['Xf_l2'])
            assert abs(np.linalg.norm(np.asarray(b.Yf).astype(np.complex128)) - float(g['Yf_l2'])) < tol * float(g['Yf_l2'])


@pytest.mark.parametrize('policy', ['cauchy', 'bb'])
def test_fused_step_size_policies_against_the_reference(backend, policy):
    """StepSizePolicyCauchy / StepSizePolicyBB (sporco/pgm/stepsize.py:67-145) beside the fused
    iteration: the inner products of the gradient from signal-sized residual spectra
    (sporco_amd_csc_pgm_resid / _pgm_resid_stats) -- against the reference's own float32 and
    float64 runs (tests/golden/pgm_step*256_*.npz; L is the policy's from the third iteration),
    and against the staged composition of the same library."""
    from conftest import load_golden
    from sporco_amd.pgm import cbpdn as pc
    from sporco_amd.pgm.stepsize import StepSizePolicyCauchy, StepSizePolicyBB
    cls = {'cauchy': StepSizePolicyCauchy, 'bb': StepSizePolicyBB}[policy]
    g32, g64 = load_golden('pgm_step%s256_f32' % policy), load_golden('pgm_step%s256_f64' % policy)
    assert len(set(np.round(g64['it_L'], 6))) > 5            # (the policy does move L)
    iters = 5 if backend == 'hostsim' else 14
    optd = {'MaxMainIter': iters, 'RelStopTol': 0.0, 'L': 50.0, 'StepSizePolicy': cls()}
    b = pc.ConvBPDN(g32['D'], g32['S'], float(g32['lmbda']), pc.ConvBPDN.Options(optd))
    assert b._fused_ok()
    b.dev.profile(True)
    X = b.solve()
    prof = b.dev.profile_read()
    assert prof['pgm_fft_momentum'][1] == iters and prof.get('fft_c2c_cols_fwd', (0, 0))[1] == 0
    its = b.getitstat()
    for g, tol in ((g32, 2e-4), (g64, 2e-4)):
        for f in ('L', 'ObjFun', 'DFid', 'RegL1', 'Rsdl'):
            assert rel_l2(np.asarray(getattr(its, f), float), g['it_' + f][:iters]) < tol, f
        if iters == 14:
            assert rel_l2(X[::16, ::16], g['X_sub']) < 10 * tol

    class Staged(cls):        # (a subclass: the fused iteration does not restate a rule it does not know)
        pass
    optd['StepSizePolicy'] = Staged()
    b0 = pc.ConvBPDN(g32['D'], g32['S'], float(g32['lmbda']), pc.ConvBPDN.Options(optd))
    assert not b0._fused_ok()
    b0.solve()
    assert rel_l2(np.asarray(its.L, float), np.asarray(b0.getitstat().L, float)) < 1e-4


def test_residual_slots_against_numpy(backend):
    """sporco_amd_csc_pgm_resid / _pgm_resid_stats: the inner products the step-size policies take of
    the gradient g = conj(Df) e, e = sum_m Df v - Sf (sporco/pgm/stepsize.py:84-87, :137-141;
    grad_f, hessian_f: pgm/cbpdn.py:263-279, :302-312), formed from signal-sized spectra -- against
    the same sums formed from the X-sized arrays by NumPy; from the reference layout and, after a
    fused iteration, from the tile-major one."""
    from sporco_amd import _lib
    from sporco_amd.pgm import cbpdn as pc
    H = 256 if backend == 'gpu' else 128
    K, N = 8, 2
    D, S = problem(H, H, K, N, seed=5)
    b = pc.ConvBPDN(D, S, 0.05, pc.ConvBPDN.Options({'MaxMainIter': 3, 'RelStopTol': 0.0, 'L': 50.0}))
    b.solve()                                    # (the iterates are tile-major now)
    dev = b.dev
    Df = np.fft.rfft2(D.astype(np.float64), (H, H), axes=(0, 1))[:, :, None, :]      # (H, Wf, 1, K)
    Sf = np.fft.rfft2(S.astype(np.float64), axes=(0, 1))[..., None]                # (H, Wf, N, 1)

    def e_of(vf):
        return np.sum(Df * vf, axis=-1, keepdims=True) - Sf

    for tiled in (True, False):
        dev.pgm_resid(_lib.VAR_YF, 0)
        dev.pgm_resid(_lib.VAR_XF, 1)
        dev.pgm_resid(_lib.VAR_XFPRV, 2)
        s_y = dev.pgm_resid_stats(0)
        s_d = dev.pgm_resid_stats(0, 1, 1, 2)
        Yf = np.asarray(b.Yf, np.complex128).reshape(H, H // 2 + 1, N, K)          # (leaves the tiled layout)
        Xf = np.asarray(b.Xf, np.complex128).reshape(Yf.shape)
        Xp = dev.download(_lib.VAR_XFPRV).astype(np.complex128).reshape(Yf.shape)
        g = np.conj(Df) * e_of(Yf)
        Hg = np.conj(Df) * np.sum(Df * g, axis=-1, keepdims=True)
        assert abs(s_y[0] - np.sum(np.abs(g) ** 2)) < 1e-4 * np.sum(np.abs(g) ** 2)
        assert abs(s_y[1] - np.sum(np.real(np.conj(g) * Hg))) < 1e-4 * np.sum(np.real(np.conj(g) * Hg))
        dg = np.conj(Df) * (e_of(Yf) - e_of(Xf))
        dx = Xf - Xp
        assert abs(s_d[0] - np.sum(np.abs(dg) ** 2)) < 1e-4 * np.sum(np.abs(dg) ** 2)
        ref = np.sum(np.real(np.conj(dx) * dg))
        assert abs(s_d[2] - ref) < 1e-4 * max(abs(ref), 1e-3 * np.sqrt(np.sum(np.abs(dx) ** 2) * np.sum(np.abs(dg) ** 2)))
