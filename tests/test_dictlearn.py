"""Parity of the dictionary update (pgm.ccmod.ConvCnstrMOD) and of
dictlrn.cbpdndl.ConvBPDNDictLearn with the reference.

The reference's own tests for these classes are smoke tests only
(tests/dictlrn/test_cbpdndl.py); numerical parity is pinned
here by trace comparison against fixtures produced by the unmodified reference
(oracle/make_golden.py): per outer iteration ObjFun/DFid/RegL1/X-residuals/rho
and the final dictionary and coefficient maps.
"""

import numpy as np
import pytest

from conftest import load_golden, rel_l2


def trace_errors(its, g, skip=('Iter', 'Time', 'Cnstr')):
    errs = {}
    for f in its._fields:
        col = getattr(its, f)
        if f in skip or 'it_' + f not in g or col[0] is None:
            continue
        errs[f] = rel_l2(np.asarray(col, dtype=float), g['it_' + f])
    return errs


def test_ccmod_pgm_trace(backend):
    from sporco_amd.pgm import ccmod
    g = load_golden('pgm_ccmod_f64')
    opt = ccmod.ConvCnstrMOD.Options({'MaxMainIter': 25, 'L': 800.0})
    c = ccmod.ConvCnstrMOD(g['Z'], g['S'], tuple(int(v) for v in g['dsz']), opt)
    c.solve()
    assert rel_l2(c.getdict(), g['D']) < 1e-9
    assert rel_l2(c.getdict(crop=False), g['Xfull']) < 1e-9
    errs = trace_errors(c.getitstat(), g)
    assert errs and max(errs.values()) < 1e-9, errs
    assert max(c.getitstat().Cnstr) < 1e-12
    # unit-norm, support-constrained filters
    D = c.getdict(crop=False)
    assert np.allclose(np.sum(D ** 2, axis=(0, 1)).ravel(), 1.0)
    assert np.all(D[5:] == 0) and np.all(D[:, 5:] == 0)


@pytest.mark.parametrize('name,dt,tol', [('cbpdndl_f64', np.float64, 1e-9),
                                         ('cbpdndl_f32', np.float32, 1e-4)])
def test_dictlearn_trace(backend, name, dt, tol):
    from sporco_amd.dictlrn import cbpdndl
    g = load_golden(name)
    opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 12, 'AccurateDFid': True},
                                            xmethod='admm', dmethod='pgm')
    b = cbpdndl.ConvBPDNDictLearn(g['D0'].astype(dt), g['S'].astype(dt), float(g['lmbda']),
                                  opt, xmethod='admm', dmethod='pgm')
    D1 = b.solve()
    assert D1.dtype == dt
    assert rel_l2(D1.squeeze(), g['D1'].squeeze()) < tol
    assert rel_l2(b.getcoef(), g['X']) < tol
    errs = trace_errors(b.getitstat(), g)
    assert set(errs) >= {'ObjFun', 'DFid', 'RegL1', 'XPrRsdl', 'XDlRsdl', 'XRho', 'D_L',
                         'D_Rsdl'}
    assert max(errs.values()) < tol, errs
    # the X-step sees the updated dictionary
    assert rel_l2(b.xstep.D.squeeze(), D1.squeeze()) == 0.0
    S = g['S'].astype(dt)
    rec = b.reconstruct().squeeze()
    assert rec.shape == S.shape


def test_dictlearn_returnx_feeds_x_to_the_dstep(backend):
    """CBPDN option ReturnX: xstep.getcoef() -- what the D-step receives, dictlrn.py:379-382 --
    is the X variable, not Y; the learned dictionary then differs from the default run."""
    from sporco_amd.dictlrn import cbpdndl
    g = load_golden('cbpdndl_returnx_f64')
    opt = cbpdndl.ConvBPDNDictLearn.Options(
        {'MaxMainIter': 8, 'AccurateDFid': True, 'CBPDN': {'ReturnX': True}},
        xmethod='admm', dmethod='pgm')
    b = cbpdndl.ConvBPDNDictLearn(g['D0'], g['S'], float(g['lmbda']), opt, xmethod='admm',
                                  dmethod='pgm')
    D1 = b.solve()
    assert rel_l2(D1.squeeze(), g['D1'].squeeze()) < 1e-9
    assert rel_l2(b.getcoef(), g['X']) < 1e-9
    assert b.reconstruct().shape == g['recon'].shape and rel_l2(b.reconstruct(), g['recon']) < 1e-9
    errs = trace_errors(b.getitstat(), g)
    assert max(errs.values()) < 1e-9, errs
    assert rel_l2(D1.squeeze(), load_golden('cbpdndl_f64')['D1'].squeeze()) > 1e-3


def test_dictlearn_variants_run(backend):
    """Option plumbing of the reference's tests/dictlrn/test_cbpdndl.py: PGM
    X-step, backtracking D-step, DictSize, inaccurate DFid, callback stop."""
    from sporco_amd.dictlrn import cbpdndl
    from sporco_amd.pgm.backtrack import BacktrackStandard
    rng = np.random.RandomState(12345)
    D0 = rng.randn(5, 5, 4)
    S = rng.randn(16, 16, 3)
    opt = cbpdndl.ConvBPDNDictLearn.Options(
        {'MaxMainIter': 4, 'CCMOD': {'Backtrack': BacktrackStandard(), 'L': 10.0}},
        xmethod='pgm', dmethod='pgm')
    b = cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='pgm', dmethod='pgm')
    D1 = b.solve()
    its = b.getitstat()
    assert D1.shape == (5, 5, 1, 1, 4) and len(its.ObjFun) == 4
    assert its.D_ItBt[0] is not None and its.X_L[0] == 500.0
    assert np.allclose(np.sum(D1 ** 2, axis=(0, 1)).ravel(), 1.0)

    stops = []
    opt = cbpdndl.ConvBPDNDictLearn.Options(
        {'MaxMainIter': 10, 'DictSize': (4, 4, 4),
         'Callback': lambda obj: stops.append(obj.j) or obj.j >= 2})
    b = cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt)
    D1 = b.solve()
    assert stops == [0, 1, 2] and D1.shape == (4, 4, 1, 1, 4)
    assert b.getitstat().ObjFun[-1] > 0          # taken from the X-step statistics

    with pytest.raises(ValueError):
        cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='pgm')
    with pytest.raises(ValueError):
        cbpdndl.ConvBPDN(D0, S, 0.1, method='nonsense')
    with pytest.raises(ValueError):
        cbpdndl.ConvBPDNDictLearn.Options(dmethod='nonsense')
    for m in ('ism', 'cg', 'cns'):                # all three ADMM D-steps are offered
        assert cbpdndl.ConvBPDNDictLearn.Options(dmethod=m)['CCMOD', 'AutoRho', 'Period'] == 10


# ---------------------------------------------------------------------------
# the tile-major D-step (csc_pgm.hip: register-resident setcoef + grouped gradient),
# engaged for float32, H and W in {256, 512}, even K <= 256
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('K', [4, 66, pytest.param(128, marks=pytest.mark.gpu),
                               pytest.param(70, marks=pytest.mark.gpu)])
def test_tiled_dstep_one_fista_step(backend, K):
    """One ConvCnstrMOD iteration against its NumPy restatement (pgm/ccmod.py:295-323).
    K > 64: the column transform of setcoef and the gradient run per 64-filter slab (two passes
    over the coefficient spectrum)."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.pgm import ccmod
    H, W, N = 256, 256, 2
    rng = np.random.RandomState(21)
    Z = (rng.randn(H, W, 1, N, K) * (rng.rand(H, W, 1, N, K) < 0.05)).astype(np.float32)
    S = rng.randn(H, W, N).astype(np.float32)
    D0 = orc.pcn(rng.randn(H, W, 1, 1, K), (5, 5, K), (H, W)).astype(np.float32)
    opt = ccmod.ConvCnstrMOD.Options({'MaxMainIter': 1, 'L': 300.0, 'X0': D0})
    c = ccmod.ConvCnstrMOD(Z, S, (5, 5, K), opt)
    assert c.dev.uses_fused_rows()
    c.solve()
    Zf = np.fft.rfftn(Z.astype(np.float64), axes=(0, 1))
    Sf = np.fft.rfftn(S.astype(np.float64).reshape(H, W, 1, N, 1), axes=(0, 1))
    Df = np.fft.rfftn(D0.astype(np.float64), axes=(0, 1))
    R = np.sum(Zf * Df, axis=4, keepdims=True) - Sf
    G = np.sum(np.conj(Zf) * R, axis=3, keepdims=True)
    V = np.fft.irfftn(Df - G / 300.0, (H, W), axes=(0, 1))
    D1 = orc.pcn(V, (5, 5, K), (H, W))
    assert rel_l2(c.getdict(crop=False), D1) < 1e-5
    its = c.getitstat()
    R1 = np.sum(Zf * np.fft.rfftn(D1, axes=(0, 1)), axis=4, keepdims=True) - Sf
    dfid = 0.5 * np.sum(np.fft.irfftn(R1, (H, W), axes=(0, 1)) ** 2)
    assert abs(its.DFid[-1] - dfid) < 1e-5 * dfid
    # the coefficient spectrum comes back in the reference layout
    assert rel_l2(c.Zf, Zf) < 1e-5


@pytest.mark.parametrize('dmethod', ['pgm', 'cns'])
def test_dictlearn_at_128_fused_vs_generic(backend, dmethod):
    """128 x 128 images: the register-resident kernels of both steps (32 x 4 splits) against the
    generic kernel chain of the same library."""
    import os
    from sporco_amd.dictlrn import cbpdndl
    rng = np.random.RandomState(1)
    D0 = rng.randn(4, 4, 4).astype(np.float32)
    S = rng.randn(128, 128, 2).astype(np.float32)
    out = []
    for generic in (False, True):
        if generic:
            os.environ['SPORCO_AMD_UNFUSED'] = '1'
        try:
            opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 2 if backend == 'hostsim' else 5},
                                                    xmethod='admm', dmethod=dmethod)
            d = cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='admm', dmethod=dmethod)
        finally:
            os.environ.pop('SPORCO_AMD_UNFUSED', None)
        assert bool(d.xstep._dev.uses_fused_rows()) == (not generic)
        out.append((d.solve().copy(), d.getitstat()))
    assert rel_l2(out[0][0], out[1][0]) < 1e-5
    for f in ('ObjFun', 'DFid', 'RegL1'):
        assert rel_l2(getattr(out[0][1], f), getattr(out[1][1], f)) < 1e-5, f


@pytest.mark.gpu
@pytest.mark.parametrize('K', [8, 7])
def test_dictlearn_fused_vs_generic(gpu_backend, K):
    """ConvBPDNDictLearn through the fused kernels == the generic kernel chain (K = 7:
    the handle pads the filter axis to 8 for the fused kernels; getdict drops the pad)."""
    import os
    from sporco_amd.dictlrn import cbpdndl
    H = W = 256
    N = 4
    rng = np.random.RandomState(3)
    D0 = rng.randn(6, 6, K).astype(np.float32)
    S = rng.randn(H, W, N).astype(np.float32)

    def run(generic):
        if generic:
            os.environ['SPORCO_AMD_OLD_ROWS'] = '1'
            os.environ['SPORCO_AMD_NO_PAD'] = '1'
        try:
            opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 6, 'AccurateDFid': True},
                                                    xmethod='admm', dmethod='pgm')
            d = cbpdndl.ConvBPDNDictLearn(D0, S, 0.1, opt, xmethod='admm', dmethod='pgm')
        finally:
            os.environ.pop('SPORCO_AMD_OLD_ROWS', None)
            os.environ.pop('SPORCO_AMD_NO_PAD', None)
        return d, d.solve()

    d, D1 = run(False)
    d0, D10 = run(True)
    assert d.xstep._dev.uses_fused_rows() and not d0.xstep._dev.uses_fused_rows()
    assert D1.shape == D10.shape and D1.shape[-1] == K
    assert rel_l2(D1, D10) < 1e-5
    assert rel_l2(d.getcoef(), d0.getcoef()) < 1e-5
    errs = {f: rel_l2(np.asarray(getattr(d.getitstat(), f), dtype=float),
                      np.asarray(getattr(d0.getitstat(), f), dtype=float))
            for f in ('ObjFun', 'DFid', 'RegL1', 'XPrRsdl', 'XDlRsdl', 'XRho', 'D_Rsdl')}
    assert max(errs.values()) < 1e-5, errs


# ---------------------------------------------------------------------------
# multi-channel dictionaries (Cd > 1): D-step and dictionary learning
# ---------------------------------------------------------------------------
def test_ccmod_pgm_multichannel_dictionary(backend):
    from sporco_amd.pgm import ccmod
    g = load_golden('pgm_ccmod_mcdict_f64')
    opt = ccmod.ConvCnstrMOD.Options({'MaxMainIter': 20, 'L': 800.0, 'ZeroMean': True})
    c = ccmod.ConvCnstrMOD(g['Z'], g['S'], tuple(int(v) for v in g['dsz']), opt)
    c.solve()
    D = c.getdict()
    assert D.shape == g['D'].shape and rel_l2(D, g['D']) < 1e-9
    assert rel_l2(c.getdict(crop=False), g['Xfull']) < 1e-9
    errs = trace_errors(c.getitstat(), g)
    assert errs and max(errs.values()) < 1e-9, errs
    # filters have unit norm over support and channels, zero mean per channel
    assert np.allclose(np.sum(D ** 2, axis=(0, 1, 2)).ravel(), 1.0)
    assert np.max(np.abs(np.mean(D, axis=(0, 1)))) < 1e-12


def test_ccmod_pgm_channelful_maps_with_multichannel_dictionary(backend):
    """Coefficient maps that carry the channels of a multi-channel dictionary -- the
    reference's broadcasting then solves one single-channel problem per channel with a shared
    step size (its tests/pgm/test_ccmod.py:175-191).  Fixture: oracle/make_golden.py gen_zchan."""
    from sporco_amd.pgm import ccmod
    g = load_golden('pgm_ccmod_zchan_f64')
    c = ccmod.ConvCnstrMOD(g['Z'], g['S'], tuple(int(v) for v in g['dsz']),
                           ccmod.ConvCnstrMOD.Options({'MaxMainIter': 20, 'L': 400.0}))
    c.solve()
    assert c.getdict().shape == g['D'].shape and rel_l2(c.getdict(), g['D']) < 1e-9
    assert rel_l2(c.X, g['X']) < 1e-9
    its = c.getitstat()
    for f in ('DFid', 'Rsdl', 'L'):
        assert rel_l2(np.asarray(getattr(its, f), float), g['it_' + f]) < 1e-9, f
    assert np.max(np.abs(np.asarray(its.Cnstr) - g['it_Cnstr'])) < 1e-11


@pytest.mark.parametrize('name,dt,tol', [
    ('cbpdndl_mcdict_f64', np.float64, 1e-9),
    pytest.param('cbpdndl_mcdict_f32', np.float32, 1e-3, marks=pytest.mark.gpu)])
def test_dictlearn_multichannel_dictionary(backend, name, dt, tol):
    from sporco_amd.dictlrn import cbpdndl
    g = load_golden(name)
    opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 10, 'AccurateDFid': True},
                                            xmethod='admm', dmethod='pgm')
    b = cbpdndl.ConvBPDNDictLearn(g['D0'].astype(dt), g['S'].astype(dt), float(g['lmbda']),
                                  opt, xmethod='admm', dmethod='pgm')
    D1 = b.solve()
    assert D1.squeeze().shape == g['D1'].squeeze().shape
    assert rel_l2(D1.squeeze(), g['D1'].squeeze()) < tol
    assert rel_l2(b.getcoef(), g['X']) < tol
    errs = trace_errors(b.getitstat(), g)
    assert max(errs.values()) < tol, errs


# ---------------------------------------------------------------------------
# dictionary recovery: a size-independent property of the PGM D-step
# ---------------------------------------------------------------------------
def _policy(name):
    from sporco_amd.pgm.backtrack import BacktrackStandard
    from sporco_amd.pgm.momentum import MomentumLinear, MomentumGenLinear
    from sporco_amd.pgm.stepsize import StepSizePolicyBB, StepSizePolicyCauchy
    return {'standard': {'Backtrack': BacktrackStandard()},
            'linear': {'Momentum': MomentumLinear()},
            'genlinear': {'Momentum': MomentumGenLinear()},
            'cauchy': {'StepSizePolicy': StepSizePolicyCauchy()},
            'bb': {'StepSizePolicy': StepSizePolicyBB()},
            'monotone': {'Monotone': True}}[name]


@pytest.mark.gpu
@pytest.mark.parametrize('N,Nd,L,policy', [
    (32, 5, 2.5, 'standard'), (32, 5, 0.5, 'standard'), (64, 8, 0.5, 'standard'),
    (32, 5, 2.5, 'linear'), (32, 5, 2.5, 'genlinear'), (64, 8, 0.5, 'bb'),
    (64, 8, 0.5, 'cauchy'), (64, 8, 50.0, 'monotone')])
def test_ccmod_recovers_the_generating_dictionary(N, Nd, L, policy):
    """A signal synthesised from a normalised zero-mean dictionary and sparse coefficient maps:
    3000 iterations of the update return that dictionary (relative residual < 1e-4, last
    iterate residual < 1e-5) under every backtracking / momentum / step-size policy.  Inputs and
    parameters of the reference's tests/pgm/test_ccmod.py:194-400 (legacy generator seeded with
    12345, drawn in that order: the fixed-L momentum runs converge on these draws, not on every
    one) -- runs that take the CPU simulator the better part of an hour each
    (profiles/r03_reference_test_files.md) and the GPU about a second."""
    from sporco_amd import cnvrep as cr
    from sporco_amd.pgm import ccmod
    rng = np.random.RandomState(12345)
    M = 4
    D0 = cr.normalise(cr.zeromean(rng.randn(Nd, Nd, M), (Nd, Nd, M), dimN=2), dimN=2)
    X = np.zeros((N, N, M))
    big = np.abs(rng.randn(N, N, M)) > 3
    X[big] = rng.randn(int(big.sum()))
    S = np.sum(np.fft.ifft2(np.fft.fft2(D0, (N, N), axes=(0, 1)) * np.fft.fft2(X, axes=(0, 1)),
                            axes=(0, 1)).real, axis=2)
    optd = dict({'Verbose': False, 'MaxMainIter': 3000, 'ZeroMean': True, 'RelStopTol': 0., 'L': L},
                **_policy(policy))
    c = ccmod.ConvCnstrMOD(X.reshape(N, N, 1, 1, M), S.reshape(N, N, 1), D0.shape,
                           ccmod.ConvCnstrMOD.Options(optd))
    c.solve()
    D1 = cr.bcrop(c.X, D0.shape).squeeze()
    assert np.linalg.norm(D0 - D1) / max(np.linalg.norm(D0), np.linalg.norm(D1)) < 1e-4
    assert np.asarray(c.getitstat().Rsdl)[-1] < 1e-5


@pytest.mark.parametrize('N', [3, 20])
def test_generic_dstep_gradient_few_and_many_images(backend, N):
    """One ConvCnstrMOD iteration on the generic chain against its NumPy restatement
    (pgm/ccmod.py:295-323) with 3 images (a wave per frequency walks them) and with 20 (the
    images spread over the waves of a workgroup): the two forms of the gradient kernel."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.pgm import ccmod
    H, W, K = 16, 24, 5
    rng = np.random.RandomState(N)
    Z = rng.randn(H, W, 1, N, K) * (rng.rand(H, W, 1, N, K) < 0.3)
    S = rng.randn(H, W, N)
    D0 = orc.pcn(rng.randn(H, W, 1, 1, K), (5, 5, K), (H, W))
    L = 40.0 * N
    c = ccmod.ConvCnstrMOD(Z, S, (5, 5, K), ccmod.ConvCnstrMOD.Options({'MaxMainIter': 1, 'L': L, 'X0': D0}))
    c.solve()
    Zf = np.fft.rfftn(Z, axes=(0, 1))
    Sf = np.fft.rfftn(S.reshape(H, W, 1, N, 1), axes=(0, 1))
    Df = np.fft.rfftn(D0, axes=(0, 1))
    R = np.sum(Zf * Df, axis=4, keepdims=True) - Sf
    G = np.sum(np.conj(Zf) * R, axis=3, keepdims=True)
    D1 = orc.pcn(np.fft.irfftn(Df - G / L, (H, W), axes=(0, 1)), (5, 5, K), (H, W))
    assert rel_l2(c.getdict(crop=False), D1) < 1e-11
    R1 = np.sum(Zf * np.fft.rfftn(D1, axes=(0, 1)), axis=4, keepdims=True) - Sf
    dfid = 0.5 * np.sum(np.fft.irfftn(R1, (H, W), axes=(0, 1)) ** 2)
    assert abs(c.getitstat().DFid[-1] - dfid) < 1e-11 * dfid


@pytest.mark.parametrize('name,xm', [('cbpdndl_dim1_admm_f64', 'admm'), ('cbpdndl_dim1_pgm_f64', 'pgm'),
                                     ('cbpdndl_dim1_mcdict_f64', 'admm')])
def test_dictlearn_dimN1_signals(backend, name, xm):
    """dimN = 1 (cbpdndl.py:385 with one-dimensional signals): both steps run the signals as images
    with a unit first axis; the arrays the caller sees have the reference's shapes and values."""
    from sporco_amd.dictlrn import cbpdndl
    g = load_golden(name)
    opt = cbpdndl.ConvBPDNDictLearn.Options(
        {'MaxMainIter': 10, 'AccurateDFid': True, 'CCMOD': {'ZeroMean': bool(g['zm'])}}, xmethod=xm, dmethod='pgm')
    b = cbpdndl.ConvBPDNDictLearn(g['D0'], g['S'], float(g['lmbda']), opt, xmethod=xm, dmethod='pgm',
                                  dimK=1, dimN=1)
    D1 = b.solve()
    assert D1.shape == g['D1'].shape and rel_l2(D1, g['D1']) < 1e-9
    assert b.getcoef().shape == g['X'].shape and rel_l2(b.getcoef(), g['X']) < 1e-9
    assert b.reconstruct().shape == g['recon'].shape and rel_l2(b.reconstruct(), g['recon']) < 1e-9
    assert rel_l2(b.reconstruct(D=b.getdict(crop=False), X=b.getcoef()), g['recon']) < 1e-9
    errs = trace_errors(b.getitstat(), g)
    assert {'ObjFun', 'DFid', 'RegL1', 'D_L', 'D_Rsdl'} <= set(errs) and max(errs.values()) < 1e-9, errs
    assert rel_l2(b.xstep.D[0], D1) == 0.0 and b.getdict(crop=False).shape == (g['S'].shape[0],) + D1.shape[1:]


def test_pgm_ccmod_dimN1(backend):
    """The PGM dictionary update on signals (pgm/ccmod.py:139 with dimN = 1), plain and masked."""
    from sporco_amd.pgm import ccmod
    g = load_golden('pgm_ccmod_dim1_f64')
    dsz = tuple(int(v) for v in g['dsz'])
    c = ccmod.ConvCnstrMOD(g['Z'], g['S'], dsz, ccmod.ConvCnstrMOD.Options(
        {'MaxMainIter': 25, 'L': 60.0, 'ZeroMean': True}), dimK=1, dimN=1)
    X = c.solve()
    assert X.shape == g['Xfull'].shape and rel_l2(X, g['Xfull']) < 1e-9
    assert c.getdict().shape == g['D'].shape and rel_l2(c.getdict(), g['D']) < 1e-9
    assert rel_l2(c.X, g['Xfull']) < 1e-9 and rel_l2(c.reconstruct(), g['recon']) < 1e-9
    assert rel_l2(c.reconstruct(D=c.X), g['recon']) < 1e-9
    errs = trace_errors(c.getitstat(), g)
    assert errs and max(errs.values()) < 1e-9, errs
    assert np.allclose(np.sum(c.X ** 2, axis=0).ravel(), 1.0) and np.all(c.X[dsz[0]:] == 0)
    assert np.abs(np.sum(c.X, axis=0)).max() < 1e-12
    assert rel_l2(c.Pcn(c.X), c.X) < 1e-12
    m = ccmod.ConvCnstrMODMask(g['Z'], g['S'], g['W'], dsz, ccmod.ConvCnstrMODMask.Options(
        {'MaxMainIter': 25, 'L': 60.0}), dimK=1, dimN=1)
    m.solve()
    assert rel_l2(m.getdict(), g['D_mask']) < 1e-9
    assert rel_l2(np.array(m.getitstat().DFid), g['DFid_mask']) < 1e-9
    with pytest.raises(NotImplementedError):
        ccmod.ConvCnstrMOD(None, np.zeros((32, 2, 3)), (5, 2, 4), ccmod.ConvCnstrMOD.Options({'ZeroMean': True}),
                           dimK=1, dimN=1)


def test_tiled_gradient_at_256_vs_numpy(backend):
    """One pgm.ccmod.ConvCnstrMOD iteration at 256 x 256, K = 64 (the dictionary-learning shape of
    BASELINE configs[4]; N = 2 keeps the CPU simulator to seconds): the tile-major gradient kernel of
    that shape -- a whole tile requested ahead, 16 rows per thread, csrc/csc_pgm.hip
    ccmod_grad_tiled_ahead_kernel<16> -- against the NumPy restatement of pgm/ccmod.py:295-323."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd import _lib
    from sporco_amd.pgm import ccmod
    H, W, K, N = 256, 256, 64, 2
    rng = np.random.RandomState(506)
    Z = (rng.randn(H, W, 1, N, K) * (rng.rand(H, W, 1, N, K) < 0.02)).astype(np.float32)
    S = rng.randn(H, W, N).astype(np.float32)
    D0 = orc.pcn(rng.randn(H, W, 1, 1, K), (8, 8, K), (H, W)).astype(np.float32)
    L = 14.0 * N * 50
    c = ccmod.ConvCnstrMOD(Z, S, (8, 8, K), ccmod.ConvCnstrMOD.Options({'MaxMainIter': 1, 'L': L, 'X0': D0}))
    assert c.dev.uses_fused_rows() and 1 <= c.dev.query(_lib.QUERY_CCMOD_GROUPS) <= N
    c.solve()
    Zf = np.fft.rfftn(Z.astype(np.float64), axes=(0, 1))
    Sf = np.fft.rfftn(S.astype(np.float64).reshape(H, W, 1, N, 1), axes=(0, 1))
    Df = np.fft.rfftn(D0.astype(np.float64), axes=(0, 1))
    R = np.sum(Zf * Df, axis=4, keepdims=True) - Sf
    G = np.sum(np.conj(Zf) * R, axis=3, keepdims=True)
    D1 = orc.pcn(np.fft.irfftn(Df - G / L, (H, W), axes=(0, 1)), (8, 8, K), (H, W))
    assert rel_l2(c.getdict(crop=False), D1) < 1e-5
    R1 = np.sum(Zf * np.fft.rfftn(D1, axes=(0, 1)), axis=4, keepdims=True) - Sf
    dfid = 0.5 * np.sum(np.fft.irfftn(R1, (H, W), axes=(0, 1)) ** 2)
    assert abs(c.getitstat().DFid[-1] - dfid) < 1e-5 * dfid


@pytest.mark.parametrize('meth', ['ism', 'cg', 'cns'])
def test_admm_dictionary_updates_dimN1(backend, meth):
    """dimN = 1 in the ADMM dictionary updates (admm/ccmod.py) alone and as the D-step of
    ConvBPDNDictLearn, against reference runs; CG is run to 1e-9."""
    from sporco_amd.admm import ccmod
    from sporco_amd.dictlrn import cbpdndl
    g = load_golden('ccmod_dim1_%s_f64' % meth)
    cls = {'ism': ccmod.ConvCnstrMOD_IterSM, 'cg': ccmod.ConvCnstrMOD_CG, 'cns': ccmod.ConvCnstrMOD_Consensus}[meth]
    tol = 1e-7 if meth == 'cg' else 1e-9
    optd = {'MaxMainIter': 15, 'ZeroMean': True, 'LinSolveCheck': True}
    if meth == 'cg':
        optd['CG'] = {'MaxIter': 500, 'StopTol': 1e-9}
    dsz = tuple(int(v) for v in g['dsz'])
    c = cls(g['Z'], g['S'], dsz, cls.Options(optd), dimK=1, dimN=1)
    Y = c.solve()
    assert c.k == int(g['k_final'])
    for a, name in ((Y, 'Y'), (c.getdict(), 'D'), (c.X, 'X'), (c.U, 'U')):
        assert a.shape == g[name].shape and rel_l2(a, g[name]) < tol, name
    errs = trace_errors(c.getitstat(), g)
    assert {'DFid', 'PrimalRsdl', 'DualRsdl', 'Rho'} <= set(errs) and max(errs[f] for f in ('DFid', 'PrimalRsdl', 'DualRsdl', 'Rho')) < tol, errs
    if meth != 'cns':
        assert c.reconstruct().shape == g['S'].shape[0:1] + (1,) + g['S'].shape[1:]
    # the setters take the reference's shapes
    c.Y, c.U = c.Y.copy(), c.U.copy()
    assert rel_l2(c.Y, g['Y']) < tol
    optl = {'MaxMainIter': 8, 'AccurateDFid': True, 'CCMOD': {'ZeroMean': True}}
    if meth == 'cg':
        optl['CCMOD']['CG'] = {'MaxIter': 500, 'StopTol': 1e-9}
    b = cbpdndl.ConvBPDNDictLearn(g['D0'], g['S'], 0.1, cbpdndl.ConvBPDNDictLearn.Options(
        optl, xmethod='admm', dmethod=meth), xmethod='admm', dmethod=meth, dimK=1, dimN=1)
    D1 = b.solve()
    assert D1.shape == g['dl_D1'].shape and rel_l2(D1, g['dl_D1']) < tol
    assert rel_l2(b.getcoef(), g['dl_X']) < tol and rel_l2(b.getitstat().ObjFun, g['dl_ObjFun']) < tol


@pytest.mark.parametrize('name', ['cbpdndl_dim3_video_f64', 'cbpdndl_dim3_pgm_f64'])
def test_dictlearn_dimN3_volumes(backend, name):
    """dimN = 3 dictionary learning against reference runs: the configuration of the reference's
    examples/scripts/cdl/cbpdndl_video.py:64-74 (consensus dictionary update, one volume, AutoRho
    in both steps) and the PGM dictionary update on two volumes -- both steps on one volume handle."""
    from sporco_amd.dictlrn import cbpdndl
    g = load_golden(name)
    lmbda = float(g['lmbda'])
    if 'video' in name:
        opt = cbpdndl.ConvBPDNDictLearn.Options(
            {'MaxMainIter': 10, 'CBPDN': {'rho': 50.0 * lmbda, 'AutoRho': {'Enabled': True}},
             'CCMOD': {'rho': 1e2, 'AutoRho': {'Enabled': True}}}, dmethod='cns')
        b = cbpdndl.ConvBPDNDictLearn(g['D0'], g['S'], lmbda, opt, dimK=0, dimN=3)
    else:
        opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 10, 'AccurateDFid': True}, xmethod='admm',
                                                dmethod='pgm')
        b = cbpdndl.ConvBPDNDictLearn(g['D0'], g['S'], lmbda, opt, xmethod='admm', dmethod='pgm', dimK=1, dimN=3)
    D1 = b.solve()
    assert D1.shape == g['D1'].shape and rel_l2(D1, g['D1']) < 1e-9
    assert b.getcoef().shape == g['X'].shape and rel_l2(b.getcoef(), g['X']) < 1e-9
    assert b.reconstruct().shape == g['recon'].shape and rel_l2(b.reconstruct(), g['recon']) < 1e-9
    assert rel_l2(b.reconstruct(D=b.getdict(crop=False), X=b.getcoef()), g['recon']) < 1e-9
    errs = trace_errors(b.getitstat(), g)
    assert {'ObjFun', 'DFid', 'RegL1', 'XPrRsdl', 'XRho'} <= set(errs) and max(errs.values()) < 1e-9, errs
    with pytest.raises(NotImplementedError):
        cbpdndl.ConvBPDNDictLearn(g['D0'], g['S'], lmbda, cbpdndl.ConvBPDNDictLearn.Options(
            {}, xmethod='admm', dmethod='ism'), xmethod='admm', dmethod='ism', dimK=0 if 'video' in name else 1, dimN=3)


def test_dictionary_updates_dimN3(backend):
    """The PGM and the consensus dictionary update alone on volumes (pgm/ccmod.py:139,
    admm/ccmod.py:653 with dimN = 3): a crop in three axes inside the device's projections."""
    from sporco_amd.admm import ccmod as accmod
    from sporco_amd.pgm import ccmod
    g = load_golden('ccmod_dim3_f64')
    dsz = tuple(int(v) for v in g['dsz'])
    c = ccmod.ConvCnstrMOD(g['Z'], g['S'], dsz, ccmod.ConvCnstrMOD.Options({'MaxMainIter': 25, 'L': 150.0}),
                           dimK=1, dimN=3)
    X = c.solve()
    assert X.shape == g['pgm_X'].shape and rel_l2(X, g['pgm_X']) < 1e-9
    assert c.getdict().shape == g['pgm_D'].shape and rel_l2(c.getdict(), g['pgm_D']) < 1e-9
    assert rel_l2(c.reconstruct(), g['pgm_recon']) < 1e-9 and rel_l2(c.reconstruct(D=c.X), g['pgm_recon']) < 1e-9
    its = c.getitstat()
    assert rel_l2(its.DFid, g['pgm_DFid']) < 1e-9 and rel_l2(its.Rsdl, g['pgm_Rsdl']) < 1e-9
    assert max(its.Cnstr) < 1e-12
    assert np.allclose(np.sum(c.X ** 2, axis=(0, 1, 2)).ravel(), 1.0)
    assert np.all(c.X[dsz[0]:] == 0) and np.all(c.X[:, dsz[1]:] == 0) and np.all(c.X[:, :, dsz[2]:] == 0)
    assert rel_l2(c.Pcn(c.X), c.X) < 1e-12
    n = accmod.ConvCnstrMOD_Consensus(g['Z'], g['S'], dsz, accmod.ConvCnstrMOD_Consensus.Options(
        {'MaxMainIter': 20, 'LinSolveCheck': True, 'rho': 3.0}), dimK=1, dimN=3)
    Y = n.solve()
    for a, key in ((Y, 'cns_Y'), (n.getdict(), 'cns_D'), (n.X, 'cns_X'), (n.U, 'cns_U')):
        assert a.shape == g[key].shape and rel_l2(a, g[key]) < 1e-9, key
    its = n.getitstat()
    for f in ('DFid', 'PrimalRsdl', 'DualRsdl'):
        assert rel_l2(getattr(its, f), g['cns_' + f]) < 1e-9, f
    n.Y, n.U = n.Y.copy(), n.U.copy()          # (the setters take the reference's shapes)
    assert rel_l2(n.Y, g['cns_Y']) < 1e-9
    for cls, kw in ((ccmod.ConvCnstrMOD, {'ZeroMean': True}), (accmod.ConvCnstrMOD_Consensus, {'ZeroMean': True})):
        with pytest.raises(NotImplementedError):
            cls(g['Z'], g['S'], dsz, cls.Options(kw), dimK=1, dimN=3)
    with pytest.raises(NotImplementedError):
        accmod.ConvCnstrMOD_IterSM(g['Z'], g['S'], dsz, dimK=1, dimN=3)
