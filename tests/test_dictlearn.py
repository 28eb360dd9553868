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
    with pytest.raises(NotImplementedError):
        cbpdndl.ConvBPDNDictLearn.Options(dmethod='cns')
