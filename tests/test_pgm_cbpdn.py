"""Parity of sporco_amd.pgm.cbpdn.ConvBPDN (FISTA) with the reference.

Golden fixtures come from the unmodified reference (oracle/make_golden.py):
every momentum rule, step-size policy and backtracking strategy the reference's
own tests exercise (tests/pgm/test_cbpdn.py:203-333), with all iterations run
(RelStopTol = 0).  Tolerance 1e-9 (float64) / 1e-4 (float32) relative l2.
"""

import pickle

import numpy as np
import pytest

from conftest import load_golden, rel_l2


def policies():
    from sporco_amd.pgm.backtrack import BacktrackStandard, BacktrackRobust
    from sporco_amd.pgm.momentum import MomentumLinear, MomentumGenLinear
    from sporco_amd.pgm.stepsize import StepSizePolicyBB, StepSizePolicyCauchy
    return {
        'pgm_default_f64': {'MaxMainIter': 40, 'L': 500.0},
        'pgm_default_f32': {'MaxMainIter': 40, 'L': 500.0, 'DataType': np.float32},
        'pgm_nonneg_nobndry_f64': {'MaxMainIter': 30, 'L': 500.0, 'NonNegCoef': True,
                                   'NoBndryCross': True},
        'pgm_btstd_f64': {'MaxMainIter': 30, 'L': 1.0, 'Backtrack': BacktrackStandard()},
        'pgm_btrobust_f64': {'MaxMainIter': 30, 'L': 1.0, 'Backtrack': BacktrackRobust()},
        'pgm_momlinear_f64': {'MaxMainIter': 30, 'L': 500.0, 'Momentum': MomentumLinear()},
        'pgm_momgenlinear_f64': {'MaxMainIter': 30, 'L': 500.0,
                                 'Momentum': MomentumGenLinear()},
        'pgm_stepbb_f64': {'MaxMainIter': 30, 'L': 500.0,
                           'StepSizePolicy': StepSizePolicyBB()},
        'pgm_stepcauchy_f64': {'MaxMainIter': 30, 'L': 500.0,
                               'StepSizePolicy': StepSizePolicyCauchy()},
        'pgm_monotone_f64': {'MaxMainIter': 30, 'L': 500.0, 'Monotone': True},
        'pgm_multichan_f64': {'MaxMainIter': 30, 'L': 500.0},
        # multi-channel dictionary: gradient summed over the channels (pgm/cbpdn.py:277-279)
        'pgm_mcdict_f64': {'MaxMainIter': 30, 'L': 500.0},
    }


NAMES = ['pgm_default_f64', 'pgm_default_f32', 'pgm_nonneg_nobndry_f64', 'pgm_btstd_f64',
         'pgm_btrobust_f64', 'pgm_momlinear_f64', 'pgm_momgenlinear_f64', 'pgm_stepbb_f64',
         'pgm_stepcauchy_f64', 'pgm_monotone_f64', 'pgm_multichan_f64', 'pgm_mcdict_f64']


@pytest.mark.parametrize('name', NAMES)
def test_golden_traces(backend, name):
    from sporco_amd.pgm import cbpdn
    g = load_golden(name)
    optd = dict(policies()[name])
    optd['RelStopTol'] = 0.0
    tol = 1e-4 if optd.get('DataType') is np.float32 else 1e-9
    b = cbpdn.ConvBPDN(g['D'], g['S'], float(g['lmbda']), cbpdn.ConvBPDN.Options(optd))
    X = b.solve()
    assert b.k == int(g['k_final'])
    assert X.shape == g['X'].shape
    assert rel_l2(X, g['X']) < tol
    assert rel_l2(b.Xf, g['Xf']) < tol
    assert abs(float(b.L) - float(g['L_final'])) < 1e-6 * float(g['L_final'])
    its = b.getitstat()
    for f in its._fields:
        col = getattr(its, f)
        if f in ('Iter', 'Time') or 'it_' + f not in g or col[0] is None:
            continue
        assert rel_l2(np.asarray(col, dtype=float), g['it_' + f]) < tol, f
    assert rel_l2(b.reconstruct(), g['recon']) < tol


def test_restart_and_pickle(backend):
    from sporco_amd.pgm import cbpdn
    g = load_golden('pgm_default_f64')
    opt = cbpdn.ConvBPDN.Options({'MaxMainIter': 15, 'L': 500.0, 'RelStopTol': 0.0})
    b = cbpdn.ConvBPDN(g['D'], g['S'], float(g['lmbda']), opt)
    b.solve()
    c = pickle.loads(pickle.dumps(b))
    b.solve()
    c.solve()
    assert b.k == 30 and c.k == 30
    assert np.linalg.norm(b.X - c.X) == 0.0
    assert rel_l2(b.getitstat().ObjFun, g['it_ObjFun'][:30]) < 1e-9


def test_options_type_check(backend):
    from sporco_amd.pgm import cbpdn
    from sporco_amd.admm import cbpdn as admm_cbpdn
    g = load_golden('pgm_default_f64')
    with pytest.raises(TypeError):
        cbpdn.ConvBPDN(g['D'], g['S'], 0.1, admm_cbpdn.ConvBPDN.Options())
    with pytest.raises(TypeError):
        admm_cbpdn.ConvBPDN(g['D'], g['S'], 0.1, cbpdn.ConvBPDN.Options())
