"""ADMM ConvBPDN with mask decoupling (sporco_amd.admm.cbpdn.ConvBPDNMaskDcpl) against fixtures
produced by the unmodified reference (oracle/make_golden.py gen_maskdcpl): both blocks of Y and
U, X, and the IterationStats traces.  float64 1e-9; float32 against the reference's own float32
run 5e-4, as in test_admm_cbpdn.py."""

import numpy as np
import pytest

from conftest import load_golden, rel_l2

AUTORHO = {'Enabled': True, 'Period': 3, 'Scaling': 2.0, 'RsdlRatio': 1.2, 'AutoScaling': True,
           'RsdlTarget': 1.0}
CASES = {
    'maskdcpl_f64': {'MaxMainIter': 30},
    'maskdcpl_f32': {'MaxMainIter': 30, 'DataType': np.float32},
    'maskdcpl_autorho_opts_f64': {'MaxMainIter': 30, 'rho': 2.0, 'RelaxParam': 1.5,
                                  'NonNegCoef': True, 'NoBndryCross': True, 'AuxVarObj': True,
                                  'LinSolveCheck': True, 'AutoRho': AUTORHO},
    'maskdcpl_multichan_f64': {'MaxMainIter': 20},
    # multi-channel dictionaries (Cd = 3): the reference's own tests/admm/test_cbpdn.py:487-518
    'maskdcpl_mcdict_f64': {'MaxMainIter': 20},
    'maskdcpl_mcdict_f32': {'MaxMainIter': 20, 'DataType': np.float32},
    'maskdcpl_mcdict_one_image_opts_f64': {'MaxMainIter': 20, 'rho': 1.5, 'RelaxParam': 1.6,
                                           'NonNegCoef': True, 'AuxVarObj': True,
                                           'LinSolveCheck': True,
                                           'AutoRho': dict(AUTORHO, Period=2)},
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_golden_traces(backend, name):
    from sporco_amd.admm import cbpdn
    g = load_golden(name)
    optd = dict(CASES[name])
    if 'wl1' in g:
        optd['L1Weight'] = g['wl1']
    f32 = optd.get('DataType') is np.float32
    tol = 5e-4 if f32 else 1e-9
    b = cbpdn.ConvBPDNMaskDcpl(g['D'], g['S'], float(g['lmbda']), g['W'],
                               cbpdn.ConvBPDNMaskDcpl.Options(optd))
    Y1 = b.solve()
    assert b.k == int(g['k_final'])
    assert Y1.shape == g['Y1'].shape and rel_l2(Y1, g['Y1']) < tol
    assert Y1.dtype == (np.float32 if f32 else np.float64)
    assert rel_l2(b.X, g['X']) < tol
    assert b.Y.shape == g['Y'].shape and rel_l2(b.Y, g['Y']) < tol
    if 'Y0' in g:       # (multi-channel dictionary: block 0 in the signal's own layout)
        assert b.var_y0().shape == g['Y0'].shape and rel_l2(b.var_y0(), g['Y0']) < tol
    else:
        assert rel_l2(b.var_y0(), g['Y'][..., :1]) < tol
    assert b.U.shape == g['U'].shape and rel_l2(b.U, g['U']) < tol
    assert rel_l2(b.reconstruct().squeeze(), g['recon'].squeeze()) < tol
    assert rel_l2(float(b.rho), float(g['rho_final'])) < tol
    its = b.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho'):
        assert rel_l2(getattr(its, f), g['it_' + f]) < tol, f
    if optd.get('LinSolveCheck'):
        assert np.max(np.abs(np.asarray(its.XSlvRelRes) - g['it_XSlvRelRes'])) < 1e-12
        assert max(its.XSlvRelRes) < 1e-12
    else:
        assert all(v is None for v in its.XSlvRelRes)


def test_surface(backend):
    from sporco_amd.admm import cbpdn
    g = load_golden('maskdcpl_f64')
    cls = cbpdn.ConvBPDNMaskDcpl
    opt = cls.Options()
    assert opt['rho'] == 1.0 and opt['RelaxParam'] == 1.8 and opt['ReturnVar'] == 'Y1'
    assert not opt['AutoRho', 'Enabled']
    # no mask = all ones; ReturnVar selects what solve() returns; solve() continues
    b = cls(g['D'], g['S'], 0.1, None, cls.Options({'MaxMainIter': 3, 'ReturnVar': 'X'}))
    X = b.solve()
    assert np.array_equal(X, b.X) and b.k == 3
    b.solve()
    assert b.k == 6
    b1 = cls(g['D'], g['S'], 0.1, np.ones(g['S'].shape), cls.Options({'MaxMainIter': 6}))
    assert rel_l2(b1.solve(), b.var_y1()) < 1e-12
    b0 = cls(g['D'], g['S'], 0.1, g['W'], cls.Options({'MaxMainIter': 2, 'ReturnVar': 'Y0'}))
    assert b0.solve().shape == b0.cri.shpS
    with pytest.raises(ValueError):
        cls(g['D'], g['S'], 0.1, None, cls.Options({'ReturnVar': 'Z'}))
    with pytest.raises(ValueError):           # a warm start must be a [block 0; block 1] array
        cls(g['D'], g['S'], 0.1, None, cls.Options({'Y0': np.zeros((16, 16, 1, 2, 2))}))



def test_masked_dictionary_learning_admm_xstep(backend):
    """ConvBPDNMaskDictLearn with the mask-decoupling X-step and the PGM D-step
    (cbpdndlmd.py:219-543; the reference's default xmethod)."""
    from sporco_amd.dictlrn import cbpdndlmd
    g = load_golden('cbpdndlmd_admm_pgm_f64')
    opt = cbpdndlmd.ConvBPDNMaskDictLearn.Options({'MaxMainIter': 10, 'AccurateDFid': True},
                                                  dmethod='pgm')
    assert opt.xmethod == 'admm' and opt['CBPDN', 'AutoRho', 'Period'] == 10
    d = cbpdndlmd.ConvBPDNMaskDictLearn(g['D0'], g['S'], float(g['lmbda']), g['W'], opt,
                                        xmethod='admm', dmethod='pgm')
    D1 = d.solve()
    assert rel_l2(D1.squeeze(), g['D1'].squeeze()) < 1e-9
    assert rel_l2(d.getcoef(), g['X']) < 1e-9
    its = d.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'XPrRsdl', 'XDlRsdl', 'XRho', 'D_L', 'D_Rsdl'):
        assert rel_l2(np.asarray(getattr(its, f), float), g['it_' + f]) < 1e-9, f
    assert max(its.Cnstr) < 1e-10


def test_odd_filter_count_against_oracle(backend):
    """Five filters, NoBndryCross, AutoRho: the X-step against the float64 oracle."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(3)
    H, K, N = 32, 5, 2
    D, S = rng.randn(6, 6, K), rng.randn(H, H, N)
    W = (rng.rand(H, H, N) > 0.3).astype(float)
    cls = cbpdn.ConvBPDNMaskDcpl
    b = cls(D, S, 0.1, W, cls.Options({'MaxMainIter': 8, 'NoBndryCross': True,
                                      'AutoRho': {'Enabled': True, 'Period': 2}}))
    Y1 = b.solve()
    r = orc.admm_cbpdn_maskdcpl(D.reshape(6, 6, 1, 1, K), S.reshape(H, H, 1, N, 1), 0.1,
                                W.reshape(H, H, 1, N, 1), maxiter=8, nobndry=True, auto_rho=True,
                                rho_period=2)
    assert rel_l2(Y1, r['Y1']) < 1e-9 and rel_l2(b.var_y0(), r['Y0']) < 1e-9
    for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(b.getitstat(), f), r[f]) < 1e-9, f


def test_image_size_float32_against_oracle(backend):
    """256 x 256 float32 -- a shape whose handle also carries the register-resident ConvBPDN
    kernels; mask decoupling runs the generic chain on it -- against the float64 oracle."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(11)
    H, K = 256, 4
    D = rng.randn(8, 8, K).astype(np.float32)
    S = rng.randn(H, H, 1).astype(np.float32)
    W = (rng.rand(H, H, 1) > 0.3).astype(np.float32)
    cls = cbpdn.ConvBPDNMaskDcpl
    b = cls(D, S, 0.1, W, cls.Options({'MaxMainIter': 3}))
    Y1 = b.solve()
    r = orc.admm_cbpdn_maskdcpl(D.reshape(8, 8, 1, 1, K).astype(np.float64),
                                S.reshape(H, H, 1, 1, 1).astype(np.float64), 0.1,
                                W.reshape(H, H, 1, 1, 1).astype(np.float64), maxiter=3)
    assert rel_l2(Y1, r['Y1']) < 1e-5
    for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl'):
        assert rel_l2(getattr(b.getitstat(), f), r[f]) < 1e-5, f


def test_warm_start_and_pickling(backend):
    """`Y0` / `U0` warm starts (two-block arrays, sporco/admm/admm.py:262-272) and pickling
    (the reference's objects pickle: tests/admm/test_cbpdn.py:631-644): a run of 4 + 4
    iterations through a pickle, and through a fresh object started from the first run's
    (Y, U), equals one run of 8."""
    import pickle
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(5)
    H = 32
    D = rng.randn(4, 4, 5)
    S = rng.randn(H, H, 2)
    W = (rng.rand(H, H, 2) > 0.3).astype(np.float64)

    def make(n, **kw):
        optd = dict({'MaxMainIter': n, 'RelStopTol': 0.0, 'AutoRho': {'Enabled': True, 'Period': 2}}, **kw)
        return cbpdn.ConvBPDNMaskDcpl(D, S, 0.1, W, cbpdn.ConvBPDNMaskDcpl.Options(optd))
    full = make(8)
    full.solve()
    a = make(4)
    a.solve()
    b = pickle.loads(pickle.dumps(a))
    b.solve()                                   # 4 more iterations, continuing at k = 4
    assert b.k == 8
    assert np.array_equal(b.Y, full.Y) and np.array_equal(b.U, full.U)
    assert np.array_equal(np.asarray(b.getitstat().ObjFun), np.asarray(full.getitstat().ObjFun))
    # warm start: a new object from (Y, U) of the 4-iteration run; rho and the iteration
    # counter are not part of a warm start, so compare against a reference-style restart
    c = make(3, Y0=a.Y, U0=a.U, rho=float(a.rho), AutoRho={'Enabled': False})
    c.solve()
    d = make(4)
    d.solve()
    d.opt['AutoRho', 'Enabled'] = False
    d.opt['MaxMainIter'] = 3
    d.solve()
    assert rel_l2(c.Y, d.Y) < 1e-12 and rel_l2(c.U, d.U) < 1e-12


# ---- the iteration on the register-resident kernels (round 4) ---------------------------------------
FUSED_OPTS = {
    'default': {},
    'options': {'NonNegCoef': True, 'NoBndryCross': True, 'AuxVarObj': True, 'RelaxParam': 1.5,
                'AutoRho': {'Enabled': True, 'Period': 2}},
    'fastsolve': {'FastSolve': True, 'AutoRho': {'Enabled': False}},
}


@pytest.mark.parametrize('H,K,N', [(128, 8, 2), pytest.param(512, 64, 2, marks=pytest.mark.gpu)])
@pytest.mark.parametrize('case', sorted(FUSED_OPTS))
def test_fused_iteration_vs_generic_chain_and_oracle(backend, case, H, K, N, monkeypatch):
    """ConvBPDNMaskDcpl on the fast-path shapes (float32, 128 / 256 / 512, even K <= 64) runs its
    X-step, block-1 epilogue and dual residual on the register-resident kernels of ConvBPDN
    (rows_fwd, fused_cols with the block-0 spectrum in the signal's place, rows_inv_post; DESIGN
    4.9): same iterates and statistics as the generic chain of the same library (which the
    reference fixtures above pin), and the float64 oracle within the float32 bar."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(11)
    D = rng.randn(8, 8, K).astype(np.float32)
    S = rng.randn(H, H, N).astype(np.float32)
    W = (rng.rand(H, H, N) > 0.3).astype(np.float32)
    cls = cbpdn.ConvBPDNMaskDcpl
    optd = dict(FUSED_OPTS[case], MaxMainIter=5)

    def run(generic):
        if generic:
            monkeypatch.setenv('SPORCO_AMD_MD_GENERIC', '1')
        else:
            monkeypatch.delenv('SPORCO_AMD_MD_GENERIC', raising=False)
        b = cls(D, S, 0.1, W, cls.Options(optd))
        b._dev.profile(True)
        Y1 = b.solve()
        prof = {k for k, v in b._dev.profile_read().items() if v[1]}
        return b, Y1, prof

    bf, Yf, pf = run(False)
    bg, Yg, pg = run(True)
    assert {'rows_fwd', 'fused_cols_sm', 'rows_inv_post'} <= pf and 'sm_solve' not in pf
    assert 'sm_solve' in pg and 'fused_cols_sm' not in pg
    # (two float32 evaluations of the same iteration: the float32 bar; measured on the MI355X at
    # 512 x 512, K = 64 against the float64 run of the generic chain: Y 1.4e-5 fused, 3e-6 generic
    # -- Y is nearly empty after five iterations --, X 2e-6, U 1.5e-6 / 6e-6, y0 1e-4 / 5e-4)
    assert rel_l2(Yf, Yg) < 1e-4 and rel_l2(bf.X, bg.X) < 1e-4 and rel_l2(bf.U, bg.U) < 1e-4
    assert rel_l2(bf.var_y0(), bg.var_y0()) < 1e-3
    assert rel_l2(bf.reconstruct(), bg.reconstruct()) < 1e-4
    if case != 'fastsolve':
        for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho'):
            assert rel_l2(getattr(bf.getitstat(), f), getattr(bg.getitstat(), f)) < 1e-4, f
    if case == 'default' and H <= 128:
        r = orc.admm_cbpdn_maskdcpl(D.reshape(8, 8, 1, 1, K).astype(np.float64),
                                    S.reshape(H, H, 1, N, 1).astype(np.float64), 0.1,
                                    W.reshape(H, H, 1, N, 1).astype(np.float64), maxiter=5)
        assert rel_l2(Yf, r['Y1']) < 1e-4
        for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
            assert rel_l2(getattr(bf.getitstat(), f), r[f]) < 1e-4, f
