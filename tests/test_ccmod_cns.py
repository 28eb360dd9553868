"""ADMM consensus dictionary update (sporco_amd.admm.ccmod.ConvCnstrMOD_Consensus) and
ConvBPDNDictLearn(dmethod='cns') against fixtures produced by the unmodified reference
(oracle/make_golden.py gen_cns).  float64 1e-9; float32 against the reference's own float32
run 3e-4 (iterates), as in test_admm_cbpdn.py."""

import numpy as np
import pytest

from conftest import load_golden, rel_l2

CASES = {
    'ccmod_cns_f64': dict(opt={'MaxMainIter': 20}),
    'ccmod_cns_f32': dict(opt={'MaxMainIter': 20, 'DataType': np.float32}),
    'ccmod_cns_autorho_zm_f64': dict(opt={
        'MaxMainIter': 25, 'ZeroMean': True, 'rho': 2.0, 'RelaxParam': 1.5,
        'AutoRho': {'Enabled': True, 'Period': 2, 'Scaling': 2.0, 'RsdlRatio': 1.2,
                    'AutoScaling': True, 'RsdlTarget': 1.0}}),
    'ccmod_cns_y0_f64': dict(opt={'MaxMainIter': 10}, y0=True),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_consensus_golden_traces(backend, name):
    from sporco_amd.admm import ccmod
    g = load_golden(name)
    optd = dict(CASES[name]['opt'])
    if CASES[name].get('y0'):
        optd['Y0'] = g['Y0']
    f32 = optd.get('DataType') is np.float32
    tol = 3e-4 if f32 else 1e-9
    c = ccmod.ConvCnstrMOD_Consensus(g['Z'], g['S'], tuple(int(v) for v in g['dsz']),
                                     ccmod.ConvCnstrMOD_Consensus.Options(optd))
    Y = c.solve()
    if 'k_final' in g:
        assert c.k == int(g['k_final'])
    assert Y.shape == g['Y'].shape and rel_l2(Y, g['Y']) < tol
    assert rel_l2(c.getdict(), g['D']) < tol
    assert c.U.shape == g['U'].shape and rel_l2(c.U, g['U']) < tol
    if 'X' in g:
        assert rel_l2(c.X, g['X']) < tol
    its = c.getitstat()
    for f in ('DFid', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho'):
        assert rel_l2(getattr(its, f), g['it_' + f]) < tol, f
    assert np.max(np.abs(np.asarray(its.Cnstr) - g['it_Cnstr'])) < (1e-5 if f32 else 1e-12)
    assert c.Y.dtype == (np.float32 if f32 else np.float64)


def test_consensus_surface(backend):
    from sporco_amd.admm import ccmod
    g = load_golden('ccmod_cns_f64')
    dsz = tuple(int(v) for v in g['dsz'])
    # factory functions of the reference module
    c = ccmod.ConvCnstrMOD(g['Z'], g['S'], dsz,
                           ccmod.ConvCnstrMODOptions({'MaxMainIter': 3}, method='cns'),
                           method='cns')
    c.solve()
    c.solve()                      # continues (admm.py:331)
    assert c.k == 6
    c2 = ccmod.ConvCnstrMOD_Consensus(g['Z'], g['S'], dsz,
                                      ccmod.ConvCnstrMOD_Consensus.Options({'MaxMainIter': 6}))
    c2.solve()
    assert rel_l2(c.Y, c2.Y) < 1e-12
    assert c.reconstruct().shape[:2] == g['S'].shape[:2]
    with pytest.raises(ValueError):
        ccmod.ConvCnstrMOD(g['Z'], g['S'], dsz, method='nosuch')
    # (AuxVarObj False and LinSolveCheck: test_consensus_options below)


@pytest.mark.parametrize('name,dt,tol', [('cbpdndl_cns_f64', np.float64, 1e-9),
                                         ('cbpdndl_cns_f32', np.float32, 1e-3)])
def test_dictlearn_consensus_trace(backend, name, dt, tol):
    from sporco_amd.dictlrn import cbpdndl
    g = load_golden(name)
    opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 10, 'AccurateDFid': True},
                                            xmethod='admm', dmethod='cns')
    d = cbpdndl.ConvBPDNDictLearn(g['D0'].astype(dt), g['S'].astype(dt), float(g['lmbda']),
                                  opt, xmethod='admm', dmethod='cns')
    D1 = d.solve()
    assert rel_l2(D1, g['D1']) < tol
    assert rel_l2(d.getcoef(), g['X']) < tol
    its = d.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'XPrRsdl', 'XDlRsdl', 'XRho', 'DPrRsdl', 'DDlRsdl',
              'DRho'):
        assert rel_l2(np.asarray(getattr(its, f), dtype=float), g['it_' + f]) < tol, f


@pytest.mark.parametrize('H,K,N', [(256, 4, 1), pytest.param(256, 4, 3, marks=pytest.mark.gpu),
                                   pytest.param(256, 64, 8, marks=pytest.mark.gpu)])
def test_consensus_on_fused_kernels(backend, H, K, N):
    """The consensus X-step through rows_fwd (broadcast Y) / the register-resident column
    kernel with per-image rank-one terms / the row inverse, against the float64 oracle and
    against the generic kernel chain of the same library."""
    import os
    from oracle import cbpdn_oracle as orc
    from sporco_amd.admm import ccmod
    rng = np.random.RandomState(H + K)
    Z = (rng.randn(H, H, 1, N, K) * (rng.rand(H, H, 1, N, K) > 0.8)).astype(np.float32)
    S = rng.randn(H, H, N).astype(np.float32)
    dsz = (6, 6, K)
    optd = {'MaxMainIter': 3, 'RelStopTol': 0.0, 'rho': 5.0}

    def run(generic):
        if generic:
            os.environ['SPORCO_AMD_CNS_GENERIC'] = '1'
        try:
            c = ccmod.ConvCnstrMOD_Consensus(Z, S, dsz, ccmod.ConvCnstrMOD_Consensus.Options(optd))
            c.solve()
        finally:
            os.environ.pop('SPORCO_AMD_CNS_GENERIC', None)
        return c

    c = run(False)
    assert c.dev.uses_fused_rows()
    if K * N <= 16:
        ref = orc.admm_ccmod_cns(Z, S.reshape(H, H, 1, N, 1), dsz, dtype=np.float64, maxiter=3,
                                 rho=5.0, rel_tol=0.0)
        assert rel_l2(c.Y, ref['Y']) < 1e-5
        assert rel_l2(c.U, ref['U']) < 1e-5
        its = c.getitstat()
        for f in ('DFid', 'PrimalRsdl', 'DualRsdl'):
            assert rel_l2(getattr(its, f), ref[f]) < 1e-5, f
    if backend == 'hostsim':
        return
    c0 = run(True)
    assert rel_l2(c.Y, c0.Y) < 1e-5 and rel_l2(c.U, c0.U) < 1e-5 and rel_l2(c.X, c0.X) < 1e-5
    for f in ('DFid', 'PrimalRsdl', 'DualRsdl'):
        assert rel_l2(np.asarray(getattr(c.getitstat(), f), float),
                      np.asarray(getattr(c0.getitstat(), f), float)) < 1e-5, f
    # (Y is a projected point: the constraint violation is float32 rounding noise)
    assert max(c.getitstat().Cnstr) < 1e-5 and max(c0.getitstat().Cnstr) < 1e-5


OPTION_CASES = {
    # objective at the blocks X_n and at their mean, LinSolveCheck, ZeroMean
    'ccmod_cns_auxfalse_chk_zm_f64': {'MaxMainIter': 12, 'AuxVarObj': False, 'LinSolveCheck': True,
                                      'ZeroMean': True},
    'ccmod_cns_fevalx_f32': {'MaxMainIter': 12, 'fEvalX': True, 'DataType': np.float32},
    # the reference's own test cases (tests/admm/test_ccmod.py:153-259): a multi-channel signal
    # with a single-channel dictionary, dimK = 0 and several images, LinSolveCheck
    'ccmod_cns_chk_multichan_dimk0_f64': {'MaxMainIter': 12, 'LinSolveCheck': True},
    'ccmod_cns_chk_multichan_f64': {'MaxMainIter': 12, 'LinSolveCheck': True},
}


@pytest.mark.parametrize('name', sorted(OPTION_CASES))
def test_consensus_options(backend, name):
    """Fixtures of oracle/make_golden.py gen_cns_options (the unmodified reference)."""
    from sporco_amd.admm import ccmod
    g = load_golden(name)
    optd = dict(OPTION_CASES[name])
    f32 = optd.get('DataType') is np.float32
    tol = 3e-4 if f32 else 1e-9
    c = ccmod.ConvCnstrMOD_Consensus(g['Z'], g['S'], tuple(int(v) for v in g['dsz']),
                                     ccmod.ConvCnstrMOD_Consensus.Options(optd), dimK=int(g['dimK']))
    Y = c.solve()
    assert c.k == int(g['k_final'])
    assert Y.shape == g['Y'].shape and rel_l2(Y, g['Y']) < tol
    assert rel_l2(c.getdict(), g['D']) < tol and rel_l2(c.U, g['U']) < tol
    assert rel_l2(c.X, g['X']) < tol
    its = c.getitstat()
    for f in ('DFid', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho'):
        assert rel_l2(getattr(its, f), g['it_' + f]) < tol, f
    assert np.max(np.abs(np.asarray(its.Cnstr) - g['it_Cnstr'])) < (1e-4 if f32 else 1e-11)
    if optd.get('LinSolveCheck'):
        # a relative residual at rounding level on both sides (the reference's own test asks
        # for < 1e-5, tests/admm/test_ccmod.py:168)
        assert max(its.XSlvRelRes) < 1e-10 and np.max(g['it_XSlvRelRes']) < 1e-10
    else:
        assert all(v is None for v in its.XSlvRelRes)


MCDICT_CASES = {
    'ccmod_cns_mcdict_f64': {'MaxMainIter': 12},
    'ccmod_cns_mcdict_opts_f64': {'MaxMainIter': 12, 'LinSolveCheck': True, 'ZeroMean': True,
                                  'AuxVarObj': False},
    'ccmod_cns_mcdict_f32': {'MaxMainIter': 12, 'DataType': np.float32},
}


@pytest.mark.parametrize('name', sorted(MCDICT_CASES))
def test_consensus_multichannel_dictionary(backend, name):
    """The consensus update of a colour dictionary (Cd = C = 3): one (Cd, M) block per image,
    whose channels share the image's system matrix (sporco/admm/ccmod.py:696-698, :766-822).
    Fixtures of oracle/make_golden.py gen_cns_mcdict (the unmodified reference)."""
    from sporco_amd.admm import ccmod
    g = load_golden(name)
    optd = dict(MCDICT_CASES[name])
    f32 = optd.get('DataType') is np.float32
    tol = 3e-4 if f32 else 1e-9
    c = ccmod.ConvCnstrMOD_Consensus(g['Z'], g['S'], tuple(int(v) for v in g['dsz']),
                                     ccmod.ConvCnstrMOD_Consensus.Options(optd))
    Y = c.solve()
    assert c.k == int(g['k_final'])
    assert Y.shape == g['Y'].shape and rel_l2(Y, g['Y']) < tol
    assert c.getdict().shape == g['D'].shape and rel_l2(c.getdict(), g['D']) < tol
    assert c.U.shape == g['U'].shape and rel_l2(c.U, g['U']) < tol
    assert c.X.shape == g['X'].shape and rel_l2(c.X, g['X']) < tol
    its = c.getitstat()
    for f in ('DFid', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho'):
        assert rel_l2(getattr(its, f), g['it_' + f]) < tol, f
    assert np.max(np.abs(np.asarray(its.Cnstr) - g['it_Cnstr'])) < (1e-4 if f32 else 1e-11)
    if optd.get('LinSolveCheck'):
        assert max(its.XSlvRelRes) < 1e-10


ZCHAN_CASES = {
    'ccmod_cns_zchan_f64': {'MaxMainIter': 20, 'LinSolveCheck': True},
    'ccmod_cns_zchan_opts_f32': {'MaxMainIter': 12, 'ZeroMean': True, 'AuxVarObj': False,
                                 'DataType': np.float32},
}


@pytest.mark.parametrize('name', sorted(ZCHAN_CASES))
def test_consensus_channelful_maps_with_multichannel_dictionary(backend, name):
    """(N, N, Nc, K, M) coefficient maps with an (Nd, Nd, Nc, M) dictionary: every (image,
    channel) block has a system matrix of its own, rho and the residuals are shared (the
    reference reaches this by broadcasting; its tests/admm/test_ccmod.py:278-295).
    Fixtures of oracle/make_golden.py gen_zchan (the unmodified reference)."""
    from sporco_amd.admm import ccmod
    g = load_golden(name)
    optd = dict(ZCHAN_CASES[name])
    f32 = optd.get('DataType') is np.float32
    tol = 3e-4 if f32 else 1e-9
    c = ccmod.ConvCnstrMOD_Consensus(g['Z'], g['S'], tuple(int(v) for v in g['dsz']),
                                     ccmod.ConvCnstrMOD_Consensus.Options(optd))
    Y = c.solve()
    assert c.k == int(g['k_final'])
    assert Y.shape == g['Y'].shape and rel_l2(Y, g['Y']) < tol
    assert c.getdict().shape == g['D'].shape and rel_l2(c.getdict(), g['D']) < tol
    assert c.U.shape == g['U'].shape and rel_l2(c.U, g['U']) < tol
    assert c.X.shape == g['X'].shape and rel_l2(c.X, g['X']) < tol
    its = c.getitstat()
    for f in ('DFid', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho'):
        assert rel_l2(getattr(its, f), g['it_' + f]) < tol, f
    assert np.max(np.abs(np.asarray(its.Cnstr) - g['it_Cnstr'])) < (1e-4 if f32 else 1e-11)
    if optd.get('LinSolveCheck'):
        assert max(its.XSlvRelRes) < 1e-10
    # the per-channel reconstruction sum_m Z_c,k,m * D_c,m
    Zf = np.fft.rfftn(g['Z'], axes=(0, 1))
    ref = np.fft.irfftn(np.sum(Zf * np.fft.rfftn(g['Y'], axes=(0, 1)), axis=4), (16, 16), axes=(0, 1))
    assert rel_l2(c.reconstruct(), ref) < tol


def test_dictlearn_consensus_colour_dictionary(backend):
    """ConvBPDNDictLearn(dmethod='cns') learning a colour dictionary (the reference's
    examples/scripts/cdl/cbpdndl_cns_clr.py in miniature)."""
    from sporco_amd.dictlrn import cbpdndl
    g = load_golden('cbpdndl_cns_mcdict_f64')
    opt = cbpdndl.ConvBPDNDictLearn.Options({'MaxMainIter': 8, 'AccurateDFid': True},
                                            xmethod='admm', dmethod='cns')
    d = cbpdndl.ConvBPDNDictLearn(g['D0'], g['S'], float(g['lmbda']), opt, xmethod='admm',
                                  dmethod='cns')
    D1 = d.solve()
    assert rel_l2(D1.squeeze(), g['D1'].squeeze()) < 1e-9
    assert rel_l2(d.getcoef(), g['X']) < 1e-9
    its = d.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'XPrRsdl', 'XDlRsdl', 'DPrRsdl', 'DDlRsdl', 'DRho'):
        assert rel_l2(np.asarray(getattr(its, f), float), g['it_' + f]) < 1e-9, f
