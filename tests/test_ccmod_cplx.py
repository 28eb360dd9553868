"""Complex-valued coefficient maps, signals and dictionary in the ADMM dictionary updates
(sporco_amd.admm.ccmod; the reference's fftn path, sporco/admm/ccmod.py:219-231, and its own tests
tests/admm/test_ccmod.py:49-140) against runs of the unmodified reference
(oracle/make_golden.py gen_ccmod_cplx).  The real and the imaginary part are the two channels of a
real handle in SPORCO_AMD_MODE_COMPLEX_PAIR (include/sporco_amd.h).

Tolerances as in tests/test_ccmod_ism_cg.py: the direct solves 1e-9 in float64 and 5e-4 in
complex64; CG at its default stopping tolerance of 1e-3 a few times that, run to 1e-9: 1e-7."""

import numpy as np
import pytest

from conftest import load_golden, rel_l2

CASES = {
    ('ism', 'f64'): ({'MaxMainIter': 20}, 1e-9),
    ('ism', 'f32'): ({'MaxMainIter': 20, 'DataType': np.complex64}, 5e-4),
    ('ism', 'chk_zm_y0_f64'): ({'MaxMainIter': 15, 'LinSolveCheck': True, 'ZeroMean': True, 'RelaxParam': 1.5,
                                'AuxVarObj': True}, 1e-9),
    ('cg', 'tight_f64'): ({'MaxMainIter': 15, 'rho': 2.0, 'AutoRho': {'Enabled': False}, 'LinSolveCheck': True,
                           'CG': {'MaxIter': 500, 'StopTol': 1e-9}}, 1e-7),
    ('cg', 'f64'): ({'MaxMainIter': 15}, 5e-3),
    ('cns', 'f64'): ({'MaxMainIter': 20}, 1e-9),
    ('cns', 'chk_zm_autorho_f64'): ({'MaxMainIter': 15, 'LinSolveCheck': True, 'ZeroMean': True,
                                     'AutoRho': {'Enabled': True}}, 1e-9),
    ('ism', 'odd_single_f64'): ({'MaxMainIter': 15}, 1e-9),
}


def dstep_class(method):
    from sporco_amd.admm import ccmod
    return {'ism': ccmod.ConvCnstrMOD_IterSM, 'cg': ccmod.ConvCnstrMOD_CG,
            'cns': ccmod.ConvCnstrMOD_Consensus}[method]


@pytest.mark.parametrize('method,case', sorted(CASES))
def test_golden_traces(backend, method, case):
    if backend == 'hostsim' and (method, case) == ('cg', 'tight_f64'):
        pytest.skip("kept for the GPU run (slow on the CPU simulator)")
    g = load_golden('ccmod_cplx_%s_%s' % (method, case))
    optd, tol = CASES[(method, case)]
    optd = dict(optd)
    if 'y0' in case or case == 'chk_zm_autorho_f64':
        optd['Y0'] = g['Y0']
    cls = dstep_class(method)
    kw = {'dimK': 0} if 'single' in case else {}
    c = cls(g['Z'], g['S'], tuple(int(v) for v in g['dsz']), cls.Options(optd), **kw)
    Y = c.solve()
    cdt = np.complex64 if case == 'f32' else np.complex128
    assert Y.dtype == cdt and c.cdtype == cdt
    assert c.k == int(g['k_final'])
    assert Y.shape == g['Y'].shape and rel_l2(Y, g['Y']) < tol
    assert c.getdict().shape == g['D'].shape and rel_l2(c.getdict(), g['D']) < tol
    assert c.U.shape == g['U'].shape and rel_l2(c.U, g['U']) < tol
    assert c.X.shape == g['X'].shape and rel_l2(c.X, g['X']) < tol
    its = c.getitstat()
    for f in ('DFid', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho'):
        assert rel_l2(getattr(its, f), g['it_' + f]) < tol, f
    assert np.max(np.abs(np.asarray(its.Cnstr) - g['it_Cnstr'])) < max(10 * tol, 1e-9)
    if optd.get('LinSolveCheck'):
        assert np.max(np.abs(np.asarray(its.XSlvRelRes) - g['it_XSlvRelRes'])) < 1e-8
    if method == 'cg':
        assert np.array_equal(np.asarray(its.XSlvCGIt), g['it_XSlvCGIt'])
    # unit complex norm, support, zero mean
    D = c.getdict(crop=False)
    assert np.allclose(np.sum(np.abs(D) ** 2, axis=(0, 1)).ravel(), 1.0, atol=1e-5 if case == 'f32' else 1e-12)
    dH, dW = int(g['dsz'][0]), int(g['dsz'][1])
    assert np.all(D[dH:] == 0) and np.all(D[:, dW:] == 0)
    if optd.get('ZeroMean'):
        assert np.abs(np.sum(D, axis=(0, 1))).max() < 1e-12


def test_surface(backend):
    """Reconstruction, the setters, and what the complex form does not take."""
    from sporco_amd.admm import ccmod, ccmodmd
    g = load_golden('ccmod_cplx_ism_f64')
    dsz = tuple(int(v) for v in g['dsz'])
    c = ccmod.ConvCnstrMOD_IterSM(g['Z'], g['S'], dsz, ccmod.ConvCnstrMOD_IterSM.Options({'MaxMainIter': 5}))
    c.solve()
    Sr = c.reconstruct()
    Df, Zf = np.fft.fftn(c.X, axes=(0, 1)), np.fft.fftn(g['Z'], axes=(0, 1))
    assert rel_l2(Sr, np.fft.ifftn(np.sum(Zf * Df, axis=4), axes=(0, 1))) < 1e-12
    assert rel_l2(c.Xf, Df) == 0.0
    # state written through the setters continues the run of an object that was never interrupted
    a = ccmod.ConvCnstrMOD_IterSM(g['Z'], g['S'], dsz, ccmod.ConvCnstrMOD_IterSM.Options(
        {'MaxMainIter': 3, 'rho': 2.0, 'AutoRho': {'Enabled': False}}))
    a.solve()
    b = ccmod.ConvCnstrMOD_IterSM(g['Z'], g['S'], dsz, ccmod.ConvCnstrMOD_IterSM.Options(
        {'MaxMainIter': 3, 'rho': 2.0, 'AutoRho': {'Enabled': False}}))
    b.solve()
    y, u = a.Y.copy(), a.U.copy()
    a.Y, a.U = y, u
    a.solve()
    b.solve()
    assert rel_l2(a.Y, b.Y) < 1e-13
    # a real problem handed over as complex arrays gives the real problem's result
    gr = load_golden('ccmod_ism_f64')
    r = ccmod.ConvCnstrMOD_IterSM(gr['Z'].astype(np.complex128), gr['S'].astype(np.complex128),
                                  tuple(int(v) for v in gr['dsz']),
                                  ccmod.ConvCnstrMOD_IterSM.Options({'MaxMainIter': 20}))
    Y = r.solve()
    assert np.abs(Y.imag).max() < 1e-12 and rel_l2(Y.real, gr['Y']) < 1e-9
    with pytest.raises(NotImplementedError):
        ccmod.ConvCnstrMOD_Consensus(g['Z'], g['S'], dsz, ccmod.ConvCnstrMOD_Consensus.Options({'AuxVarObj': False}))
    with pytest.raises(NotImplementedError):
        ccmod.ConvCnstrMOD_IterSM(g['Z'], g['S'][:, :, np.newaxis].repeat(2, axis=2), dsz,
                                  ccmod.ConvCnstrMOD_IterSM.Options())
    W = np.ones(g['S'].shape)
    with pytest.raises((NotImplementedError, TypeError)):
        ccmodmd.ConvCnstrMODMaskDcpl_Consensus(g['Z'], g['S'], W, dsz)
