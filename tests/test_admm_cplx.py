"""admm.cbpdn.ConvBPDN on complex-valued signals and dictionaries (sporco/admm/cbpdn.py:209-217;
the reference's tests/admm/test_cbpdn.py:179-201): the real and imaginary parts run as the channel
pair of the real machinery (sporco_amd/admm/cbpdn_cplx.py, csrc/ck_admm.hip sm_cplx_kernel) --
against runs of the unmodified reference (tests/golden/admm_cplx_*.npz)."""

import numpy as np
import pytest

from conftest import load_golden, rel_l2

CASES = {
    'admm_cplx_default_f64': ({'MaxMainIter': 30}, 1e-10),
    'admm_cplx_fixedrho_auxvar_f64': ({'MaxMainIter': 25, 'rho': 2.0, 'RelaxParam': 1.0,
                                       'AutoRho': {'Enabled': False}, 'AuxVarObj': True}, 1e-10),
    'admm_cplx_default_f32': ({'MaxMainIter': 30}, 2e-4),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_complex_convbpdn_against_the_reference(backend, name):
    from sporco_amd.admm import cbpdn
    g = load_golden(name)
    optd, tol = CASES[name]
    b = cbpdn.ConvBPDN(g['D'], g['S'], float(g['lmbda']), cbpdn.ConvBPDN.Options(optd))
    Y = b.solve()
    assert np.iscomplexobj(Y) and Y.dtype == g['Y'].dtype and Y.shape == g['Y'].shape
    assert rel_l2(Y, g['Y']) < tol and rel_l2(b.X, g['X']) < tol and rel_l2(b.U, g['U']) < tol
    assert rel_l2(b.reconstruct(), g['recon']) < tol
    assert abs(float(b.rho) - float(g['rho_final'])) < tol * float(g['rho_final'])
    its = b.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'EpsPrimal', 'EpsDual', 'Rho'):
        assert rel_l2(np.asarray(getattr(its, f), float), g['it_' + f]) < 10 * tol, f


def test_reference_test_10cplx_setting(backend):
    """tests/admm/test_cbpdn.py:179-201 of the reference: recovery of a sparse complex code."""
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(3)
    N, M, Nd = 32, 4, 8
    D = rng.randn(Nd, Nd, M) + 1j * rng.randn(Nd, Nd, M)
    X0 = np.zeros((N, N, M), complex)
    xp = np.abs(rng.randn(N, N, M)) > 3
    X0[xp] = rng.randn(int(xp.sum())) + 1j * rng.randn(int(xp.sum()))
    S = np.sum(np.fft.ifft2(np.fft.fft2(D, (N, N), axes=(0, 1)) * np.fft.fft2(X0, axes=(0, 1)), axes=(0, 1)), axis=2)
    opt = cbpdn.ConvBPDN.Options({'Verbose': False, 'MaxMainIter': 500, 'RelStopTol': 1e-3, 'rho': 1e-1,
                                  'AutoRho': {'Enabled': False}})
    b = cbpdn.ConvBPDN(D, S, 1e-4, opt)
    b.solve()
    X1 = b.Y.squeeze()
    assert rel_l2(X1, X0) < 5e-5 ** 0.5 * 10      # (rrs < 5e-5 in the reference's metric)
    assert rel_l2(b.reconstruct().squeeze(), S) < 1e-2


def test_what_complex_mode_does_not_take(backend):
    from sporco_amd.admm import cbpdn
    rng = np.random.RandomState(1)
    D = rng.randn(4, 4, 3) + 1j * rng.randn(4, 4, 3)
    S = rng.randn(16, 16) + 1j * rng.randn(16, 16)
    with pytest.raises(NotImplementedError):
        cbpdn.ConvBPDN(D, S, None)
    with pytest.raises(NotImplementedError):
        cbpdn.ConvBPDN(D, S, 0.1, cbpdn.ConvBPDN.Options({'NonNegCoef': True}))
    with pytest.raises(NotImplementedError):
        cbpdn.ConvBPDN(D, rng.randn(16, 16, 3, 2) + 0j, 0.1, dimK=1)       # multi-channel signal
