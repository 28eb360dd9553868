"""The fused X-step (csc_fused.hip: column FFT + Sherman-Morrison + column IFFT in
registers, tile-major intermediates) against (a) the NumPy oracle and (b) the
unfused kernel chain of the same library.

The fused path engages for float32, H in {256, 512}, even K <= 64; the widths
are kept tiny so that the CPU fiber simulator finishes in seconds (the row
kernels are size-generic, so W does not matter to the code under test).

Tolerances: 1e-4 relative l2 on Y against the float64 oracle (the BASELINE bar);
fused vs unfused float32 runs agree to 2e-5 after 12 adaptive-rho iterations
(same arithmetic up to summation order and a 1-ulp reciprocal).
"""

import os

import numpy as np
import pytest

from conftest import rel_l2


def problem(H, W, K, N, seed, C=None):
    rng = np.random.RandomState(seed)
    D = rng.randn(4, 4, K).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    shape = (H, W, N) if C is None else (H, W, C, N)
    S = rng.randn(*shape).astype(np.float32)
    return D, S


def solve(D, S, optd, unfused=False, joint=False):
    from sporco_amd.admm import cbpdn
    if unfused:
        os.environ['SPORCO_AMD_UNFUSED'] = '1'
    try:
        if joint:
            b = cbpdn.ConvBPDNJoint(D, S, 0.05, 0.02, cbpdn.ConvBPDNJoint.Options(optd))
        else:
            b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
    finally:
        os.environ.pop('SPORCO_AMD_UNFUSED', None)
    Y = b.solve()
    return b, Y


@pytest.mark.parametrize('H,W,K,N', [(256, 16, 8, 2), (512, 12, 64, 1), (512, 8, 6, 3),
                                     (256, 10, 64, 2)])
def test_fused_matches_oracle_and_unfused(backend, H, W, K, N):
    from oracle import cbpdn_oracle as orc
    D, S = problem(H, W, K, N, seed=H + K)
    optd = {'MaxMainIter': 12, 'RelStopTol': 0.0}
    b, Y = solve(D, S, optd)
    b0, Y0 = solve(D, S, optd, unfused=True)
    ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05,
                         dtype=np.float64, maxiter=12, rel_tol=0.0)
    assert rel_l2(Y, ref['Y']) < 1e-4
    assert rel_l2(Y, Y0) < 2e-5
    its, its0 = b.getitstat(), b0.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(its, f), ref[f]) < 1e-3, f
        assert rel_l2(getattr(its, f), getattr(its0, f)) < 1e-4, f
    # X and Xf stay available after a fused step (Xf is rebuilt from X on demand)
    assert rel_l2(b.X, ref['X']) < 1e-4
    assert rel_l2(b.Xf, np.fft.rfftn(b.X.astype(np.float64), axes=(0, 1))) < 1e-5
    assert rel_l2(b.reconstruct(), b0.reconstruct()) < 2e-5


def test_fused_joint_multichannel(backend):
    from oracle import cbpdn_oracle as orc
    H, W, K, N, C = 256, 8, 16, 2, 3
    D, S = problem(H, W, K, N, seed=5, C=C)
    optd = {'MaxMainIter': 10, 'RelStopTol': 0.0}
    b, Y = solve(D, S, optd, joint=True)
    ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, C, N, 1), 0.05, mu=0.02,
                         dtype=np.float64, maxiter=10, rel_tol=0.0)
    assert rel_l2(Y, ref['Y']) < 1e-4
    assert rel_l2(b.getitstat().ObjFun, ref['ObjFun']) < 1e-3


def test_fused_fixed_rho_fastsolve_and_setdict(backend):
    """FastSolve (no sums read back) and a dictionary change between solves."""
    H, W, K, N = 256, 8, 8, 2
    D, S = problem(H, W, K, N, seed=9)
    optd = {'MaxMainIter': 6, 'RelStopTol': 0.0, 'rho': 2.0, 'FastSolve': True,
            'AutoRho': {'Enabled': False}}
    b, Y = solve(D, S, optd)
    b0, Y0 = solve(D, S, optd, unfused=True)
    assert rel_l2(Y, Y0) < 1e-5
    D2 = D[..., ::-1].copy()
    for s in (b, b0):
        s.setdict(D2.reshape(s.cri.shpD))
        s.solve()
    assert rel_l2(b.Y, b0.Y) < 1e-5
