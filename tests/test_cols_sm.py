"""The generic X-step's column pass as one kernel (fft.h fft_cols_sm): forward transform along H
in place in LDS (decimation in frequency), Sherman-Morrison solve at the digit-reversed
positions (sporco/linalg.py:232-297), inverse transform (decimation in time) -- against the
three kernels it replaces (SPORCO_AMD_NO_COLS_SM=1: fft_c2c, launch_sm_solve, fft_c2c), which
the float64 fixtures of test_admm_cbpdn.py pin to the reference; plus the oracle directly."""

import os

import numpy as np
import pytest

from conftest import rel_l2

CASES = {
    # H, W, K, N, dtype       (lengths with radices 2, 3, 4, 5, 7, 8)
    'f32_48x40_k8': (48, 40, 8, 2, np.float32),
    'f64_30x36_k4': (30, 36, 4, 2, np.float64),
    'f64_63x56_k16': (63, 56, 16, 1, np.float64),
    'f32_32x24_k64': (32, 24, 64, 1, np.float32),
    'f64_35x20_k2': (35, 20, 2, 3, np.float64),
    'f32_96x80_k32': (96, 80, 32, 2, np.float32),
    'f64_36x30_k6': (36, 30, 6, 2, np.float64),        # filter counts that are not powers of two
    'f32_40x48_k24': (40, 48, 24, 1, np.float32),
    'f64_24x40_k50': (24, 40, 50, 1, np.float64),
    'f32_56x28_k8': (56, 28, 8, 2, np.float32),         # 7-point butterflies in both directions
    'f64_49x42_k4': (49, 42, 4, 1, np.float64),
}


def run(D, S, optd, fused, slab=None, generic=False):
    from sporco_amd.admm import cbpdn
    if not fused:
        os.environ['SPORCO_AMD_NO_COLS_SM'] = '1'
    if slab:
        os.environ['SPORCO_AMD_COLS_SM_FORCE_SLAB'] = str(slab)
    if generic:      # (sizes the mixed-radix register kernels serve since round 6)
        os.environ['SPORCO_AMD_UNFUSED'] = '1'
    try:
        b = cbpdn.ConvBPDN(D, S, 0.05, cbpdn.ConvBPDN.Options(optd))
        b._dev.profile(True)
        b.solve()
        prof = b._dev.profile_read()
    finally:
        os.environ.pop('SPORCO_AMD_NO_COLS_SM', None)
        os.environ.pop('SPORCO_AMD_COLS_SM_FORCE_SLAB', None)
        os.environ.pop('SPORCO_AMD_UNFUSED', None)
    return b, prof


@pytest.mark.parametrize('name', sorted(CASES))
def test_fused_column_pass_equals_the_three_kernels(backend, name):
    H, W, K, N, dt = CASES[name]
    if backend == 'hostsim' and name == 'f32_96x80_k32':
        pytest.skip("kept short on the CPU simulator")
    rng = np.random.RandomState(len(name))
    D = rng.randn(5, 5, K).astype(dt)
    S = rng.randn(H, W, N).astype(dt)
    optd = {'MaxMainIter': 8, 'RelStopTol': 0.0, 'DataType': dt}
    a, pa = run(D, S, optd, False)
    b, pb = run(D, S, optd, True)
    assert pa['fft_c2c_cols_fwd'][1] == 8 and pb['fft_c2c_cols_fwd'][1] == 0
    assert pb['fft_c2c_cols_inv'][1] == 0 and pb['sm_solve'][1] == 8
    tol = 1e-11 if dt == np.float64 else 2e-5
    for v in ('Y', 'U', 'X'):
        assert rel_l2(getattr(a, v), getattr(b, v)) < tol, v
    # a reader of Xf gets the spectrum of X, not the half-transformed buffer
    assert rel_l2(np.asarray(b.Xf), np.fft.rfft2(np.asarray(b.X, np.float64), axes=(0, 1))) < 10 * tol
    ia, ib = a.getitstat(), b.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(np.asarray(getattr(ia, f)), np.asarray(getattr(ib, f))) < tol, f


def test_fused_column_pass_against_the_oracle(backend):
    from oracle import cbpdn_oracle as orc
    H, W, K, N = 40, 48, 8, 2
    rng = np.random.RandomState(7)
    D = rng.randn(4, 4, K)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, W, N)
    b, prof = run(D, S, {'MaxMainIter': 10, 'RelStopTol': 0.0}, True)
    assert prof['fft_c2c_cols_fwd'][1] == 0
    ref = orc.admm_cbpdn(D.reshape(4, 4, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05, dtype=np.float64,
                         maxiter=10, rel_tol=0.0)
    assert rel_l2(b.Y, ref['Y']) < 1e-10 and rel_l2(b.U, ref['U']) < 1e-10
    st = b.getitstat()
    for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(np.asarray(getattr(st, f)), ref[f]) < 1e-10, f


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,K,N,dt', [(384, 320, 32, 2, np.float32), (240, 480, 64, 2, np.float32),
                                        (256, 192, 32, 2, np.float64), (360, 300, 16, 2, np.float32),
                                        (320, 240, 48, 2, np.float32)])
def test_generic_chain_at_mid_sizes_against_the_oracle(gpu_backend, H, W, K, N, dt):
    """The whole generic chain as it runs outside the register kernels (single-array state, fused
    column pass with radices 8, 4, 2, 3, 5, 64-byte tiles for the mid-sized float32 lines) against
    the float64 oracle at sizes of a few hundred points."""
    from oracle import cbpdn_oracle as orc
    from sporco_amd import _lib
    rng = np.random.RandomState(H + K)
    D = rng.randn(8, 8, K)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, W, N)
    b, prof = run(D.astype(dt), S.astype(dt), {'MaxMainIter': 10, 'RelStopTol': 0.0, 'DataType': dt}, True,
                  generic=True)
    assert not b._dev.uses_fused_rows()
    assert prof['fft_c2c_cols_fwd'][1] == 0 and prof['sm_solve'][1] == 10
    ref = orc.admm_cbpdn(D.reshape(8, 8, 1, 1, K), S.reshape(H, W, 1, N, 1), 0.05, dtype=np.float64,
                         maxiter=10, rel_tol=0.0)
    tol = 1e-10 if dt == np.float64 else 1e-4
    assert rel_l2(b.Y, ref['Y']) < tol and rel_l2(b.U, ref['U']) < tol and rel_l2(b.X, ref['X']) < tol
    st = b.getitstat()
    for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(np.asarray(getattr(st, f)), ref[f]) < 10 * tol, f


@pytest.mark.parametrize('H,W,K,slab,dt', [(48, 40, 16, 8, np.float32), (30, 36, 12, 8, np.float64),
                                           (63, 24, 20, 4, np.float64), (32, 32, 64, 32, np.float32)])
def test_slab_form_of_the_fused_column_pass(backend, H, W, K, slab, dt):
    """Tiles beyond LDS go through in slabs of filters (cols_sm_slab_kernel: forward transform and
    the slab's share of the inner product, then solve and inverse transform per slab); forced here
    at small sizes (SPORCO_AMD_COLS_SM_FORCE_SLAB), incl. a last slab that is not full."""
    rng = np.random.RandomState(H + K)
    D = rng.randn(5, 5, K).astype(dt)
    S = rng.randn(H, W, 2).astype(dt)
    optd = {'MaxMainIter': 6, 'RelStopTol': 0.0, 'DataType': dt}
    a, pa = run(D, S, optd, False)
    b, pb = run(D, S, optd, True, slab=slab)
    assert pa['fft_c2c_cols_fwd'][1] == 6 and pb['fft_c2c_cols_fwd'][1] == 0 and pb['sm_solve'][1] == 6
    tol = 1e-11 if dt == np.float64 else 2e-5
    for v in ('Y', 'U', 'X'):
        assert rel_l2(getattr(a, v), getattr(b, v)) < tol, v
    ia, ib = a.getitstat(), b.getitstat()
    for f in ('ObjFun', 'DFid', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(np.asarray(getattr(ia, f)), np.asarray(getattr(ib, f))) < tol, f


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,K,dt', [(384, 320, 64, np.float32), (256, 192, 64, np.float64)])
def test_slab_form_at_the_sizes_it_is_for(gpu_backend, H, W, K, dt):
    from oracle import cbpdn_oracle as orc
    rng = np.random.RandomState(H)
    D = rng.randn(8, 8, K)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    S = rng.randn(H, W, 2)
    b, prof = run(D.astype(dt), S.astype(dt), {'MaxMainIter': 8, 'RelStopTol': 0.0, 'DataType': dt}, True,
                  generic=True)
    assert prof['fft_c2c_cols_fwd'][1] == 0 and prof['sm_solve'][1] == 8
    ref = orc.admm_cbpdn(D.reshape(8, 8, 1, 1, K), S.reshape(H, W, 1, 2, 1), 0.05, dtype=np.float64,
                         maxiter=8, rel_tol=0.0)
    tol = 1e-10 if dt == np.float64 else 1e-4
    assert rel_l2(b.Y, ref['Y']) < tol and rel_l2(b.U, ref['U']) < tol
    for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(np.asarray(getattr(b.getitstat(), f)), ref[f]) < 10 * tol, f
