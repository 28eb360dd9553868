"""The generic ADMM iteration (any size, float32 / float64) with its epilogue fused into the
half-spectrum -> real row pass of irfftn (fft.h fft_c2r_post: relax_AX + ystep + ustep + sums on
every output element, sporco/admm/admm.py:877-885, cbpdn.py:614-620, admm.py:434-486;
SPORCO_AMD_C2R_POST=1 -- an opt-in: it measures no faster, profiles/r03q_generic_chain.md)
against the default, the two kernels apart: same element function (csc_post_elem.h); the sums
are reduced over other workgroups, so the statistics -- and through the rho schedule the
iterates -- agree to rounding."""

import os

import numpy as np
import pytest

from conftest import rel_l2


def run(cls_name, D, S, args, optd, fused, wrap=None):
    from sporco_amd.admm import cbpdn
    cls = getattr(cbpdn, cls_name)
    old = os.environ.get('SPORCO_AMD_C2R_POST')
    os.environ['SPORCO_AMD_C2R_POST'] = '1' if fused else '0'
    try:
        if wrap is not None:
            b = cbpdn.AddMaskSim(cls, D, S, wrap, *args, opt=cls.Options(optd))
            inner = b.cbpdn
        else:
            b = inner = cls(D, S, *args, cls.Options(optd))
        inner._dev.profile(True)
        b.solve()
        prof = inner._dev.profile_read()
    finally:
        if old is None:
            os.environ.pop('SPORCO_AMD_C2R_POST', None)
        else:
            os.environ['SPORCO_AMD_C2R_POST'] = old
    return inner, prof


CASES = {
    # (class, H, W, K, N, dtype, options, mask)
    'f64_odd_size': ('ConvBPDN', 15, 18, 5, 2, np.float64, {'MaxMainIter': 8}, False),
    'f64_odd_columns': ('ConvBPDN', 16, 16, 3, 1, np.float64, {'MaxMainIter': 6, 'NonNegCoef': True}, False),
    'f32_size_24': ('ConvBPDN', 24, 24, 4, 2, np.float32,
                    {'MaxMainIter': 8, 'NoBndryCross': True, 'AuxVarObj': True}, False),
    'f64_weights': ('ConvBPDN', 16, 20, 4, 2, np.float64, {'MaxMainIter': 6, 'L1Weight': 'array'}, False),
    'f64_gradreg': ('ConvBPDNGradReg', 16, 16, 4, 2, np.float64, {'MaxMainIter': 6}, False),
    'f64_ams': ('ConvBPDN', 18, 16, 4, 2, np.float64, {'MaxMainIter': 6}, True),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_fused_epilogue_equals_two_kernels(backend, name):
    cls_name, H, W, K, N, dt, optd, mask = CASES[name]
    rng = np.random.RandomState(len(name))
    D = rng.randn(5, 5, K).astype(dt)
    S = rng.randn(H, W, N).astype(dt)
    optd = dict(optd)
    if optd.get('L1Weight') == 'array':
        optd['L1Weight'] = (0.5 + rng.rand(H, W, 1, N, K)).astype(dt)
    args = (0.1, 0.05) if cls_name == 'ConvBPDNGradReg' else (0.1,)
    wm = (rng.rand(H, W, N) > 0.3).astype(dt) if mask else None
    a, pa = run(cls_name, D, S, args, optd, False, wm)
    b, pb = run(cls_name, D, S, args, optd, True, wm)
    assert pa['admm_post'][1] == optd['MaxMainIter'] and pb['admm_post'][1] == 0
    assert pa['fft_c2r_rows'][1] >= pb['fft_c2r_rows'][1] == optd['MaxMainIter']
    tol = 1e-12 if dt == np.float64 else 2e-6
    for v in ('Y', 'U', 'X'):
        assert rel_l2(getattr(a, v), getattr(b, v)) < tol, v
    ia, ib = a.getitstat(), b.getitstat()
    for f in ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(np.asarray(getattr(ia, f)), np.asarray(getattr(ib, f))) < tol, f


def test_what_keeps_the_two_kernels(backend):
    """ConvBPDNJoint (its epilogue couples the channels) and LinSolveCheck (its residual is
    evaluated from X) keep the separate epilogue kernel."""
    rng = np.random.RandomState(1)
    D = rng.randn(5, 5, 4)
    S = rng.randn(16, 18, 3, 2)
    b, prof = run('ConvBPDNJoint', D, S, (0.1, 0.05), {'MaxMainIter': 4}, True)
    assert prof['admm_post'][1] == 4
    S1 = rng.randn(16, 18, 2)
    b, prof = run('ConvBPDN', D, S1, (0.1,), {'MaxMainIter': 4, 'LinSolveCheck': True}, True)
    assert prof['admm_post'][1] == 4 and max(b.getitstat().XSlvRelRes) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize('H,W,K,N,dt', [(384, 384, 16, 2, np.float32), (200, 300, 8, 1, np.float64)])
def test_fused_epilogue_against_oracle(gpu_backend, H, W, K, N, dt):
    """Sizes the register kernels do not serve (a 2^a 3 length, a 2^a 3 5^b one, float64) against
    the float64 oracle."""
    from oracle import cbpdn_oracle as orc
    rng = np.random.RandomState(H + W)
    D = rng.randn(8, 8, K).astype(dt)
    S = rng.randn(H, W, N).astype(dt)
    b, prof = run('ConvBPDN', D, S, (0.1,), {'MaxMainIter': 6}, True)
    assert prof['admm_post'][1] == 0
    r = orc.admm_cbpdn(D.reshape(8, 8, 1, 1, K).astype(np.float64),
                       S.reshape(H, W, 1, N, 1).astype(np.float64), 0.1, dtype=np.float64, maxiter=6)
    tol = 1e-9 if dt == np.float64 else 1e-4
    assert rel_l2(b.Y, r['Y']) < tol
    for f in ('ObjFun', 'PrimalRsdl', 'DualRsdl', 'Rho'):
        assert rel_l2(getattr(b.getitstat(), f), r[f]) < tol, f
