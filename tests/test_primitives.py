"""The exported primitives -- sporco_amd.linalg / prox / fft, i.e. the C-ABI entry points
sporco_amd_rfftn2 / irfftn2 / solvedbi_sm / inner / prox_l1 / prox_l1w / prox_sl1l2 /
rfl2norm2 (include/sporco_amd.h) -- against the reference outputs in
tests/golden/primitives.npz (written by oracle/make_golden.py from sporco.linalg, sporco.prox
and sporco.fft on the shapes of the reference's own unit tests: tests/test_linalg.py:147-159,
:298-311, tests/test_prox.py:77-95, tests/test_fft.py:15-28).

Tolerances: float64 1e-12 (transforms, solves), exact for the shrinkage operators; the float32
transform fixture 1e-5.
"""

import numpy as np
import pytest

from conftest import load_golden, rel_l2


def test_solvedbi_sm_and_inner(backend):
    from sporco_amd import linalg
    g = load_golden('primitives')
    ah, b, rho = g['sm_ah'], g['sm_b'], float(g['sm_rho'])
    x = linalg.solvedbi_sm(ah, rho, b)
    assert x.shape == g['sm_x'].shape and x.dtype == np.complex128
    assert rel_l2(x, g['sm_x']) < 1e-12
    # the defining property (tests/test_linalg.py:147-159): (rho I + a a^H) x = b
    a = np.conj(ah)
    lhs = rho * x + a * np.sum(ah * x, axis=4, keepdims=True)
    assert linalg.rrs(lhs, b) < 1e-11
    assert rel_l2(linalg.solvedbi_sm_c(ah, a, rho), g['sm_c']) < 1e-12
    # `c` is accepted and ignored (the kernel forms the denominator itself)
    assert np.array_equal(linalg.solvedbi_sm(ah, rho, b, g['sm_c']), x)
    ip = linalg.inner(ah, b, axis=4)
    assert ip.shape == g['inner_ab'].shape and rel_l2(ip, g['inner_ab']) < 1e-12
    # float32 inputs stay float32
    x32 = linalg.solvedbi_sm(ah.astype(np.complex64), rho, b.astype(np.complex64))
    assert x32.dtype == np.complex64 and rel_l2(x32, g['sm_x']) < 1e-5
    with pytest.raises(NotImplementedError):
        linalg.inner(ah, b, axis=0)


def test_rfftn_irfftn_rfl2norm2(backend):
    from sporco_amd import fft
    g = load_golden('primitives')
    af = fft.rfftn(g['fft_a'], axes=(0, 1))
    assert af.shape == g['fft_af'].shape and rel_l2(af, g['fft_af']) < 1e-12
    ar = fft.irfftn(g['fft_af'], (12, 9), axes=(0, 1))
    assert ar.shape == g['fft_ar'].shape and rel_l2(ar, g['fft_ar']) < 1e-12
    a2f = fft.rfftn(g['fft_a2'], axes=(0, 1))
    assert a2f.dtype == np.complex64 and rel_l2(a2f, g['fft_a2f']) < 1e-5
    # zero-padded transform of a filter bank (rfftn(D, Nv), cbpdn.py:247)
    assert rel_l2(fft.rfftn(g['fft_d'], (12, 9), axes=(0, 1)), g['fft_df']) < 1e-12
    # Parseval on the half spectrum, odd and even row lengths (tests/test_fft.py:15-28)
    n_odd = fft.rfl2norm2(g['fft_af'], g['fft_a'].shape, axis=(0, 1))
    assert abs(n_odd - float(g['nrm_odd'])) < 1e-12 * float(g['nrm_odd'])
    assert abs(n_odd - np.sum(g['fft_a'] ** 2)) < 1e-11 * n_odd
    n_even = fft.rfl2norm2(g['fft_a2f'], g['fft_a2'].shape, axis=(0, 1))
    assert abs(n_even - float(g['nrm_even'])) < 1e-5 * float(g['nrm_even'])


def test_prox_operators(backend):
    from sporco_amd import prox
    g = load_golden('primitives')
    v, a = g['prox_v'], float(g['prox_alpha'])
    assert np.array_equal(prox.prox_l1(v, a), g['prox_l1'])
    # array-valued threshold: one weight per filter, broadcast (tests/test_prox.py:77-81 shape)
    assert np.array_equal(prox.prox_l1(v, a * g['prox_w']), g['prox_l1w'])
    full = np.broadcast_to(a * g['prox_w'], v.shape).copy()
    assert np.array_equal(prox.prox_l1(v, full), g['prox_l1w'])
    assert np.array_equal(prox.prox_l1(v, (a * g['prox_w']).ravel()), g['prox_l1w'])
    # against the minimiser definition: prox(v) = argmin_x 0.5 (x - v)^2 + alpha |x|
    x = prox.prox_l1(v, a)
    assert np.all(np.abs(x) <= np.maximum(np.abs(v) - a, 0) + 1e-15)
    assert rel_l2(prox.prox_l2(v, 0.9, axis=2), g['prox_l2']) < 1e-14
    assert rel_l2(prox.prox_sl1l2(v, a, 0.9, axis=2), g['prox_sl1l2']) < 1e-14
    # all-zero groups stay zero (no 0/0)
    z = prox.prox_sl1l2(g['prox_vz'], a, 0.9, axis=2)
    assert np.all(np.isfinite(z)) and rel_l2(z, g['prox_sl1l2_z']) < 1e-14
    v32 = v.astype(np.float32)
    x32 = prox.prox_l1(v32, np.float32(a))
    assert x32.dtype == np.float32
    assert np.array_equal(x32, np.sign(v32) * np.maximum(np.abs(v32) - np.float32(a), 0))
    with pytest.raises(ValueError):
        prox.prox_l1(v, np.ones(7))
    with pytest.raises(NotImplementedError):
        prox.prox_l1(v.astype(np.complex128), a)


@pytest.mark.parametrize('shape', [(56, 28, 3), (49, 42, 2), (63, 35, 4), (21, 15, 1), (45, 75, 2), (96, 80, 2)])
@pytest.mark.parametrize('dt', [np.float64, np.float32])
def test_rfftn_lengths_with_factors_3_5_7(backend, shape, dt):
    """Lines whose lengths carry the factors 3, 5 and 7 (fft.hip: their butterflies carry the
    constants in the instruction stream) against numpy.fft, both directions (sporco/fft.py:257-314)."""
    from sporco_amd import fft as sf
    rng = np.random.RandomState(shape[0])
    x = rng.randn(*shape).astype(dt)
    X = sf.rfftn(x, None, (0, 1))
    tol = 1e-14 if dt == np.float64 else 2e-6
    assert rel_l2(X, np.fft.rfft2(x.astype(np.float64), axes=(0, 1))) < tol
    assert rel_l2(sf.irfftn(X, shape[:2], (0, 1)), x) < tol
