"""sporco_amd.signal / sporco_amd.fft.fftconv against outputs of the reference
(oracle/make_golden.py gen_signal): the pre/post-processing around the solvers."""

import numpy as np

from conftest import load_golden, rel_l2


def test_tikhonov_filter(backend):
    from sporco_amd import signal
    g = load_golden('signal_prims')
    sl, sh = signal.tikhonov_filter(g['s2'], 5.0, 16)
    assert sl.dtype == g['sl2'].dtype and rel_l2(sl, g['sl2']) < 1e-12
    assert rel_l2(sh, g['sh2']) < 1e-12
    sl, sh = signal.tikhonov_filter(g['s3'], 2.0, 4)
    assert sl.dtype == np.float32 and sl.shape == g['sl3'].shape
    assert rel_l2(sl, g['sl3']) < 1e-5 and rel_l2(sh, g['sh3']) < 1e-5
    assert rel_l2(sl + sh, g['s3']) < 1e-6


def test_fftconv_and_gradient_filters(backend):
    from sporco_amd import fft, signal
    g = load_golden('signal_prims')
    assert rel_l2(fft.fftconv(g['d'], g['x'], axes=(0, 1), origin=(2, 2)), g['cv']) < 1e-12
    assert rel_l2(fft.fftconv(g['k3'], g['s2']), g['cv1']) < 1e-12
    Gf, GHGf = signal.gradient_filters(5, (0, 1), (12, 9), dtype=np.float64)
    assert Gf.shape == g['Gf'].shape and rel_l2(Gf, g['Gf']) < 1e-12
    assert rel_l2(GHGf, g['GHGf']) < 1e-12
