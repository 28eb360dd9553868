"""Device-resident pre / post-processing: signal.tikhonov_filter and
fft.fftconv as device pipelines, DeviceArray hand-over into the solvers, and the transfer
accounting that shows the cbpdn_gry pipeline (examples/scripts/csc/cbpdn_gry.py:45-77 of the
reference: highpass filter -> ConvBPDN -> reconstruct -> add the lowpass part) makes one upload
of the image and one download of the result.

Reference outputs: tests/golden/signal_prims.npz (oracle/make_golden.py gen_signal: sporco.signal /
sporco.fft of the unmodified reference).  Tolerances: float64 1e-12, float32 1e-5."""

import numpy as np
import pytest

from conftest import load_golden, rel_l2


def test_tikhonov_filter_device_matches_reference(backend):
    from sporco_amd import signal
    from sporco_amd.device import DeviceArray
    from sporco_amd import _lib
    g = load_golden('signal_prims')
    for s, lm, npd, rl, rh in ((g['s2'], 5.0, 16, g['sl2'], g['sh2']),
                               (g['s3'], 2.0, 4, g['sl3'], g['sh3'])):
        tol = 1e-5 if s.dtype == np.float32 else 1e-12
        _lib.transfer_stats(reset=True)
        sl, sh = signal.tikhonov_filter(s, lm, npd)
        st = _lib.transfer_stats()
        assert sl.dtype == s.dtype and sl.shape == s.shape
        assert rel_l2(sl, rl) < tol and rel_l2(sh, rh) < tol
        # host arrays in and out: one upload of s, one download (both results together)
        assert st['h2d_calls'] == 1 and st['d2h_calls'] == 1
        assert st['h2d_bytes'] == s.nbytes and st['d2h_bytes'] == 2 * s.nbytes
        # device arrays in and out: nothing crosses
        sd = DeviceArray.from_host(s)
        _lib.transfer_stats(reset=True)
        dl, dh = signal.tikhonov_filter(sd, lm, npd)
        st = _lib.transfer_stats()
        assert st['h2d_bytes'] == 0 and st['d2h_bytes'] == 0
        assert rel_l2(dl.get(), rl) < tol and rel_l2(dh.get(), rh) < tol
        assert rel_l2((dl + dh).get(), s) < 10 * tol


def test_fftconv_device_matches_reference(backend):
    from sporco_amd import fft
    from sporco_amd.device import DeviceArray
    g = load_golden('signal_prims')
    for a, b, ref, origin in ((g['d'], g['x'], g['cv'], (2, 2)), (g['k3'], g['s2'], g['cv1'], None)):
        out = fft.fftconv(a, b, origin=origin)
        assert out.shape == ref.shape and rel_l2(out, ref) < 1e-12
        outd = fft.fftconv(DeviceArray.from_host(a), DeviceArray.from_host(b), origin=origin)
        assert isinstance(outd, DeviceArray) and rel_l2(outd.get(), ref) < 1e-12
    # broadcasting of the trailing axes: (h, w, 1, M) against (H, W, N, 1)
    rng = np.random.RandomState(2)
    a4, b4 = rng.randn(3, 3, 1, 4), rng.randn(10, 9, 2, 1)
    ref4 = np.fft.irfftn(np.fft.rfftn(a4, (10, 9), axes=(0, 1)) * np.fft.rfftn(b4, axes=(0, 1)),
                         (10, 9), axes=(0, 1))
    assert rel_l2(fft.fftconv(a4, b4), ref4) < 1e-12
    # the sparse-synthesis recipe of tests/admm/test_cbpdn.py:160-165: S = sum_m d_m * x_m
    rng = np.random.RandomState(1)
    D = rng.randn(8, 8, 4)
    X0 = rng.randn(32, 32, 4) * (rng.rand(32, 32, 4) > 0.9)
    S = np.sum(fft.fftconv(D, X0), axis=2)
    Sref = np.sum(np.fft.irfftn(np.fft.rfftn(D, (32, 32), axes=(0, 1)) *
                                np.fft.rfftn(X0, axes=(0, 1)), (32, 32), axes=(0, 1)), axis=2)
    assert rel_l2(S, Sref) < 1e-12


@pytest.mark.parametrize('H', [64, pytest.param(512, marks=pytest.mark.gpu)])
def test_cbpdn_gry_pipeline_one_upload_one_download(backend, H):
    """img -> tikhonov_filter -> ConvBPDN(D, sh) -> solve -> reconstruct -> sl + shr -> host:
    the image goes up once, the reconstructed image comes down once; the dictionary (a few
    KB) is the only other upload, the per-iteration scalars the only other downloads."""
    from sporco_amd import signal, _lib
    from sporco_amd.admm import cbpdn
    from sporco_amd.device import DeviceArray
    rng = np.random.RandomState(5)
    K = 8 if H == 64 else 64
    img = rng.rand(H, H).astype(np.float32)
    D = rng.randn(8, 8, K).astype(np.float32)
    D /= np.sqrt(np.sum(D ** 2, axis=(0, 1), keepdims=True))
    opt = cbpdn.ConvBPDN.Options({'MaxMainIter': 10, 'RelStopTol': 0.0})
    # host reference of the same pipeline
    sl0, sh0 = signal.tikhonov_filter(img, 5.0, 16)
    b0 = cbpdn.ConvBPDN(D, sh0, 0.05, opt, dimK=0)
    b0.solve()
    imgr0 = sl0 + b0.reconstruct().squeeze()
    # device-resident pipeline
    _lib.transfer_stats(reset=True)
    imgd = DeviceArray.from_host(img)
    sl, sh = signal.tikhonov_filter(imgd, 5.0, 16)
    b = cbpdn.ConvBPDN(D, sh, 0.05, opt, dimK=0, resident=True)
    X = b.solve()
    assert isinstance(X, DeviceArray) and X.shape == b.cri.shpX
    shr = b.reconstruct(device=True)
    imgr = (sl + shr.reshape(sl.shape)).get()
    st = _lib.transfer_stats()
    assert rel_l2(imgr, imgr0) < 1e-5
    small = 64 * 1024
    assert img.nbytes <= st['h2d_bytes'] <= img.nbytes + D.nbytes + small
    assert img.nbytes <= st['d2h_bytes'] <= img.nbytes + small
    # the coefficient maps are still there when asked for
    assert rel_l2(X.get(), b0.Y) < 1e-5
    assert rel_l2(b.S.squeeze(), sh0) < 1e-6
