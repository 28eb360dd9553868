"""Signal pre-processing around the convolutional solvers, with the call signatures of
``sporco.signal``: the lowpass / highpass split every CSC example applies to its images
before sparse coding (``tikhonov_filter``) and the gradient filter spectra of
``ConvBPDNGradReg``.  The transforms run on the GPU through :mod:`sporco_amd.fft`."""

import numpy as np

from . import fft as sfft

__all__ = ['gradient_filters', 'tikhonov_filter']


def gradient_filters(ndim, axes, axshp, dtype=None):
    """Frequency-domain gradient operators ``Gf`` and ``GHGf = sum_i |Gf_i|^2`` for the
    two-tap difference filters along ``axes`` (sporco/signal.py:204-240).  Real dtypes and
    axes (0, 1) (the half-spectrum layout of the solvers).

    The DFT of the filter (1, -1) along an axis of length n is 1 - exp(-2 pi i f / n); it is
    written down directly, on the half spectrum of the last transformed axis."""
    if dtype is None:
        dtype = np.float32
    dtype = np.dtype(dtype)
    if dtype.kind == 'c':
        raise NotImplementedError("sporco_amd handles real-valued signals")
    axes = tuple(axes)
    if axes != (0, 1):
        raise NotImplementedError("sporco_amd.signal handles axes (0, 1)")
    cdt = sfft.complex_dtype(dtype)
    n0, n1 = int(axshp[0]), int(axshp[1])
    shape = (n0, n1 // 2 + 1) + (1,) * (ndim - 2)
    Gf = np.zeros(shape + (2,), dtype=cdt)
    f0 = np.arange(n0).reshape((n0, 1) + (1,) * (ndim - 2))
    f1 = np.arange(n1 // 2 + 1).reshape((1, n1 // 2 + 1) + (1,) * (ndim - 2))
    # (a length-1 axis crops the filter to its first tap: the transform is then 1)
    Gf[..., 0] = (1.0 - np.exp(-2j * np.pi * f0 / n0)) if n0 > 1 else 1.0
    Gf[..., 1] = (1.0 - np.exp(-2j * np.pi * f1 / n1)) if n1 > 1 else 1.0
    GHGf = np.sum(Gf.real ** 2 + Gf.imag ** 2, axis=-1).astype(dtype)
    return Gf, GHGf


def tikhonov_filter(s, lmbda, npd=16):
    r"""Lowpass / highpass split by Tikhonov regularisation with a gradient operator:
    ``sl = argmin_x (1/2)||x - s||^2 + (lmbda/2) sum_i ||G_i x||^2``, ``sh = s - sl``
    (sporco/signal.py:244-301): symmetric padding by ``npd``, division by
    ``1 + lmbda sum_i |G_i|^2`` in the DFT domain, crop.  ``s`` is (H, W) or (H, W, ...) real."""
    s = np.asarray(s)
    if np.iscomplexobj(s):
        raise NotImplementedError("sporco_amd handles real-valued signals")
    sdt = s.dtype
    wrk = s if s.dtype in (np.float32, np.float64) else s.astype(np.float64)
    Hp, Wp = s.shape[0] + 2 * npd, s.shape[1] + 2 * npd
    # |FFT of [-1, 1]|^2 along an axis of length n is 2 - 2 cos(2 pi f / n)
    gr = 2.0 - 2.0 * np.cos(2.0 * np.pi * np.arange(Hp) / Hp)
    gc = 2.0 - 2.0 * np.cos(2.0 * np.pi * np.arange(Wp // 2 + 1) / Wp)
    A = 1.0 + lmbda * (gr[:, np.newaxis] + gc[np.newaxis, :])
    A = A.reshape(A.shape + (1,) * (s.ndim - 2)).astype(wrk.dtype)
    sp = np.pad(wrk, ((npd, npd),) * 2 + ((0, 0),) * (s.ndim - 2), 'symmetric')
    spf = sfft.rfftn(sp, None, (0, 1))
    spf /= A
    sp = sfft.irfftn(spf, (Hp, Wp), (0, 1))
    slp = sp[npd:Hp - npd, npd:Wp - npd]
    shp = wrk - slp
    return slp.astype(sdt), shp.astype(sdt)
