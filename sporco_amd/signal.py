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
    ``1 + lmbda sum_i |G_i|^2`` in the DFT domain, crop -- all on the device
    (``sporco_amd_tikhonov_filter_dev``).  ``s`` is (H, W) or (H, W, ...) real: a NumPy array
    (one upload, one download of the two results together) or a
    :class:`sporco_amd.device.DeviceArray`, in which case the results are device arrays too and
    nothing crosses PCIe -- ``ConvBPDN(D, sh, ...)`` then takes the highpass part as it is."""
    from . import _lib
    from .device import DeviceArray
    dev_in = isinstance(s, DeviceArray)
    if not dev_in:
        s = np.asarray(s)
        if np.iscomplexobj(s):
            raise NotImplementedError("sporco_amd handles real-valued signals")
    sdt = s.dtype
    sd = s if dev_in else DeviceArray.from_host(s)
    H, W = sd.shape[0], sd.shape[1]
    P = int(np.prod(sd.shape[2:])) if sd.ndim > 2 else 1
    # the two results side by side in one allocation: one download for a host caller
    both = DeviceArray((2,) + sd.shape, sd.dtype)
    slp = DeviceArray(sd.shape, sd.dtype, ptr=both.ptr, base=both)
    shp = DeviceArray(sd.shape, sd.dtype, ptr=both.ptr + sd.nbytes, base=both)
    vp = _lib.ctypes.c_void_p
    _lib.check(_lib.lib().sporco_amd_tikhonov_filter_dev(
        _lib.dtype_code(sd.dtype), H, W, P, vp(sd.ptr), float(lmbda), int(npd), vp(slp.ptr),
        vp(shp.ptr)))
    if dev_in:
        return slp, shp
    out = both.get()
    return out[0].astype(sdt), out[1].astype(sdt)
