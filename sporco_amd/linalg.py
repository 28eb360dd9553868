"""Device linear-algebra primitives with the signatures of ``sporco.linalg``.

``inner`` (sporco/linalg.py:41-88), ``solvedbi_sm`` / ``solvedbi_sm_c``
(:232-297) and ``rrs`` (:883-910), for 5-D arrays in the internal layout with
the sum / solve taken along the last (filter) axis.
"""

import numpy as np

from . import _lib
from .fft import real_dtype


def _split(ah, b):
    """Check the (pixels..., 1, 1, K) x (pixels..., C, N, K) broadcast pattern."""
    ah = np.asarray(ah)
    b = np.asarray(b)
    if ah.ndim != b.ndim or ah.ndim < 3:
        raise ValueError("operands must have the same number (>= 3) of dimensions")
    K = b.shape[-1]
    nd = b.ndim
    if ah.shape[-1] != K or ah.shape[-2] != 1 or ah.shape[-3] != 1 or \
            ah.shape[:nd - 3] != b.shape[:nd - 3]:
        raise NotImplementedError(
            "sporco_amd.linalg handles a of shape (..., 1, 1, K) against b of shape "
            "(..., C, N, K)")
    npix = int(np.prod(b.shape[:nd - 3])) if nd > 3 else 1
    CN = b.shape[-3] * b.shape[-2]
    return npix, CN, K


def _cplx(a, cdt):
    return np.ascontiguousarray(a, dtype=cdt)


def inner(x, y, axis=-1):
    """sum(x * y, axis, keepdims=True) on device (no conjugation)."""
    x = np.asarray(x)
    y = np.asarray(y)
    if axis not in (-1, y.ndim - 1):
        raise NotImplementedError("sporco_amd.linalg.inner sums over the last axis")
    npix, CN, K = _split(x, y)
    cdt = np.result_type(x.dtype, y.dtype, np.complex64)
    out = np.empty(y.shape[:-1] + (1,), dtype=cdt)
    _lib.check(_lib.lib().sporco_amd_inner(_lib.dtype_code(real_dtype(cdt)), npix, CN, K,
                                           _lib._ptr(_cplx(x, cdt)), _lib._ptr(_cplx(y, cdt)),
                                           _lib._ptr(out)))
    return out


def solvedbi_sm_c(ah, a, rho, axis=4):
    """c = ah / (<ah, a> + rho): kept for API parity; the device solve does not
    need it (the denominator is formed inside the kernel)."""
    ah = np.asarray(ah)
    return ah / (np.sum(ah * a, axis=axis, keepdims=True) + rho)


def solvedbi_sm(ah, rho, b, c=None, axis=4):
    """Solve (rho I + a a^H) x = b along the filter axis by Sherman-Morrison.

    ``c`` is accepted for signature compatibility and ignored.
    """
    b = np.asarray(b)
    if axis not in (-1, b.ndim - 1):
        raise NotImplementedError("sporco_amd.linalg.solvedbi_sm solves along the last axis")
    npix, CN, K = _split(ah, b)
    cdt = np.result_type(np.asarray(ah).dtype, b.dtype, np.complex64)
    x = np.empty(b.shape, dtype=cdt)
    _lib.check(_lib.lib().sporco_amd_solvedbi_sm(
        _lib.dtype_code(real_dtype(cdt)), npix, CN, K, _lib._ptr(_cplx(ah, cdt)), float(rho),
        _lib._ptr(_cplx(b, cdt)), _lib._ptr(x)))
    return x


def rrs(ax, b):
    """Relative residual ||b - ax|| / max(||ax||, ||b||) (host scalars)."""
    ax = np.asarray(ax)
    b = np.asarray(b)
    nrm = max(np.linalg.norm(ax.ravel()), np.linalg.norm(b.ravel()))
    if nrm == 0.0:
        return 0.0
    return np.linalg.norm((ax - b).ravel()) / nrm
