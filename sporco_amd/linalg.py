"""Device linear-algebra primitives with the signatures of ``sporco.linalg``.

``inner`` (sporco/linalg.py:41-88), ``solvedbi_sm`` / ``solvedbi_sm_c``
(:232-297) and ``rrs`` (:883-910).  Any axis and any pair of operands that broadcast against each
other, as in the reference; the layout the solvers use -- ``a`` of shape (pixels..., 1, 1, K)
against ``b`` of shape (pixels..., C, N, K), sum / solve along the last axis -- goes to the device
as it is, every other case after the axis has been moved last and the operands expanded.
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


def _fast_pattern(ah, b, axis):
    """(npix, CN, K) when the operands have the solvers' layout and ``axis`` is the last, else None."""
    ah, b = np.asarray(ah), np.asarray(b)
    if ah.ndim != b.ndim or ah.ndim < 3 or axis not in (-1, b.ndim - 1):
        return None
    try:
        return _split(ah, b)
    except (NotImplementedError, ValueError):
        return None


def _moved_last(x, y, axis):
    """The operands expanded against each other with ``axis`` moved last, as contiguous 2-D
    (M, K) arrays, and the expanded shape (axis still in place)."""
    xb, yb = np.broadcast_arrays(x, y)
    nd = xb.ndim
    ax = axis + nd if axis < 0 else axis
    if not 0 <= ax < nd:
        raise ValueError("axis %d out of range for %d dimensions" % (axis, nd))
    K = xb.shape[ax]
    xm = np.ascontiguousarray(np.moveaxis(xb, ax, -1)).reshape(-1, K)
    ym = np.ascontiguousarray(np.moveaxis(yb, ax, -1)).reshape(-1, K)
    return xm, ym, xb.shape, ax


def inner(x, y, axis=-1):
    """sum(x * y, axis, keepdims=True) on device, no conjugation (sporco/linalg.py:41-88): any
    axis, operands that broadcast against each other."""
    x = np.asarray(x)
    y = np.asarray(y)
    real = not (np.iscomplexobj(x) or np.iscomplexobj(y))
    cdt = np.result_type(x.dtype, y.dtype, np.complex64)
    pat = _fast_pattern(x, y, axis)
    if pat is not None:
        npix, CN, K = pat
        out = np.empty(y.shape[:-1] + (1,), dtype=cdt)
        _lib.check(_lib.lib().sporco_amd_inner(_lib.dtype_code(real_dtype(cdt)), npix, CN, K,
                                               _lib._ptr(_cplx(x, cdt)), _lib._ptr(_cplx(y, cdt)),
                                               _lib._ptr(out)))
    else:
        xm, ym, shp, ax = _moved_last(x, y, axis)
        M, K = xm.shape
        o = np.empty((M, 1), dtype=cdt)
        _lib.check(_lib.lib().sporco_amd_inner(_lib.dtype_code(real_dtype(cdt)), M, 1, K,
                                               _lib._ptr(_cplx(xm, cdt)), _lib._ptr(_cplx(ym, cdt)),
                                               _lib._ptr(o)))
        rest = tuple(n for i, n in enumerate(shp) if i != ax)
        out = np.moveaxis(o.reshape(rest + (1,)), -1, ax)
    return out.real.astype(np.result_type(x.dtype, y.dtype, np.float32)) if real else out


def solvedbi_sm_c(ah, a, rho, axis=4):
    """c = ah / (<ah, a> + rho) (sporco/linalg.py:232-256): the sum on the device.  The device
    solve itself does not need it (the denominator is formed inside the kernel)."""
    ah = np.asarray(ah)
    return ah / (inner(ah, a, axis=axis) + rho)


def solvedbi_sm(ah, rho, b, c=None, axis=4):
    """Solve (rho I + a a^H) x = b along the filter axis by Sherman-Morrison.

    ``c`` is accepted for signature compatibility and ignored.
    """
    b = np.asarray(b)
    ah = np.asarray(ah)
    cdt = np.result_type(ah.dtype, b.dtype, np.complex64)
    # real operands give a real solution, as the reference's expression does (and as inner() above)
    real = not (np.iscomplexobj(ah) or np.iscomplexobj(b))
    rdt = np.result_type(ah.dtype, b.dtype, np.float32)
    pat = _fast_pattern(ah, b, axis)
    if pat is not None:
        npix, CN, K = pat
        x = np.empty(b.shape, dtype=cdt)
        _lib.check(_lib.lib().sporco_amd_solvedbi_sm(
            _lib.dtype_code(real_dtype(cdt)), npix, CN, K, _lib._ptr(_cplx(ah, cdt)), float(rho),
            _lib._ptr(_cplx(b, cdt)), _lib._ptr(x)))
        return x.real.astype(rdt) if real else x
    # any other axis / broadcast pattern: one system per row after the axis has been moved last
    am, bm, shp, ax = _moved_last(ah, b, axis)
    M, K = bm.shape
    xm = np.empty((M, K), dtype=cdt)
    _lib.check(_lib.lib().sporco_amd_solvedbi_sm(
        _lib.dtype_code(real_dtype(cdt)), M, 1, K, _lib._ptr(_cplx(am, cdt)), float(rho),
        _lib._ptr(_cplx(bm, cdt)), _lib._ptr(xm)))
    rest = tuple(n for i, n in enumerate(shp) if i != ax)
    x = np.moveaxis(xm.reshape(rest + (K,)), -1, ax)
    return x.real.astype(rdt) if real else x


def rrs(ax, b):
    """Relative residual ||b - ax|| / max(||ax||, ||b||) (host scalars)."""
    ax = np.asarray(ax)
    b = np.asarray(b)
    nrm = max(np.linalg.norm(ax.ravel()), np.linalg.norm(b.ravel()))
    if nrm == 0.0:
        return 0.0
    return np.linalg.norm((ax - b).ravel()) / nrm
