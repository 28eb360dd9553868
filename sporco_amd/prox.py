"""Device proximal operators with the signatures of ``sporco.prox``.

``prox_l1`` (sporco/prox/_lp.py:144-183) and ``prox_sl1l2``
(sporco/prox/_l21.py:51-88) for real arrays and scalar parameters; inside the
solvers the same arithmetic runs fused into the ADMM / PGM epilogue kernels,
where weight arrays are supported.
"""

import numpy as np

from . import _lib


def _real(v):
    v = np.asarray(v)
    if np.iscomplexobj(v):
        raise NotImplementedError("sporco_amd.prox handles real arrays only")
    if v.dtype not in (np.float32, np.float64):
        v = v.astype(np.float64)
    return np.ascontiguousarray(v)


def prox_l1(v, alpha):
    """sign(v) * max(|v| - alpha, 0); ``alpha`` a scalar or an array that broadcasts against
    ``v`` (sporco/prox/_lp.py:144-183)."""
    v = _real(v)
    out = np.empty_like(v)
    if np.ndim(alpha) == 0:
        _lib.check(_lib.lib().sporco_amd_prox_l1(_lib.dtype_code(v.dtype), v.size, _lib._ptr(v),
                                                 float(alpha), _lib._ptr(out)))
        return out
    a = np.asarray(alpha, dtype=v.dtype)
    if a.ndim > v.ndim or np.broadcast_shapes(a.shape, v.shape) != v.shape:
        raise ValueError("alpha of shape %s does not broadcast against v of shape %s"
                         % (a.shape, v.shape))
    a = a.reshape((1,) * (v.ndim - a.ndim) + a.shape)
    if v.ndim <= 5:
        vs = [1] * (5 - v.ndim) + list(v.shape)
        as_ = [1] * (5 - v.ndim) + [na if nv > 1 else 1 for nv, na in zip(v.shape, a.shape)]
    else:           # more than five axes: expand alpha (no compact 5-D form in general)
        a = np.broadcast_to(a, v.shape)
        vs, as_ = [1, 1, 1, 1, v.size], [1, 1, 1, 1, v.size]
    i64 = _lib.ctypes.c_int64
    _lib.check(_lib.lib().sporco_amd_prox_l1w(_lib.dtype_code(v.dtype), (i64 * 5)(*vs), _lib._ptr(v),
                                              (i64 * 5)(*as_), _lib._ptr(np.ascontiguousarray(a)),
                                              _lib._ptr(out)))
    return out


def prox_sl1l2(v, alpha, beta, axis=None):
    """prox of alpha*||.||_1 + beta*||.||_2 with the l2 norm over ``axis``."""
    if np.ndim(alpha) != 0 or np.ndim(beta) != 0:
        raise NotImplementedError("array-valued parameters are handled inside the solvers only")
    v = _real(v)
    if axis is None:
        outer, C, inner = 1, v.size, 1
    else:
        if isinstance(axis, (tuple, list)):
            if len(axis) != 1:
                raise NotImplementedError("a single l2 axis is supported")
            axis = axis[0]
        axis = axis % v.ndim
        outer = int(np.prod(v.shape[:axis])) if axis > 0 else 1
        C = v.shape[axis]
        inner = int(np.prod(v.shape[axis + 1:])) if axis + 1 < v.ndim else 1
    out = np.empty_like(v)
    _lib.check(_lib.lib().sporco_amd_prox_sl1l2(_lib.dtype_code(v.dtype), outer, C, inner,
                                                _lib._ptr(v), float(alpha), float(beta),
                                                _lib._ptr(out)))
    return out


def prox_l2(v, alpha, axis=None):
    """v/||v|| * max(0, ||v|| - alpha): the alpha_1 = 0 case of :func:`prox_sl1l2`."""
    return prox_sl1l2(v, 0.0, alpha, axis)


def norm_l1(x, axis=None):
    return np.sum(np.abs(x), axis=axis, keepdims=axis is not None)
