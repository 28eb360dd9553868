"""Device FFT primitives with the call signatures of ``sporco.fft``.

Mirrors ``rfftn`` / ``irfftn`` / ``rfl2norm2`` of the reference
(sporco/fft.py:257-314, :449-484).  The case the convolutional solvers use -- transform axes
(0, 1) of an array whose remaining axes form the batch -- goes to the device as it is; any other
pair of axes, or a single axis, after those axes have been moved to the front.
"""

import ctypes

import numpy as np

from . import _lib


def complex_dtype(dtype):
    """Complex dtype matching a real one (sporco/fft.py:44-71)."""
    return np.dtype(np.complex64 if np.dtype(dtype) in (np.float32, np.complex64)
                    else np.complex128)


def real_dtype(dtype):
    """Real dtype matching a complex one (sporco/fft.py:74-101)."""
    return np.dtype(np.float32 if np.dtype(dtype) in (np.float32, np.complex64)
                    else np.float64)


def _check_axes(axes):
    if axes is not None and tuple(axes) != (0, 1):
        raise NotImplementedError("sporco_amd.fft transforms axes (0, 1) only")


def _norm_axes(axes, nd):
    """One or two distinct axes, non-negative; None when they are (0, 1) already."""
    if axes is None:
        axes = (0, 1)
    if np.ndim(axes) == 0:
        axes = (int(axes),)
    ax = tuple(int(x) + nd if int(x) < 0 else int(x) for x in axes)
    if len(ax) not in (1, 2) or len(set(ax)) != len(ax) or not all(0 <= x < nd for x in ax):
        raise NotImplementedError("sporco_amd.fft transforms one or two distinct axes (got %r for "
                                  "%d dimensions)" % (axes, nd))
    return None if ax == (0, 1) else ax


def rfftn(a, s=None, axes=(0, 1)):
    """Unnormalised real FFT over ``axes`` (one or two axes; the last one listed is the halved
    one, as in numpy.fft.rfftn); ``s`` zero-pads (or crops) first."""
    a = np.asarray(a)
    ax = _norm_axes(axes, a.ndim)
    if ax is not None:
        # the transform axes to the front (a single axis behind a unit one), transform, and back
        if len(ax) == 1:
            m = np.moveaxis(a, ax[0], 0)[np.newaxis]
            out = rfftn(m, None if s is None else (1, int(s[0])))
            return np.moveaxis(out[0], 0, ax[0])
        out = rfftn(np.moveaxis(a, ax, (0, 1)), s)
        return np.moveaxis(out, (0, 1), ax)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    if s is not None and tuple(s) != a.shape[:2]:
        pad = np.zeros(tuple(s) + a.shape[2:], dtype=a.dtype)
        sl = tuple(slice(0, min(x, y)) for x, y in zip(s, a.shape[:2]))
        pad[sl] = a[sl]
        a = pad
    a = np.ascontiguousarray(a)
    H, W = a.shape[0], a.shape[1]
    P = int(np.prod(a.shape[2:])) if a.ndim > 2 else 1
    out = np.empty((H, W // 2 + 1) + a.shape[2:], dtype=complex_dtype(a.dtype))
    _lib.check(_lib.lib().sporco_amd_rfftn2(_lib.dtype_code(a.dtype), H, W, P,
                                            _lib._ptr(a), _lib._ptr(out)))
    return out


def irfftn(a, s, axes=(0, 1)):
    """Inverse of :func:`rfftn`; ``s`` (the lengths of the transformed axes) is required: the
    last one may be odd."""
    a = np.asarray(a)
    ax = _norm_axes(axes, a.ndim)
    if ax is not None:
        if len(ax) == 1:
            out = irfftn(np.moveaxis(a, ax[0], 0)[np.newaxis], (1, int(s[0])))
            return np.moveaxis(out[0], 0, ax[0])
        return np.moveaxis(irfftn(np.moveaxis(a, ax, (0, 1)), s), (0, 1), ax)
    a = np.ascontiguousarray(a)
    if a.dtype not in (np.complex64, np.complex128):
        a = a.astype(np.complex128)
    H, W = int(s[0]), int(s[1])
    if a.shape[0] != H or a.shape[1] != W // 2 + 1:
        raise ValueError("spectrum shape %s does not match s=%s" % (a.shape, (H, W)))
    P = int(np.prod(a.shape[2:])) if a.ndim > 2 else 1
    out = np.empty((H, W) + a.shape[2:], dtype=real_dtype(a.dtype))
    _lib.check(_lib.lib().sporco_amd_irfftn2(_lib.dtype_code(out.dtype), H, W, P,
                                             _lib._ptr(a), _lib._ptr(out)))
    return out


def rfl2norm2(xf, xs, axis=(0, 1)):
    """Squared l2 norm of the array whose :func:`rfftn` is ``xf``."""
    _check_axes(axis)
    xf = np.ascontiguousarray(xf)
    if xf.dtype not in (np.complex64, np.complex128):
        xf = xf.astype(np.complex128)
    H, W = int(xs[0]), int(xs[1])
    P = int(np.prod(xf.shape[2:])) if xf.ndim > 2 else 1
    out = ctypes.c_double(0.0)
    _lib.check(_lib.lib().sporco_amd_rfl2norm2(_lib.dtype_code(real_dtype(xf.dtype)), H, W,
                                               P, _lib._ptr(xf), ctypes.byref(out)))
    return out.value


def fftconv(a, b, axes=(0, 1), origin=None):
    """Circular convolution of real arrays over axes (0, 1) by multiplication in the DFT
    domain, ``origin`` shifting the result (sporco/fft.py:376-417), on the device
    (``sporco_amd_fftconv_dev``).  The remaining axes of ``a`` and ``b`` broadcast against each
    other.  NumPy arrays: one upload per operand, one download; device arrays
    (:class:`sporco_amd.device.DeviceArray`) in, device array out."""
    from .device import DeviceArray
    _check_axes(axes)
    dev = isinstance(a, DeviceArray) or isinstance(b, DeviceArray)
    ops = []
    for x in (a, b):
        if not isinstance(x, DeviceArray):
            x = np.asarray(x)
            if np.iscomplexobj(x):
                raise NotImplementedError("sporco_amd.fft handles real-valued arrays")
        ops.append(x)
    a, b = ops
    dt = np.result_type(a.dtype, b.dtype, np.float32)
    nd = max(a.ndim, b.ndim)
    ash = tuple(a.shape) + (1,) * (nd - a.ndim)
    bsh = tuple(b.shape) + (1,) * (nd - b.ndim)
    osh = tuple(max(x, y) for x, y in zip(ash, bsh))
    for x, y, o in zip(ash[2:], bsh[2:], osh[2:]):
        if x not in (1, o) or y not in (1, o):
            raise ValueError("operands of shapes %s and %s do not broadcast" % (ash, bsh))
    # merge the trailing axes into at most three groups on which both operands behave alike
    groups = []
    for x, y in zip(ash[2:], bsh[2:]):
        kind = (x == 1, y == 1)
        if x == 1 and y == 1:
            continue
        if groups and groups[-1][2] == kind:
            groups[-1][0] *= x
            groups[-1][1] *= y
        else:
            groups.append([x, y, kind])
    if len(groups) > 3:
        raise NotImplementedError("fftconv: more than three broadcast groups of trailing axes")
    while len(groups) < 3:
        groups.insert(0, [1, 1, None])
    i64 = ctypes.c_int64
    da = (i64 * 3)(*[g[0] for g in groups])
    db = (i64 * 3)(*[g[1] for g in groups])

    def to_dev(x):
        if isinstance(x, DeviceArray):
            if x.dtype != dt:
                raise TypeError("device operands must share one dtype")
            return x
        return DeviceArray.from_host(np.ascontiguousarray(x, dtype=dt))
    ad, bd = to_dev(a), to_dev(b)
    out = DeviceArray(osh, dt)
    og = (0, 0) if origin is None else tuple(int(v) for v in origin)
    vp = ctypes.c_void_p
    _lib.check(_lib.lib().sporco_amd_fftconv_dev(
        _lib.dtype_code(dt), ash[0], ash[1], da, vp(ad.ptr), bsh[0], bsh[1], db, vp(bd.ptr),
        og[0], og[1], vp(out.ptr)))
    return out if dev else out.get()
