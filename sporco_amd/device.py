"""Device-resident arrays: what lets a whole pipeline -- highpass filtering, sparse coding,
reconstruction -- stay in HBM between its one upload and its one download (SURVEY.md 8(f)
rank 4; the reference's examples/scripts/csc/cbpdn_gry.py pipeline).

A :class:`DeviceArray` is a contiguous real array in device memory with a NumPy-like shape and
dtype.  ``DeviceArray.from_host(a)`` uploads, ``x.get()`` (or ``numpy.asarray(x)``) downloads;
:func:`sporco_amd.signal.tikhonov_filter`, :func:`sporco_amd.fft.fftconv` and the solver
constructors accept and return them; ``x + y`` / ``x - y`` run on the device.
"""

import ctypes

import numpy as np

from . import _lib

__all__ = ['DeviceArray', 'is_device_array']


class DeviceArray(object):
    """Contiguous float32 / float64 array in device memory (C order)."""

    def __init__(self, shape, dtype, ptr=None, base=None):
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        if self.dtype not in (np.float32, np.float64):
            raise TypeError("device arrays are float32 or float64")
        self._base = base            # keeps the owner of a view (or a solver handle) alive
        self._own = ptr is None
        if ptr is None:
            p = ctypes.c_void_p()
            _lib.check(_lib.lib().sporco_amd_dev_malloc(max(self.nbytes, 1), ctypes.byref(p)))
            ptr = p.value
        self.ptr = int(ptr)

    # -- NumPy-like metadata ------------------------------------------------------------------
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        shape = tuple(int(s) for s in shape)
        if -1 in shape:
            known = int(np.prod([s for s in shape if s != -1]))
            shape = tuple(self.size // known if s == -1 else s for s in shape)
        if int(np.prod(shape)) != self.size:
            raise ValueError("cannot reshape device array of size %d into %s" % (self.size, shape))
        return DeviceArray(shape, self.dtype, ptr=self.ptr, base=self)

    def squeeze(self):
        return self.reshape(tuple(s for s in self.shape if s != 1))

    # -- transfers ------------------------------------------------------------------------------
    @classmethod
    def from_host(cls, a):
        a = np.asarray(a)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float64)
        a = np.ascontiguousarray(a)
        out = cls(a.shape, a.dtype)
        _lib.check(_lib.lib().sporco_amd_dev_upload(ctypes.c_void_p(out.ptr), _lib._ptr(a),
                                                    a.nbytes))
        return out

    def get(self):
        out = np.empty(self.shape, dtype=self.dtype)
        _lib.check(_lib.lib().sporco_amd_dev_download(_lib._ptr(out), ctypes.c_void_p(self.ptr),
                                                      self.nbytes))
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.get()
        return a if dtype is None else a.astype(dtype)

    # -- arithmetic on the device ------------------------------------------------------------------
    def _axpby(self, a, b, other):
        if not isinstance(other, DeviceArray) or other.shape != self.shape or \
                other.dtype != self.dtype:
            raise TypeError("device arithmetic needs a DeviceArray of the same shape and dtype")
        out = DeviceArray(self.shape, self.dtype)
        _lib.check(_lib.lib().sporco_amd_dev_axpby(_lib.dtype_code(self.dtype), self.size, a,
                                                   ctypes.c_void_p(self.ptr), b,
                                                   ctypes.c_void_p(other.ptr),
                                                   ctypes.c_void_p(out.ptr)))
        return out

    def __add__(self, other):
        return self._axpby(1.0, 1.0, other)

    def __sub__(self, other):
        return self._axpby(1.0, -1.0, other)

    def __mul__(self, scalar):
        out = DeviceArray(self.shape, self.dtype)
        _lib.check(_lib.lib().sporco_amd_dev_axpby(_lib.dtype_code(self.dtype), self.size,
                                                   float(scalar), ctypes.c_void_p(self.ptr), 0.0,
                                                   None, ctypes.c_void_p(out.ptr)))
        return out

    __rmul__ = __mul__

    def __del__(self):
        if getattr(self, '_own', False) and getattr(self, 'ptr', 0):
            try:
                _lib.lib().sporco_amd_dev_free(ctypes.c_void_p(self.ptr))
            except Exception:      # interpreter shutdown
                pass
            self.ptr = 0

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s)" % (self.shape, self.dtype.name)


def is_device_array(a):
    return isinstance(a, DeviceArray)
