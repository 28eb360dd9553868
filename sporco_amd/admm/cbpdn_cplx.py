"""admm.cbpdn.ConvBPDN for complex-valued signals and dictionaries (sporco/admm/cbpdn.py:209-217:
``real_dtype`` False, ``fftn`` / ``ifftn`` in place of ``rfftn`` / ``irfftn``; the reference's
``tests/admm/test_cbpdn.py:179-201``).

The real and imaginary parts of the signal and of the coefficient maps are the two CHANNELS of a
real solver, so every transform on the device stays a real one; the X step pairs the channels (at
a stored half-spectrum frequency f: A + iB is the spectrum at f, A - iB the conjugate of the one at
-f; two Sherman-Morrison solves, ``csrc/ck_admm.hip sm_cplx_kernel``), and the complex soft
threshold of the y step (``prox_l1`` of a complex array shrinks the modulus, prox/_lp.py:144-183)
is the l2 shrinkage over the channel pair -- the y step of ``ConvBPDNJoint`` with an l1 weight of
zero and ``mu = lambda``, whose l2,1 sum is the complex ``||x||_1``.  Residuals and the data
fidelity term are sums of squares over (re, im) and carry over unchanged.

Served: a single-channel complex signal (any number of images), a single-channel dictionary, scalar
``lmbda``; not ``Y0`` / ``U0`` / ``L1Weight`` arrays / ``NonNegCoef`` / ``LinSolveCheck`` (they
raise).  The other complex classes of the reference (``ccmod``) raise as before.
"""

import collections

import numpy as np

from .. import cnvrep as cr
from . import cbpdn as _c

__all__ = ['ComplexConvBPDN']


class _PairJoint(_c.ConvBPDNJoint):
    """The real solver underneath: ConvBPDNJoint on the (re, im) channel pair with the
    imaginary part of the dictionary uploaded beside the real one."""

    _D_imag = None

    def setdict(self, D=None):
        super(_PairJoint, self).setdict(D)
        self._dev.set_dict_imag(self._D_imag)


def _split(a, axis):
    return np.concatenate([a.real, a.imag], axis=axis)


class ComplexConvBPDN(object):
    """Same constructor, options, attributes and statistics as ``ConvBPDN``; arrays are complex."""

    Options = _c.ConvBPDN.Options
    itstat_fields_objfn = _c.ConvBPDN.itstat_fields_objfn

    def __init__(self, D, S, lmbda=None, opt=None, dimK=None, dimN=2, **backend):
        if opt is None:
            opt = _c.ConvBPDN.Options()
        if not isinstance(opt, _c.GenericConvBPDN.Options):
            raise TypeError("Parameter opt must be an instance of ConvBPDN.Options")
        if dimN != 2:
            raise NotImplementedError("sporco_amd handles dimN = 2 (images)")
        D, S = np.asarray(D), np.asarray(S)
        self.cri = cr.CSC_ConvRepIndexing(D, S, dimK=dimK, dimN=dimN)
        if self.cri.Cd > 1 or self.cri.C > 1:
            raise NotImplementedError("complex-valued ConvBPDN: single-channel signal and dictionary")
        if lmbda is None:
            raise NotImplementedError("complex-valued ConvBPDN: lmbda must be given")
        for key in ('Y0', 'U0'):
            if opt[key] is not None:
                raise NotImplementedError("complex-valued ConvBPDN: option %s is not supported" % key)
        if np.ndim(opt['L1Weight']) > 0 and np.size(opt['L1Weight']) > 1:
            raise NotImplementedError("complex-valued ConvBPDN: scalar L1Weight only")
        if opt['NonNegCoef'] or opt['LinSolveCheck']:
            raise NotImplementedError("complex-valued ConvBPDN: NonNegCoef / LinSolveCheck are not supported")
        cdt = np.dtype(opt['DataType']) if opt['DataType'] is not None else np.result_type(S.dtype, np.complex64)
        if cdt.kind != 'c':
            cdt = np.result_type(cdt, np.complex64)
        self.dtype = np.dtype(cdt)
        rdt = np.float32 if self.dtype == np.complex64 else np.float64
        self.real_dtype = False
        self.opt = opt
        self.lmbda = rdt(lmbda)
        H, W = self.cri.Nv
        N, M = self.cri.K, self.cri.M
        self.D = np.asarray(D.reshape(self.cri.shpD), dtype=self.dtype)
        self.S = np.asarray(S.reshape(self.cri.shpS), dtype=self.dtype)
        Sr = _split(self.S, 2)[..., 0].astype(rdt)                 # (H, W, 2, N)
        jopt = _c.ConvBPDNJoint.Options(dict(opt))
        w = float(np.asarray(opt['L1Weight']).ravel()[0])
        jopt['L1Weight'] = 1.0
        jopt['L21Weight'] = w
        jopt['DataType'] = rdt
        if opt['rho'] is None:
            jopt['rho'] = float(50.0 * self.lmbda + 1.0)                      # cbpdn.py:584
        if opt['AutoRho', 'RsdlTarget'] is None:
            jopt['AutoRho', 'RsdlTarget'] = (float(1.0 + 18.3 ** (np.log10(self.lmbda) + 1.0))
                                             if self.lmbda != 0.0 else 1.0)   # cbpdn.py:588-593
        if backend.get('reducer') is not None:
            # (the tolerances below count complex elements of ONE rank's shard; image shards of a
            # complex-valued problem are not covered by any test)
            raise NotImplementedError('complex-valued signals are not supported with image shards')
        inner = _PairJoint.__new__(_PairJoint)
        inner._D_imag = np.ascontiguousarray(self.D.imag.reshape(self.D.shape[0], self.D.shape[1], M).astype(rdt))
        Dre = np.ascontiguousarray(self.D.real.reshape(self.D.shape[0], self.D.shape[1], M).astype(rdt))
        inner.__init__(Dre, Sr, 0.0, float(self.lmbda), jopt, dimK=1, dimN=2, **backend)
        # residual tolerances count complex elements (admm.py:462-486 with Nx = Nc = prod(shpX))
        inner.Nx = inner.Nc = int(np.prod(self.cri.shpX))
        self._inner = inner
        self.timer = inner.timer
        fields = list(inner.IterationStats._fields)
        fields.remove('RegL21')
        self.IterationStats = collections.namedtuple('IterationStats', fields)

    # -- state ------------------------------------------------------------------------------
    def _pair(self, a):
        a = np.asarray(a)
        return (a[:, :, 0:1] + 1j * a[:, :, 1:2]).astype(self.dtype)

    X = property(lambda self: self._pair(self._inner.X))
    Y = property(lambda self: self._pair(self._inner.Y))
    U = property(lambda self: self._pair(self._inner.U))
    rho = property(lambda self: self._inner.rho)
    k = property(lambda self: self._inner.k)

    @property
    def itstat(self):
        return [self._convert(t) for t in self._inner.itstat]

    def _convert(self, t):
        d = t._asdict()
        d['RegL1'] = d.pop('RegL21')          # sum of moduli = the l2,1 sum over the (re, im) pair
        return self.IterationStats(**d)

    def getitstat(self):
        from ..util import transpose_ntpl_list
        return transpose_ntpl_list(self.itstat)

    # -- solver interface -------------------------------------------------------------------
    def solve(self):
        self._inner._return_min = False
        self._inner.solve()
        return self.getmin()

    def getmin(self):
        return self.X if self.opt['ReturnX'] else self.Y

    def getcoef(self):
        return self.getmin()

    def var_x(self):
        return self.X

    def var_y(self):
        return self.Y

    def var_u(self):
        return self.U

    def setdict(self, D=None):
        if D is not None:
            self.D = np.asarray(D, dtype=self.dtype).reshape(self.D.shape)
        rdt = self._inner.dtype
        M = self.cri.M
        self._inner._D_imag = np.ascontiguousarray(
            self.D.imag.reshape(self.D.shape[0], self.D.shape[1], M).astype(rdt))
        self._inner.setdict(np.ascontiguousarray(self.D.real.astype(rdt)).reshape(self._inner.D.shape))

    def reconstruct(self, X=None):
        """ifftn(sum_m Df fftn(X)) (cbpdn.py:373-380)."""
        if X is None:
            Xr = None
        else:
            X = np.asarray(X, dtype=self.dtype).reshape(self.cri.shpX)
            Xr = _split(X, 2).astype(self._inner.dtype)
        R = np.asarray(self._inner.reconstruct(Xr))
        return self._pair(R.reshape(R.shape[0], R.shape[1], 2, -1))          # (H, W, C = 1, N)
