"""Convolutional dictionary learning (ConvBPDN X-step / constrained MOD D-step).

Drop-in for ``sporco.dictlrn.cbpdndl`` (sporco/dictlrn/cbpdndl.py:31-524):
``ConvBPDNDictLearn(D0, S, lmbda, opt, xmethod, dmethod, dimK, dimN)`` with the
same Options tree (``CBPDN`` / ``CCMOD`` sub-options built from the selected
inner solver classes) and IterationStats.  ``xmethod`` is ``'admm'`` or
``'pgm'``; ``dmethod`` is ``'pgm'`` (the reference default) or one of the ADMM updates of
sporco_amd.admm.ccmod: ``'cns'`` (consensus), ``'ism'`` (iterated Sherman-Morrison, up to 8
training images) or ``'cg'`` (conjugate gradients).

Both inner solvers share ONE device handle: after the X-step the coefficient
maps are transformed in place on the GPU for the D-step (``setcoef``), and after
the D-step the new dictionary spectrum is copied device-to-device into the
X-step (``setdict``); nothing X-sized crosses PCIe during learning.
"""

import copy

import numpy as np

from . import common as dc
from . import dictlrn
from .. import _lib
from .. import cnvrep as cr
from ..admm import cbpdn as admm_cbpdn
from ..admm import ccmod as admm_ccmod
from ..pgm import cbpdn as pgm_cbpdn
from ..pgm import ccmod as pgm_ccmod

__all__ = ['cbpdn_class_label_lookup', 'ConvBPDNOptionsDefaults', 'ConvBPDNOptions',
           'ConvBPDN', 'ccmod_class_label_lookup', 'ConvCnstrMODOptionsDefaults',
           'ConvCnstrMODOptions', 'ConvCnstrMOD', 'ConvBPDNDictLearn']

_XCLS = {'admm': admm_cbpdn.ConvBPDN, 'pgm': pgm_cbpdn.ConvBPDN}
_DCLS = {'pgm': pgm_ccmod.ConvCnstrMOD, 'cns': admm_ccmod.ConvCnstrMOD_Consensus,
         'ism': admm_ccmod.ConvCnstrMOD_IterSM, 'cg': admm_ccmod.ConvCnstrMOD_CG}
_dyn = {}


def cbpdn_class_label_lookup(label):
    if label in _XCLS:
        return _XCLS[label]
    raise ValueError('Unknown ConvBPDN solver method %s' % label)


def ccmod_class_label_lookup(label):
    if label in _DCLS:
        return _DCLS[label]
    raise ValueError('Unknown ConvCnstrMOD solver method %s' % label)


def ConvBPDNOptionsDefaults(method='admm'):
    """Inner X-step defaults inside dictionary learning (cbpdndl.py:43-56)."""
    dflt = copy.deepcopy(cbpdn_class_label_lookup(method).Options.defaults)
    dflt.update({'MaxMainIter': 1})
    if method == 'admm':
        dflt['AutoRho'].update({'Period': 10, 'AutoScaling': False, 'RsdlRatio': 10.0,
                                'Scaling': 2.0, 'RsdlTarget': 1.0})
    return dflt


def ConvCnstrMODOptionsDefaults(method='pgm'):
    """Inner D-step defaults inside dictionary learning (cbpdndl.py:139-152)."""
    dflt = copy.deepcopy(ccmod_class_label_lookup(method).Options.defaults)
    dflt.update({'MaxMainIter': 1})
    if method != 'pgm':
        dflt['AutoRho'].update({'Period': 10, 'AutoScaling': False, 'RsdlRatio': 10.0,
                                'Scaling': 2.0, 'RsdlTarget': 1.0})
    return dflt


def _dynamic(kind, base, method):
    """Subclass of ``base`` registered under a module-level name so that
    instances pickle (the role of common._fix_dynamic_class_lookup)."""
    key = (kind, method)
    if key not in _dyn:
        name = '_%s_%s' % (kind, method)
        cls = type(name, (base,), {'__module__': __name__})
        cls.__qualname__ = name
        globals()[name] = cls
        _dyn[key] = cls
    return _dyn[key]


def ConvBPDNOptions(opt=None, method='admm'):
    base = cbpdn_class_label_lookup(method).Options
    return _dynamic('ConvBPDNOptions', base, method)(opt)


def ConvCnstrMODOptions(opt=None, method='pgm'):
    base = ccmod_class_label_lookup(method).Options
    return _dynamic('ConvCnstrMODOptions', base, method)(opt)


def ConvBPDN(*args, **kwargs):
    """Construct the sparse coding solver selected by keyword ``method``."""
    method = kwargs.pop('method', 'admm')
    return _dynamic('ConvBPDN', cbpdn_class_label_lookup(method), method)(*args, **kwargs)


def ConvCnstrMOD(*args, **kwargs):
    """Construct the dictionary update solver selected by keyword ``method``."""
    method = kwargs.pop('method', 'pgm')
    return _dynamic('ConvCnstrMOD', ccmod_class_label_lookup(method), method)(*args, **kwargs)


class ConvBPDNDictLearn(dictlrn.DictLearn):
    r"""Minimise (1/2) sum_k ||sum_m d_m * x_{k,m} - s_k||^2 + lambda sum ||x_{k,m}||_1
    over coefficient maps and unit-norm, support-constrained filters by
    interleaving one sparse coding step and one dictionary update step per
    outer iteration.

    IterationStats fields: ``Iter, ObjFun, DFid, RegL1, Cnstr``, then
    ``XPrRsdl, XDlRsdl, XRho`` (ADMM X-step) or ``X_L, X_Rsdl`` (PGM X-step),
    then ``D_L, D_Rsdl`` (+ backtracking fields when enabled), ``Time``.
    """

    class Options(dictlrn.DictLearn.Options):
        """Adds ``AccurateDFid``, ``DictSize``, ``CBPDN``, ``CCMOD``
        (cbpdndl.py:332-382)."""

        defaults = copy.deepcopy(dictlrn.DictLearn.Options.defaults)
        defaults.update({'DictSize': None, 'AccurateDFid': False})

        def __init__(self, opt=None, xmethod=None, dmethod=None):
            self.xmethod = 'admm' if xmethod is None else xmethod
            self.dmethod = 'pgm' if dmethod is None else dmethod
            self.defaults.update({'CBPDN': ConvBPDNOptionsDefaults(self.xmethod),
                                  'CCMOD': ConvCnstrMODOptionsDefaults(self.dmethod)})
            dictlrn.DictLearn.Options.__init__(self, {
                'CBPDN': ConvBPDNOptions(self.defaults['CBPDN'], method=self.xmethod),
                'CCMOD': ConvCnstrMODOptions(self.defaults['CCMOD'], method=self.dmethod)})
            self.update({} if opt is None else opt)

    _dim1 = False       # (set per object by __init__; classes that share the coupling methods keep it)
    _dim3 = False

    def __init__(self, D0, S, lmbda=None, opt=None, xmethod=None, dmethod=None, dimK=1,
                 dimN=2, device=0, stream=None, reducer=None):
        """Arguments as in the reference (cbpdndl.py:385-423).  Backend keywords: ``device``,
        ``stream``, and ``reducer`` (:class:`sporco_amd.dist.TorchReducer`) for one process per
        GPU with ``S`` holding this rank's block of the training images: the
        X-step exchanges its per-iteration sums, the D-step all-reduces its gradient; every
        rank ends with the same dictionary.  Offered for ``dmethod='pgm'`` and ``'cns'`` (whose consensus
        average over the images becomes an all-reduce) with either X-step."""
        self._reducer = reducer
        if opt is None:
            opt = ConvBPDNDictLearn.Options(xmethod=xmethod, dmethod=dmethod)
        if xmethod is None:
            xmethod = opt.xmethod
        if dmethod is None:
            dmethod = opt.dmethod
        if reducer is not None and dmethod not in ('pgm', 'cns'):
            raise NotImplementedError(
                "image sharding is offered for dmethod='pgm' (gradient all-reduce) and 'cns' "
                "(consensus average all-reduce); 'ism' / 'cg' solve with all images' coefficient "
                "spectra at once")
        if opt.xmethod != xmethod or opt.dmethod != dmethod:
            raise ValueError('Parameters xmethod and dmethod must have the same values used '
                             'to initialise the Options object')
        self.opt, self.xmethod, self.dmethod = opt, xmethod, dmethod
        dsz = D0.shape if opt['DictSize'] is None else opt['DictSize']
        self._dim3 = dimN == 3
        if self._dim3:
            # volumes: the steps fold the first two axes themselves (admm/cbpdn.py, pgm/ccmod.py,
            # admm/ccmod.py) and share a volume handle; the set-up arithmetic below is dimN-generic
            if dmethod not in ('pgm', 'cns') or xmethod not in ('admm', 'pgm'):
                raise NotImplementedError("dimN = 3 is offered with xmethod 'admm' / 'pgm' and "
                                          "dmethod 'pgm' / 'cns'")
        self._dim1 = dimN == 1
        if self._dim1:
            # signals: both steps run them as images with a unit first axis (admm/cbpdn.py,
            # pgm/ccmod.py); the set-up arithmetic below does the same and drops the axis again
            from ..pgm.ccmod import _dsz_unit_axis
            cri = cr.CDU_ConvRepIndexing(_dsz_unit_axis(dsz), np.asarray(S)[np.newaxis], dimK, 2)
            D0 = cr.Pcn(np.asarray(D0)[np.newaxis], _dsz_unit_axis(dsz), cri.Nv, 2, cri.dimCd,
                        crp=True, zm=opt['CCMOD', 'ZeroMean'])
            opt['CCMOD'].update({('X0' if dmethod == 'pgm' else 'Y0'):
                                 cr.zpad(cr.stdformD(D0, cri.Cd, cri.M, 2), cri.Nv)})
            D0 = D0[0]
        else:
            cri = cr.CDU_ConvRepIndexing(dsz, S, dimK, dimN)
            # normalise the initial dictionary and hand it (zero-padded) to the D-step
            D0 = cr.Pcn(D0, dsz, cri.Nv, dimN, cri.dimCd, crp=True, zm=opt['CCMOD', 'ZeroMean'])
            optname = 'X0' if dmethod == 'pgm' else 'Y0'        # (cbpdndl.py:443-445)
            opt['CCMOD'].update({optname: cr.zpad(cr.stdformD(D0, cri.Cd, cri.M, dimN), cri.Nv)})
        bk = {} if reducer is None else {'reducer': reducer}
        xstep = ConvBPDN(D0, S, lmbda, opt['CBPDN'], method=xmethod, dimK=dimK, dimN=dimN,
                         device=device, stream=stream, **bk)
        if xmethod == 'admm' and not opt['CBPDN', 'ReturnX']:
            # the alternation only ever consumes Y (var_y) of the X-step: tell the
            # device that X / Xf of the inner iterations are never read.  (With ReturnX the
            # D-step is fed xstep.getcoef() = X, as in the reference, dictlrn.py:379-382.)
            xstep._no_x = True
            # ... and between its one-iteration solves nothing but setcoef(Y) looks at the iterate:
            # it stays in the single-array form of the fused iteration (csc_rows.h)
            xstep._dev.set_hint(_lib.HINT_KEEP_VFORM, 1)
        xdev = xstep._dev if xmethod == 'admm' else xstep.dev
        xdev = getattr(xdev, '_raw', xdev)     # (the D-step's own sums are reduced explicitly)
        dstep = ConvCnstrMOD(None, S, dsz, opt['CCMOD'], method=dmethod, dimK=dimK, dimN=dimN,
                             dev=xdev, **bk)
        # dictlrn.DictLearn.solve ignores what the inner solve() calls return
        # (dictlrn.py:333,338): do not copy the iterates to the host every outer iteration
        xstep._return_min = False
        dstep._return_min = False
        isc = dictlrn.IterStatsConfig(
            isfld=dc.isfld(xmethod, dmethod, opt), isxmap=dc.isxmap(xmethod, opt),
            isdmap=dc.isdmap(dmethod), evlmap=dc.evlmap(opt['AccurateDFid']),
            hdrtxt=dc.hdrtxt(xmethod, dmethod, opt), hdrmap=dc.hdrmap(xmethod, dmethod, opt),
            fmtmap={'It_X': '%4d', 'It_D': '%4d'})
        super(ConvBPDNDictLearn, self).__init__(xstep, dstep, opt, isc)

    # -- coupling between the two steps, kept on the device -------------------------------
    def _coef_var(self):
        """The device array behind ``xstep.getcoef()``: X for the PGM X-step, and for the ADMM
        ones what ``ReturnX`` / ``ReturnVar`` select (admm.py:955-956, cbpdn.py:1693-1704)."""
        if self.xmethod != 'admm':
            return _lib.VAR_X
        o = self.xstep.opt
        if 'ReturnVar' in o:
            if o['ReturnVar'] == 'Y0':
                raise NotImplementedError("ReturnVar 'Y0' is not a coefficient array")
            return _lib.VAR_X if o['ReturnVar'] == 'X' else _lib.VAR_Y
        return _lib.VAR_X if o['ReturnX'] else _lib.VAR_Y

    def post_xstep(self):
        """dstep.setcoef(xstep.getcoef()) (dictlrn.py:379-382) without the host trip."""
        self.dstep.setcoef_from_device(self._coef_var())

    def post_dstep(self):
        """xstep.setdict(dstep.getdict()) (dictlrn.py:386-389), device to device."""
        dsz = self.dstep.cri.mxsz       # (largest support of a multi-scale dictionary)
        self.dstep.dev.setdict_from_dstep(dsz[0], dsz[1])
        x = self.xstep
        x._cache.pop(_lib.VAR_DF, None)
        if hasattr(x, '_fcache'):
            x._fcache.clear()
        # keep the host attribute current: a (dH, dW, 1, 1, M) crop, a few KB
        x.D = self.dstep.getdict()
        if self._dim1:
            x.D = x.D[np.newaxis]       # (the X-step keeps its arrays with the unit axis)
        if self._dim3:                  # (... and a volume's dictionary zero-padded and folded)
            Dz, Hs = x._dim3
            x.D = cr.fold3(cr.zpad(x.D, (Dz, Hs, x.cri.Nv[1])), Dz, Hs)

    def getdict(self, crop=True):
        return self.dstep.getdict(crop=crop)

    def getcoef(self):
        return self.xstep.getcoef()

    def reconstruct(self, D=None, X=None):
        """irfftn(sum_m rfftn(D) rfftn(X)) (cbpdndl.py:486-498)."""
        if D is None and X is None:
            dev = self.xstep._dev if self.xmethod == 'admm' else self.xstep.dev
            r = dev.reconstruct(self._coef_var())         # (5-D, as the reference's inner())
            if self._dim3:
                return cr.unfold3(r, *self.xstep._dim3)
            return r[0] if self._dim1 else r
        if D is None:
            D = self.getdict(crop=False)
        if X is None:
            X = self.getcoef()
        Nv = self.dstep.cri.Nv
        if self._dim3:
            Nv = self.dstep._cri3.Nv
            Df = np.fft.rfftn(D, Nv, axes=(0, 1, 2))
            Xf = np.fft.rfftn(X, Nv, axes=(0, 1, 2))
            return np.fft.irfftn(np.sum(Df * Xf, axis=5, keepdims=True), Nv, axes=(0, 1, 2))
        if self._dim1:
            Df = np.fft.rfft(D, Nv[1], axis=0)
            Xf = np.fft.rfft(X, Nv[1], axis=0)
            return np.fft.irfft(np.sum(Df * Xf, axis=3, keepdims=True), Nv[1], axis=0)
        Df = np.fft.rfftn(D, Nv, axes=(0, 1))
        Xf = np.fft.rfftn(X, Nv, axes=(0, 1))
        return np.fft.irfftn(np.sum(Df * Xf, axis=4, keepdims=True), Nv, axes=(0, 1))

    def evaluate(self):
        """Objective after the D update (``AccurateDFid``; cbpdndl.py:502-524):
        data fidelity with the new dictionary and the current coefficients
        (already on device as Zf), l1 norm of the coefficients."""
        if not self.opt['AccurateDFid']:
            return None
        dev = self.dstep.dev
        if self.xmethod == 'admm' and self._coef_var() != _lib.VAR_Y:
            # ReturnX: the D-step was given X, but the reference evaluates at var_y()
            # (cbpdndl.py:509-511): reconstruct D * Y with the new dictionary (the X-step
            # already has it) and take the residual in the spatial domain -- signal sized
            rec = self.xstep._dev.reconstruct(_lib.VAR_Y)
            dfd = 0.5 * float(np.sum((rec.astype(np.float64) - self.xstep.S) ** 2))
            rl1 = dev.asum(_lib.VAR_Y)
            if self._reducer is not None:
                dfd, rl1 = self._reducer.sum([dfd, rl1])
            return dict(DFid=dfd, RegL1=rl1, ObjFun=dfd + self.xstep.lmbda * rl1)
        dfd = dev.ccmod_eval(_lib.VAR_DXF)[_lib.PGM_DFID] / 2.0
        rl1 = dev.asum(self._coef_var())
        if self._reducer is not None:
            dfd, rl1 = self._reducer.sum([dfd, rl1])
        return dict(DFid=dfd, RegL1=rl1, ObjFun=dfd + self.xstep.lmbda * rl1)
