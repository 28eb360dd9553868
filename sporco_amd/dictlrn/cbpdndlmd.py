"""Convolutional dictionary learning with a spatial mask in the data fidelity term.

Drop-in for ``sporco.dictlrn.cbpdndlmd.ConvBPDNMaskDictLearn`` (sporco/dictlrn/cbpdndlmd.py:
219-543): ``xmethod='admm'`` (the default, mask decoupling:
:class:`sporco_amd.admm.cbpdn.ConvBPDNMaskDcpl`) or ``'pgm'``
(:class:`sporco_amd.pgm.cbpdn.ConvBPDNMask`); ``dmethod='pgm'``
(:class:`sporco_amd.pgm.ccmod.ConvCnstrMODMask`), ``'ism'`` or ``'cg'`` (mask decoupling:
:mod:`sporco_amd.admm.ccmodmd`) or ``'cns'`` (its consensus form).
Both steps share one device handle, as in :mod:`sporco_amd.dictlrn.cbpdndl`.
"""

import copy

from . import cbpdndl
from . import common as dc
from . import dictlrn
from .. import _lib
from .. import cnvrep as cr
from ..admm import cbpdn as admm_cbpdn
from ..admm import ccmodmd as admm_ccmodmd
from ..pgm import cbpdn as pgm_cbpdn
from ..pgm import ccmod as pgm_ccmod

__all__ = ['ConvBPDNMaskDictLearn']


def _x_class(method):
    if method == 'pgm':
        return pgm_cbpdn.ConvBPDNMask
    if method == 'admm':
        return admm_cbpdn.ConvBPDNMaskDcpl
    raise ValueError('Unknown ConvBPDNMask solver method %s' % method)


def _d_class(method):
    if method == 'pgm':
        return pgm_ccmod.ConvCnstrMODMask
    if method == 'ism':
        return admm_ccmodmd.ConvCnstrMODMaskDcpl_IterSM
    if method == 'cg':
        return admm_ccmodmd.ConvCnstrMODMaskDcpl_CG
    if method == 'cns':
        return admm_ccmodmd.ConvCnstrMODMaskDcpl_Consensus
    raise ValueError('Unknown ConvCnstrMODMask solver method %s' % method)


class ConvBPDNMaskDictLearn(cbpdndl.ConvBPDNDictLearn):
    r"""Minimise (1/2) sum_k ||W (sum_m d_m * x_{k,m} - s_k)||^2 + lambda sum ||x_{k,m}||_1
    over coefficient maps and constrained filters by alternating one masked sparse coding step
    and one masked dictionary update per outer iteration."""

    class Options(dictlrn.DictLearn.Options):
        """``AccurateDFid``, ``DictSize``, ``CBPDN``, ``CCMOD`` (cbpdndlmd.py:331-380)."""

        defaults = copy.deepcopy(dictlrn.DictLearn.Options.defaults)
        defaults.update({'DictSize': None, 'AccurateDFid': False})

        def __init__(self, opt=None, xmethod=None, dmethod=None):
            self.xmethod = 'admm' if xmethod is None else xmethod
            self.dmethod = 'pgm' if dmethod is None else dmethod
            xcls, dcls = _x_class(self.xmethod), _d_class(self.dmethod)
            xd = copy.deepcopy(xcls.Options.defaults)
            xd.update({'MaxMainIter': 1})
            if self.xmethod == 'admm':                         # (cbpdndlmd.py:51-55)
                xd['AutoRho'].update({'Period': 10, 'AutoScaling': False, 'RsdlRatio': 10.0,
                                      'Scaling': 2.0, 'RsdlTarget': 1.0})
            dd = copy.deepcopy(dcls.Options.defaults)
            dd.update({'MaxMainIter': 1})
            if self.dmethod != 'pgm':                          # (cbpdndlmd.py:147-151)
                dd['AutoRho'].update({'Period': 10, 'AutoScaling': False, 'RsdlRatio': 10.0,
                                      'Scaling': 2.0, 'RsdlTarget': 1.0})
            self.defaults.update({'CBPDN': xd, 'CCMOD': dd})
            dictlrn.DictLearn.Options.__init__(self, {'CBPDN': xcls.Options(xd),
                                                      'CCMOD': dcls.Options(dd)})
            self.update({} if opt is None else opt)

    def __init__(self, D0, S, lmbda, W, opt=None, xmethod=None, dmethod=None, dimK=1, dimN=2,
                 device=0, stream=None, reducer=None):
        """``W``: mask compatible with the *internal* layout of ``S`` (cbpdndlmd.py:383-395),
        e.g. (H, W, 1, K) for K greyscale images.  ``reducer``: as for
        :class:`sporco_amd.dictlrn.cbpdndl.ConvBPDNDictLearn` (``S`` and ``W`` hold this rank's
        images; ``dmethod='pgm'``)."""
        if opt is None:
            opt = ConvBPDNMaskDictLearn.Options(xmethod=xmethod, dmethod=dmethod)
        if xmethod is None:
            xmethod = opt.xmethod
        if dmethod is None:
            dmethod = opt.dmethod
        if opt.xmethod != xmethod or opt.dmethod != dmethod:
            raise ValueError('Parameters xmethod and dmethod must have the same values used '
                             'to initialise the Options object')
        xcls, dcls = _x_class(xmethod), _d_class(dmethod)
        if reducer is not None and dmethod not in ('pgm', 'cns'):
            raise NotImplementedError(
                "image sharding is offered for dmethod='pgm' (gradient all-reduce) and 'cns' "
                "(consensus average all-reduce); 'ism' / 'cg' solve with all images' coefficient "
                "spectra at once")
        self._reducer = reducer
        bk = {} if reducer is None else {'reducer': reducer}
        self.opt, self.xmethod, self.dmethod = opt, xmethod, dmethod
        dsz = D0.shape if opt['DictSize'] is None else opt['DictSize']
        cri = cr.CDU_ConvRepIndexing(dsz, S, dimK, dimN)
        D0 = cr.Pcn(D0, dsz, cri.Nv, dimN, cri.dimCd, crp=True, zm=opt['CCMOD', 'ZeroMean'])
        # PGM: X0; ADMM: block 1 of Y0, block 0 zero (cbpdndlmd.py:433-444)
        optname = 'X0' if dmethod == 'pgm' else 'Y0'
        opt['CCMOD'].update({optname: cr.zpad(cr.stdformD(D0, cri.Cd, cri.M, dimN), cri.Nv)})
        xstep = xcls(D0, S, lmbda, W, opt['CBPDN'], dimK=dimK, dimN=dimN, device=device,
                     stream=stream, **bk)
        xdev = xstep._dev if xmethod == 'admm' else xstep.dev
        xdev = getattr(xdev, '_raw', xdev)
        dstep = dcls(None, S, W, dsz, opt['CCMOD'], dimK=dimK, dimN=dimN, dev=xdev, **bk)
        xstep._return_min = False
        dstep._return_min = False
        isc = dictlrn.IterStatsConfig(
            isfld=dc.isfld(xmethod, dmethod, opt), isxmap=dc.isxmap(xmethod, opt),
            isdmap=dc.isdmap(dmethod), evlmap=dc.evlmap(opt['AccurateDFid']),
            hdrtxt=dc.hdrtxt(xmethod, dmethod, opt), hdrmap=dc.hdrmap(xmethod, dmethod, opt),
            fmtmap={'It_X': '%4d', 'It_D': '%4d'})
        dictlrn.DictLearn.__init__(self, xstep, dstep, opt, isc)

    def evaluate(self):
        """Objective after the D update with the mask applied (cbpdndlmd.py:522-543)."""
        if not self.opt['AccurateDFid']:
            return None
        dev = self.dstep.dev
        dfd = dev.masked_grad(_lib.VAR_DXF, True, False)[_lib.PGM_DFID] / 2.0
        rl1 = dev.asum(self._coef_var())
        if self._reducer is not None:
            dfd, rl1 = self._reducer.sum([dfd, rl1])
        return dict(DFid=dfd, RegL1=rl1, ObjFun=dfd + self.xstep.lmbda * rl1)
