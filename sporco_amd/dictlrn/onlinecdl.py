"""Online convolutional dictionary learning by stochastic gradient descent, on the GPU.

Drop-in for ``sporco.dictlrn.onlinecdl.OnlineConvBPDNDictLearn``
(sporco/dictlrn/onlinecdl.py:33-460): same constructor, Options tree, ``solve(S)`` per
training image (or mini-batch) and IterationStats fields.  Each ``solve`` call runs the ADMM
ConvBPDN X-step of this package for the new data with the current dictionary, then takes one
projected gradient step on the dictionary with step size ``eta_a / (j + eta_b)``.

The two steps share the X-step's device handle: the coefficient maps are transformed in place
for the gradient (``ccmod_setcoef`` / ``ccmod_grad`` at the current dictionary spectrum), the
step, inverse transform and constraint projection run on the device
(``sporco_amd_csc_ccmod_sgd_step``), and only the cropped dictionary (a few KB) returns to the
host.  ``OnlineConvBPDNMaskDictLearn`` (onlinecdl.py:464-600) does the same with a spatial mask:
X-step by mask decoupling (:class:`sporco_amd.admm.cbpdn.ConvBPDNMaskDcpl`), gradient through
``sporco_amd_csc_masked_grad``.
"""

import copy

import numpy as np

from .. import _lib
from .. import cdict
from .. import common
from .. import cnvrep as cr
from .. import util
from ..admm import cbpdn

__all__ = ['OnlineConvBPDNDictLearn', 'OnlineConvBPDNMaskDictLearn']


class OnlineConvBPDNDictLearn(common.IterativeSolver):
    r"""Stochastic-gradient online convolutional dictionary learning (onlinecdl.py:33-460).

    IterationStats fields: ``Iter, ObjFun, DFid, RegL1, PrimalRsdl, DualRsdl, Rho, Cnstr,
    DeltaD, Eta, Time``.
    """

    class Options(cdict.ConstrainedDict):
        """``Verbose, StatusHeader, IterTimer, DictSize, DataType, ZeroMean, eta_a, eta_b,
        CBPDN`` as onlinecdl.py:45-102 (X-step default: 100 iterations, AutoRho period 10).
        ``CUDA_CBPDN`` is accepted for compatibility and must stay False: the X-step always
        runs on the GPU here."""

        defaults = {'Verbose': False, 'StatusHeader': True, 'IterTimer': 'solve',
                    'DictSize': None, 'DataType': None, 'ZeroMean': False, 'eta_a': 10.0,
                    'eta_b': 5.0, 'CUDA_CBPDN': False,
                    'CBPDN': copy.deepcopy(cbpdn.ConvBPDN.Options.defaults)}

        def __init__(self, opt=None):
            cdict.ConstrainedDict.__init__(self, {
                'CBPDN': cbpdn.ConvBPDN.Options({
                    'MaxMainIter': 100,
                    'AutoRho': {'Period': 10, 'AutoScaling': False, 'RsdlRatio': 10.0,
                                'Scaling': 2.0, 'RsdlTarget': 1.0}})})
            self.update({} if opt is None else opt)

    fwiter = 4
    fpothr = 2
    itstat_fields_objfn = ('ObjFun', 'DFid', 'RegL1')
    itstat_fields_alg = ('PrimalRsdl', 'DualRsdl', 'Rho', 'Cnstr', 'DeltaD', 'Eta')
    itstat_fields_extra = ()

    def __new__(cls, *args, **kwargs):
        obj = super(OnlineConvBPDNDictLearn, cls).__new__(cls)
        obj.timer = util.Timer(['init', 'solve', 'solve_wo_eval'])
        obj.timer.start('init')
        return obj

    def __init__(self, D0, lmbda=None, opt=None, dimK=None, dimN=2, device=0, stream=None):
        """``D0, lmbda, opt, dimK, dimN`` as in the reference (onlinecdl.py:135-207); ``device``
        / ``stream`` select the GPU and HIP stream of the X-step handles."""
        if opt is None:
            opt = OnlineConvBPDNDictLearn.Options()
        if not isinstance(opt, OnlineConvBPDNDictLearn.Options):
            raise TypeError('Parameter opt must be an instance of '
                            'OnlineConvBPDNDictLearn.Options')
        if opt['CUDA_CBPDN']:
            raise ValueError('CUDA_CBPDN selects the sporco_cuda extension; this backend runs '
                             'the X-step on the AMD GPU already')
        if dimN != 2:
            raise NotImplementedError("sporco_amd handles dimN = 2 (images)")
        self.opt = opt
        self.dimK, self.dimN = dimK, dimN
        self._device, self._stream = device, stream
        self.set_dtype(opt, D0.dtype)
        if self.dtype not in (np.float32, np.float64):
            raise TypeError("sporco_amd works in float32 or float64, not %s" % self.dtype)
        self.lmbda = lmbda
        self.eta_a, self.eta_b = opt['eta_a'], opt['eta_b']
        self.set_attr('eta', opt['eta_a'] / opt['eta_b'], dval=2.0, dtype=self.dtype)
        self.dsz = D0.shape if opt['DictSize'] is None else opt['DictSize']
        self.cri = None
        ds = cr.DictionarySize(self.dsz, dimN)
        self._mxsz, self._fsz = ds.mxsz, ds.fsz     # (multi-scale DictSize: per-filter supports)
        self._dimCd = ds.ndim - dimN - 1
        D0 = cr.stdformD(D0, ds.nchn, ds.nflt, dimN).astype(self.dtype)
        self.D = cr.Pcn(D0, self.dsz, (), dimN, self._dimCd, crp=True, zm=opt['ZeroMean'])
        self.Dprv = self.D.copy()
        self.itstat = []
        self.j = 0
        self.display_config()

    def solve(self, S, dimK=None):
        """Sparse coding and dictionary update for training data ``S``
        (onlinecdl.py:211-240); returns the updated dictionary."""
        if dimK is None and self.dimK is not None:
            dimK = self.dimK
        if self.j == 0:
            self.display_start()
        self.timer.start(['solve', 'solve_wo_eval'])
        self.init_vars(S, dimK)
        self.xstep(S, self.lmbda, dimK)
        self.dstep()
        self.timer.stop('solve_wo_eval')
        self.manage_itstat()
        self.j += 1
        self.timer.stop('solve')
        return self.getdict()

    def init_vars(self, S, dimK):
        Nv = S.shape[0:self.dimN]
        if self.cri is None or Nv != self.cri.Nv:
            self.cri = cr.CDU_ConvRepIndexing(self.dsz, S, dimK, self.dimN)

    def xstep(self, S, lmbda, dimK):
        """ConvBPDN for the new data with the current dictionary (onlinecdl.py:267-287); the
        solver object (and with it the device handle holding the coefficient maps) is kept
        for the dictionary step."""
        self._release_xstep()
        x = cbpdn.ConvBPDN(self.D.squeeze(), S, lmbda, self.opt['CBPDN'], dimK=dimK,
                           dimN=self.cri.dimN, device=self._device, stream=self._stream)
        x._return_min = False          # the coefficient maps stay on the device
        x.solve()
        self._xstep = x
        self.xstep_itstat = x.itstat[-1] if x.itstat else None

    def _release_xstep(self):
        """Free the previous call's device arrays before the next X-step allocates its own
        (a handle holds a few times the coefficient arrays)."""
        old = getattr(self, '_xstep', None)
        if old is not None:
            old._dev.close()
            self._xstep = None

    def dstep(self):
        """One projected SGD step (onlinecdl.py:310-333): gradient of the data fidelity term at
        the current dictionary, summed over images (and channels for a single-channel
        dictionary), step ``eta``, ``D = Pcn(G)``."""
        dev = self._xstep._dev
        if self._fsz is not None:
            dev.set_filter_sizes(self._fsz)
        dev.ccmod_setcoef(_lib.VAR_Y)            # Zf = rfftn(getcoef())
        dev.ccmod_grad(_lib.VAR_DF)
        self.eta = self.eta_a / (self.j + self.eta_b)
        sums = dev.ccmod_sgd_step(self.eta, self._mxsz[0], self._mxsz[1], self.opt['ZeroMean'])
        self._cnstr = np.sqrt(sums[_lib.OUT_CNSTR])
        self.Dprv[:] = self.D
        self.D[:] = dev.ccmod_getdict(self._mxsz[0], self._mxsz[1]).reshape(self.D.shape)

    @property
    def G(self):
        """The unprojected gradient-step result is not kept (its only use, the ``Cnstr``
        statistic, is computed on the device)."""
        raise AttributeError("G is not kept by the device D-step")

    def getcoef(self):
        """Coefficient maps of the last ``solve`` call."""
        return self._xstep.getcoef()

    # -- statistics and status table ------------------------------------------------------------
    # (display column, IterationStats field) pairs of the Verbose table (onlinecdl.py:362-378)
    _COLUMNS = (('Itn', 'Iter'), ('X r', 'PrimalRsdl'), ('X s', 'DualRsdl'), (u'X ρ', 'Rho'),
                ('D cnstr', 'Cnstr'), ('D dlt', 'DeltaD'), (u'D η', 'Eta'))
    # fields copied from the last row of the X-step's statistics (zero when it kept none)
    _FROM_XSTEP = ('ObjFun', 'DFid', 'RegL1', 'PrimalRsdl', 'DualRsdl', 'Rho')

    @classmethod
    def hdrtxt(cls):
        return tuple(col for col, _ in cls._COLUMNS)

    @classmethod
    def hdrval(cls):
        return dict(cls._COLUMNS)

    def getdict(self):
        return self.D

    def itstat_extra(self):
        return ()

    def iteration_stats(self):
        """One IterationStats row (onlinecdl.py:382-402): the X-step's last objective /
        residual values, the dictionary step's constraint distance, change and step size."""
        row = {f: 0.0 if self.xstep_itstat is None else getattr(self.xstep_itstat, f)
               for f in self._FROM_XSTEP}
        row.update(Iter=self.j, Cnstr=self._cnstr, DeltaD=np.linalg.norm(self.D - self.Dprv),
                   Eta=self.eta, Time=self.timer.elapsed(self.opt['IterTimer']))
        cls = type(self).IterationStats
        extra = dict(zip(type(self).itstat_fields_extra, self.itstat_extra()))
        return cls(**dict(row, **extra))

    def manage_itstat(self):
        itst = self.iteration_stats()
        self.itstat.append(itst)
        self.display_status(self.fmtstr, itst)

    def getitstat(self):
        return util.transpose_ntpl_list(self.itstat)

    def _verbose(self, header=False):
        return self.opt['Verbose'] and (self.opt['StatusHeader'] or not header)

    def display_config(self):
        self.hdrstr, self.fmtstr, self.nsep = '', '', 0
        if self._verbose():
            self.hdrstr, self.fmtstr, self.nsep = common.solve_status_str(
                self.hdrtxt(), fwdth0=self.fwiter, fprec=self.fpothr)

    def display_start(self):
        if self._verbose(header=True):
            print(self.hdrstr + "\n" + "-" * self.nsep)

    def display_status(self, fmtstr, itst):
        if self._verbose():
            print(fmtstr % tuple(getattr(itst, field) for _, field in self._COLUMNS))

    def display_end(self):
        if self._verbose(header=True):
            print("-" * self.nsep)


class OnlineConvBPDNMaskDictLearn(OnlineConvBPDNDictLearn):
    r"""Online convolutional dictionary learning with a spatial mask in the data fidelity term
    (onlinecdl.py:464-600): ``solve(S, W)`` per training image with its mask."""

    class Options(OnlineConvBPDNDictLearn.Options):
        """As :class:`OnlineConvBPDNDictLearn.Options` with ``CBPDN`` the options of
        :class:`sporco_amd.admm.cbpdn.ConvBPDNMaskDcpl` (onlinecdl.py:477-509; note that this
        leaves the X-step at that class's ``MaxMainIter`` of 1000 unless set)."""

        defaults = copy.deepcopy(OnlineConvBPDNDictLearn.Options.defaults)
        defaults.update({'CBPDN': copy.deepcopy(cbpdn.ConvBPDNMaskDcpl.Options.defaults)})

        def __init__(self, opt=None):
            OnlineConvBPDNDictLearn.Options.__init__(self, {
                'CBPDN': cbpdn.ConvBPDNMaskDcpl.Options({
                    'AutoRho': {'Period': 10, 'AutoScaling': False, 'RsdlRatio': 10.0,
                                'Scaling': 2.0, 'RsdlTarget': 1.0}})})
            self.update({} if opt is None else opt)

    def solve(self, S, W=None, dimK=None):
        """Sparse coding and dictionary update for training data ``S`` with mask ``W``
        (onlinecdl.py:513-545)."""
        if dimK is None and self.dimK is not None:
            dimK = self.dimK
        if self.j == 0:
            self.display_start()
        self.timer.start(['solve', 'solve_wo_eval'])
        self.init_vars(S, dimK)
        if W is None:
            W = np.array([1.0], dtype=self.dtype)
        self.xstep(S, W, self.lmbda, dimK)
        self.dstep(W)
        self.timer.stop('solve_wo_eval')
        self.manage_itstat()
        self.j += 1
        self.timer.stop('solve')
        return self.getdict()

    def xstep(self, S, W, lmbda, dimK):
        """ConvBPDNMaskDcpl for the new data with the current dictionary (:549-570)."""
        self._release_xstep()
        x = cbpdn.ConvBPDNMaskDcpl(self.D.squeeze(), S, lmbda, np.asarray(W), self.opt['CBPDN'],
                                   dimK=dimK, dimN=self.cri.dimN, device=self._device,
                                   stream=self._stream)
        x._return_min = False
        x.solve()
        self._xstep = x
        self.xstep_itstat = x.itstat[-1] if x.itstat else None

    def dstep(self, W):
        """Projected SGD step with the residual weighted by the mask (once, in the spatial
        domain) before the adjoint (:574-590); the mask is the one the X-step uploaded."""
        dev = self._xstep._dev
        if self._fsz is not None:
            dev.set_filter_sizes(self._fsz)
        dev.ccmod_setcoef(_lib.VAR_Y)                     # Zf = rfftn(y1), the coefficient maps
        dev.copy(_lib.VAR_DYF, _lib.VAR_DF)
        dev.masked_grad(_lib.VAR_DYF, True, 2)
        self.eta = self.eta_a / (self.j + self.eta_b)
        sums = dev.ccmod_sgd_step(self.eta, self._mxsz[0], self._mxsz[1], self.opt['ZeroMean'])
        self._cnstr = np.sqrt(sums[_lib.OUT_CNSTR])
        self.Dprv[:] = self.D
        self.D[:] = dev.ccmod_getdict(self._mxsz[0], self._mxsz[1]).reshape(self.D.shape)
